// "Plane" activation format and the MFMA kernels that consume it (gfx950).
//
// A plane tensor holds an activation as the bf16 pieces the matrix cores consume, laid out so that the
// contraction kernels stage it with LDS-DMA copies only (no VALU, no registers):
//
//     [pixel][chunk = channel / 16][plane p < P][16 channels] bf16        (32 bytes per plane row)
//
// P = 1: plain bf16 NHWC (the bf16 compute mode: activations live in HBM as bf16).
// P = 3: float32 carried as three bf16 pieces, v = h + m + l with h = bf16(v), m = bf16(v - h),
//        l = bf16(v - h - m) (RNE; 3 x 8 significand bits): the float32-by-3xbf16 arithmetic reads the pieces
//        straight from HBM instead of splitting every element again in every consumer.
// Channels are padded with zeros to a multiple of 16.  A tensor is produced ONCE, by the elementwise kernel that
// materialises the activation anyway (BatchNorm-apply + ReLU, pooling, BatchNorm backward), and read by every
// contraction that consumes it (forward conv, its weight gradient, the input-gradient conv).
#pragma once
#include "kernels.hpp"

namespace rfi {

typedef unsigned short bf16_t;

struct PlaneSeg {                 // one K-segment of a contraction input
    const bf16_t* p = nullptr;
    int64_t pstride = 0;          // bf16 elements per pixel (>= nchunks * P * 16)
    int nchunks = 0;              // 16-channel chunks
};
__host__ __device__ static inline int plane_chunks(int C) { return (C + 15) / 16; }
static inline size_t plane_elems(int64_t pixels, int C, int P) { return (size_t)pixels * plane_chunks(C) * P * 16; }

// ---------------------------------------------------------------- producers
// fp32 [M][C] view (pstride floats per pixel) -> planes; optional BatchNorm-apply + (Leaky)ReLU in flight
// x16 != null: the input is a bfloat16 NHWC tensor (a raw conv output of the bf16 data flow) instead of x
void launch_act_split(rfi_ctx* ctx, View x, int64_t M, int C, InXform xf, int P, bf16_t* out, int64_t out_pstride,
                      const bf16_t* x16 = nullptr, int64_t x16_pstride = 0);
// a = act(y*scale+shift) -> skip planes (full resolution) + 2x2 max-pooled planes (H, W even)
void launch_bn_relu_pool_planes(rfi_ctx* ctx, const float* y, int N, int H, int W, int C, const float* scale,
                                const float* shift, float slope, int P, bf16_t* skip, int64_t skip_pstride,
                                bf16_t* pooled, int64_t pooled_pstride, const bf16_t* y16 = nullptr, int64_t y16_pstride = 0);
// ResNet-style encoder on the bf16 data flow (bfloat16 [pixel][>= C] tensors, 16-byte aligned rows, C % 8 == 0):
// out = relu(y * scale + shift + shortcut), shortcut = s * s_scale + s_shift (a projection's raw output) or s (s_scale null)
void launch_bn_add_relu16(rfi_ctx* ctx, const bf16_t* y, int64_t y_ps, const float* scale, const float* shift, const bf16_t* s, int64_t s_ps,
                          const float* s_scale, const float* s_shift, int64_t M, int C, bf16_t* out, int64_t out_ps);
// dz = (g0 + g1 + g2) * (a > 0): g1 / g2 (a float32 or bfloat16 view) / the mask are optional
void launch_relu_mask_sum16(rfi_ctx* ctx, const bf16_t* g0, int64_t p0, const bf16_t* g1, int64_t p1, YRef g2, const bf16_t* a, int64_t pa,
                            int64_t M, int C, bf16_t* dz, int64_t pz);
// input gradient of a 3x3 stride-2 pad-1 conv as four 2x2 stride-1 pad-0 contractions of dY, one per parity class
// c = 2 py + px of the input pixel (written with output stride 2): float32 filters [class][tap t = 2 ty + tx][Cin][K] with
// K = Cout (class 0: 2 Cout -- its second K segment is the 1x1 stride-2 projection `wp` [Cout][Cin], whose input gradient
// lands on the same pixels; null: zeros; w3 may be null too).  Tap (ty, tx) of class (py, px) is filter tap r = py ? (ty ? 0 : 2) : (ty ? none : 1)
// (and the same along x); taps without a filter entry are zero.  s2_class_floats: floats of the whole table.
static inline size_t s2_class_floats(int Cout, int Cin) { return (size_t)20 * Cin * Cout; }
static inline size_t s2_class_offset(int c, int Cout, int Cin) { return c == 0 ? 0 : (size_t)(8 + 4 * (c - 1)) * Cin * Cout; }
void launch_w_s2_classes(rfi_ctx* ctx, const float* w3, const float* wp, int Cout, int Cin, float* dst);
// planes -> fp32 (tests / debugging): sum of the pieces
void launch_planes_to_f32(rfi_ctx* ctx, const bf16_t* in, int64_t in_pstride, int64_t M, int C, int P, float* out,
                          int out_pstride);

// ---------------------------------------------------------------- filters in MFMA B-operand order
// wB[z][tap][kc][cb][plane][lane 0..63][8] bf16: lane (j = lane & 31, lh = lane >> 5) holds filter values of
// output channel cb*32 + j for input channels 16*kc' + 8*lh + e (kc' = chunk within its segment), so one
// wave-wide 16-byte load is one whole operand fragment.  Input channels come in up to two K-segments
// (seg_c[0] + seg_c[1] = Cin of the source), each padded to whole chunks.
struct WBDesc {
    const float* src;             // [taps][Cout][Cin] float32 (forward layout or dgrad layout)
    bf16_t* dst;
    int taps, Cout, Cin;
    int seg_c[2];                 // channels per K-segment (seg_c[1] may be 0)
    int P;
};
static inline int wb_kchunks(const WBDesc& d) { return plane_chunks(d.seg_c[0]) + (d.seg_c[1] ? plane_chunks(d.seg_c[1]) : 0); }
static inline size_t wb_elems(int taps, int Cout, int seg0, int seg1, int P) {
    const size_t kc = (size_t)plane_chunks(seg0) + (seg1 ? plane_chunks(seg1) : 0);
    return (size_t)taps * kc * ((Cout + 31) / 32) * P * 512;
}
void launch_weights_to_wb(rfi_ctx* ctx, const WBDesc* descs_dev, int n, double total_bytes);   // one block row per desc
void launch_weights_to_wb_one(rfi_ctx* ctx, const WBDesc& d);                                   // uploads the descriptor itself

// ---------------------------------------------------------------- conv-like contraction on planes
//   y[n,oy,ox,co] = bias[co] + sum_{tap,ci} X[n, oy*S+r-pad, ox*S+s-pad, ci] * W[tap][co][ci]
struct PConvArgs {
    PlaneSeg x[2];
    int nseg = 1;
    int P = 3;
    int N = 0, H = 0, W = 0;          // output grid
    int Hin = 0, Win = 0;
    int Cout = 0;
    const bf16_t* wB = nullptr;
    const float* bias = nullptr;
    float* y = nullptr;               // raw float32 NHWC output (pre-BatchNorm), y_pstride floats per pixel
    int y_pstride = 0;
    bf16_t* y16 = nullptr;            // bf16 data flow: the raw output as bfloat16 NHWC instead (y unused; y_pstride counts bf16
                                      // elements); the statistics are then those of the ROUNDED values
    bool round_y = false;             // float32 output holding bf16-rounded values (same arithmetic as y16, for consumers that
                                      // read float32 tensors)
    int Hout = 0, Wout = 0;
    int osy = 1, osx = 1, ooy = 0, oox = 0;
    int zblocks = 0;                  // > 0 (ConvTranspose2d(k2, s2) as ONE 1x1 contraction with 4 Cz output channels, Cz = 32 zblocks):
                                      // output channel block cb belongs to filter tap z = cb / zblocks = 2 a + b and lands at pixel
                                      // (oy osy + ooy + a, ox osx + oox + b), channels 32 (cb % zblocks) ..; bias has Cz entries
    int R = 3, S = 1, pad = 1;
    double* stats = nullptr;          // as ConvArgs::stats
    int stats_max_records = 0;
    int stats_records = 0;
    // bwd_y16 != null (with stats; input-gradient convs): the output is the gradient dA w.r.t. the ACTIVATED output of a
    // conv + BatchNorm + (Leaky)ReLU layer whose raw bfloat16 conv output is bwd_y16 (same pixels, bwd_yps elements per
    // pixel).  The records then hold that layer's BatchNorm-BACKWARD sums (sum dz, sum dz * xhat with dz = dA * act'(z),
    // of the values as stored) instead of the output statistics: bn_bwd_reduce's pass over dA and Y disappears.
    const bf16_t* bwd_y16 = nullptr;
    int64_t bwd_yps = 0;
    const float* bwd_scale = nullptr;
    const float* bwd_shift = nullptr;
    const float* bwd_mean = nullptr;
    const float* bwd_invstd = nullptr;
    float bwd_slope = 0.0f;
    double algo_flops = -1;
};
void launch_pconv(rfi_ctx* ctx, PConvArgs& a);
void launch_pconv_from_f32(rfi_ctx* ctx, ConvArgs& c, int P);     // float32 tensors in: temporary plane copies

// ---------------------------------------------------------------- weight gradient on planes
//   dW[tap][cy][cx] = sum_{n,y,x} Yop[n,y,x,cy] * Xop[n, y*S+r-pad, x*S+s-pad, cx]   (Xop in up to two K-segments)
struct PWgradArgs {
    PlaneSeg xop[2];
    int nseg = 1;
    int seg_c[2] = {0, 0};            // true channel counts of the Xop segments (Cx = seg_c[0] + seg_c[1])
    int cx_layout = 0;                // cx entries of the dw layout (>= Cx; e.g. the stem's input channels padded to 4):
                                      // entries Cx .. cx_layout - 1 are written as zeros.  0: = Cx
    PlaneSeg yop;
    int Cy = 0;
    int P = 3;
    int N = 0, H = 0, W = 0;          // grid of the Yop pixels
    int Hx = 0, Wx = 0;
    int R = 3, S = 1, pad = 1;
    float* dw = nullptr;              // dw[tap*tap_stride + cy*sy + cx*sx]
    int64_t tap_stride = 0;
    int sy = 0, sx = 0;
    float* slab = nullptr;
    size_t slab_floats = 0;
    double algo_flops = -1;
};
size_t pwgrad_slab_floats(const PWgradArgs& a);
void launch_pwgrad(rfi_ctx* ctx, const PWgradArgs& a);
void launch_pwgrad_from_f32(rfi_ctx* ctx, const WgradArgs& w, int P);     // float32 tensors in: temporary plane copies
size_t pwgrad_slab_floats_f32(const WgradArgs& w);

}  // namespace rfi
