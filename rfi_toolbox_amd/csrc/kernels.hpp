// Launch wrappers of every HIP kernel in librfi_hip.so.  All pointers are device pointers,
// all launches go to ctx->stream, nothing here synchronises.
#pragma once
#include "common.hpp"

namespace rfi {

// Optional per-channel transform applied when a kernel LOADS an activation:
//   v = v * scale[c] + shift[c];  if (relu) v = max(v, 0)     (scale == nullptr -> identity)
// This is how BatchNorm-apply + ReLU of the producing layer is folded into its consumers.
struct InXform {
    const float* scale = nullptr;
    const float* shift = nullptr;
    int relu = 0;          // 0: none, 1: ReLU, 2: LeakyReLU(slope)
    float slope = 0.0f;
};
static inline InXform act_xform(const float* scale, const float* shift, float slope) {
    return InXform{scale, shift, slope != 0.0f ? 2 : 1, slope};
}

// A [N,H,W,C] activation view: `pstride` floats between consecutive pixels (>= C; a
// channel slice of a wider concat buffer is a view with pstride = total channels).
struct View {
    const float* p = nullptr;
    int pstride = 0;
};
struct MutView {
    float* p = nullptr;
    int pstride = 0;
};

// ---------------------------------------------------------------- generic "conv-like" contraction
//   y[n, oy, ox, co] (+)= bias[co] + sum_{tap=(r,s), ci} X[n, oy*S + r - pad, ox*S + s - pad, ci] * Wf[tap][co][ci]
// out-of-range input pixels read as zero (after the load transform).
// R x R taps, input stride S.  Cases used:
//   conv3x3 fwd   R=3 S=1 pad=1            Wf = forward layout  [tap][cout][cin]
//   conv3x3 dgrad R=3 S=1 pad=1            Wf = dgrad layout    [tap'][cin][cout] (taps flipped)
//   convT   fwd   R=1 S=1 pad=0, 4 launches-in-one (grid z = (a,b)); output scattered to (2y+a, 2x+b)
//   convT   dgrad R=2 S=2 pad=0            Wf = dgrad layout    [tap][cin][cout]
struct ConvArgs {
    View x;
    int N = 0, H = 0, W = 0;          // output grid ("m" space)
    int Hin = 0, Win = 0;             // input spatial size
    int Cin = 0, Cout = 0;
    const float* w = nullptr;         // [taps][Cout][Cin]
    const float* w3 = nullptr;        // 3 x bf16 mode: the same filters pre-split (launch_weights_to_x3); null: split on the fly
    const unsigned short* wB3 = nullptr;   // 3 x bf16 mode, 3x3 stride-1 convs: the same filters in MFMA B-operand order with three
                                      // planes (planes.hpp, launch_weights_to_wb, one K segment): the wave-specialised
                                      // kernel (conv_ws.hip) runs where this is set and the shape is eligible
    const unsigned short* wB1 = nullptr;   // bf16 mode on float32 tensors (`bf16`): the same with ONE plane (bf16-rounded filters)
    const float* bias = nullptr;      // [Cout] or null
    MutView y;
    int Hout = 0, Wout = 0;           // spatial size of the tensor y points into
    int osy = 1, osx = 1;             // output pixel = (oy*osy + ooy, ox*osx + oox)
    int ooy = 0, oox = 0;
    int R = 3, S = 1, pad = 1;
    int zgroups = 1;                  // convT fwd: 4 (a,b) groups: w += z*Cout*Cin, (ooy,oox) = (z/2, z%2)
    unsigned short* y16 = nullptr;    // convT forward only: write the output as bfloat16 NHWC (a P = 1 plane tensor, planes.hpp)
                                      // instead of float32; y.pstride then counts bf16 elements of that tensor, y.p is unused
    int fold = 0;                     // > 0 (set by launch_conv): the four convT phases folded into the channel dimension --
                                      // Cout = 4 * fold, output channel co belongs to phase z = co / fold, channel co % fold
    InXform xf;
    double algo_flops = -1;           // algorithmic FLOPs for the profile (default: from the shape)
    bool bf16 = false;                // MFMA kernels: operands rounded to bf16 in registers, fp32 accumulate
    bool bf16x3 = false;              // MFMA kernels: float32 operands split into 3 bf16 pieces, 6 bf16 MFMAs per K=16
    int planes = 0;                   // 1 / 3: run on the plane kernels through temporary plane copies (IMPL_PLANES_*)
    // optional: per-channel (sum, sumsq) of the OUTPUT in the epilogue (BatchNorm statistics).  in: stats
    // = [records][Cout][2] doubles with room for stats_max_records; out: stats_records = records written
    // (0 and stats = null when the launch could not provide them: direct kernel, unaligned output)
    double* stats = nullptr;
    int stats_max_records = 0;
    int stats_records = 0;
    // optional: an event that completes WITH this kernel (hipExtLaunchKernel's stop event: the dispatch's own completion
    // signal, no event-record packet behind it -- that packet costs the stream's next kernel ~6.5 us).  done_used: the
    // launcher honoured it (else the caller records an event as usual)
    hipEvent_t done = nullptr;
    bool done_used = false;
    // with `stats`: the output is dA, the gradient w.r.t. the ACTIVATED output of a Conv+BN layer whose raw output
    // is bwd_y (same shape as the output).  The records then hold the BatchNorm-BACKWARD sums (sum dz, sum dz * xhat;
    // dz = dA * act'(y*scale+shift), xhat = (y - mean) * invstd) instead of (sum y, sum y^2): bn_bwd_reduce's pass
    // over dA and Y disappears (launch_bn_bwd_finalize_records finishes them).
    const float* bwd_y = nullptr;
    const float *bwd_scale = nullptr, *bwd_shift = nullptr, *bwd_mean = nullptr, *bwd_invstd = nullptr;
    float bwd_slope = 0.0f;
#ifdef RFI_DIAG_STAMPS
    unsigned long long* stamps = nullptr;   // diagnostic build only: per-workgroup phase cycle sums
#endif
};

// 5 / 6: the plane kernels (planes.hpp) in the 3 x bf16 / bf16 arithmetic; callers holding float32 tensors get
// temporary plane copies (kernel-level ABI, tests)
// 7 / 8: the wave-specialised kernels (conv_ws.hip, gemm_ws.hip) in the 3 x bf16 / the bf16 arithmetic; callers without
// ConvArgs::wB3 / wB1 get a temporary filter copy
enum ConvImpl { IMPL_AUTO = 0, IMPL_DIRECT = 1, IMPL_MFMA = 2, IMPL_MFMA_BF16 = 3, IMPL_MFMA_BF16X3 = 4,
                IMPL_PLANES_X3 = 5, IMPL_PLANES_BF16 = 6, IMPL_WS_X3 = 7, IMPL_WS_BF16 = 8 };
bool conv_ws_eligible(const ConvArgs& a);
void launch_conv_ws(rfi_ctx* ctx, ConvArgs& a, const unsigned short* wB, int P);
bool gemm_ws_eligible(const ConvArgs& a);        // gemm_ws.hip: transposed conv (k2, s2) forward / input gradient, 1x1 convs
void launch_gemm_ws(rfi_ctx* ctx, ConvArgs& a, const unsigned short* wB3);

bool conv_mfma_eligible(const ConvArgs& a);
bool bf16_k16();           // RFI_BF16_K16=1: the float32-tensor bf16 mode runs the K = 16 MFMA on the split path's data flow (conv_mfma.hip)
// filters [taps][Cout][Cin] -> [taps][Cout][ceil(Cin/16)][h16 | m16 | l16] bf16 records (24 floats each)
size_t weights_x3_floats(int taps, int Cout, int Cin);
void launch_weights_to_x3(rfi_ctx* ctx, const float* w, int taps, int Cout, int Cin, float* out);
struct X3Desc { const float* src; float* dst; int64_t rows; int Cin; int nchunks; };
void launch_weights_to_x3_batched(rfi_ctx* ctx, const X3Desc* descs_dev, int n, double total_bytes);
// conv_stem.hip: the first 3x3 conv of a network (Cin = 4 padded, Cout 32 / 64) with K = (tap, channel) packed into three k-steps
bool conv_stem_eligible(const ConvArgs& a);
void launch_conv_stem(rfi_ctx* ctx, ConvArgs& a);
void launch_conv(rfi_ctx* ctx, ConvArgs& a, int impl = IMPL_AUTO);

// ---------------------------------------------------------------- weight gradient
//   dW[tap][cy][cx] = sum_{n,y,x} Yop[n,y,x,cy] * Xop[n, y*S + r - pad, x*S + s - pad, cx]
// written at dw[tap*tap_stride + cy*sy + cx*sx].
//   conv3x3 wgrad: Xop = layer input (cx = cin), Yop = dY (cy = cout), out [tap][cout][cin]: sy=Cin, sx=1
//   convT   wgrad: Xop = dUp (cx = cout, R=2,S=2), Yop = layer input (cy = cin), out [tap][cout][cin]: sy=1, sx=Cin
struct WgradArgs {
    View xop, yop;
    InXform xf_x, xf_y;
    int N = 0, H = 0, W = 0;          // grid of the Yop pixels
    int Hx = 0, Wx = 0;               // spatial size of Xop
    int Cx = 0, Cy = 0;
    int R = 3, S = 1, pad = 1;
    float* dw = nullptr;
    int64_t tap_stride = 0;
    int sy = 0, sx = 0;
    float* slab = nullptr;            // workspace for split partials
    size_t slab_floats = 0;
    double algo_flops = -1;
    bool bf16 = false;                // MFMA kernel: operands rounded to bf16 in registers, fp32 accumulate
    bool bf16x3 = false;              // MFMA kernel: float32 emulated by 3 x bf16 pieces
    int planes = 0;                   // 1 / 3: plane kernel through temporary plane copies (IMPL_PLANES_*)
};
size_t wgrad_slab_floats(const WgradArgs& a, int impl);
bool wgrad_ws_eligible(const WgradArgs& a);      // wgrad_ws.hip: the wave-specialised kernel (float32 tensors, 3 x bf16 or bf16 operands)
void launch_wgrad_ws(rfi_ctx* ctx, const WgradArgs& a);
size_t wgrad_ws_slab_floats(const WgradArgs& a);  // slab workspace of the shapes only wgrad_ws covers (R = 2 / stride 2), else 0
bool wgrad_stem_eligible(const WgradArgs& a);    // wgrad_stem.hip: the first conv of a network (Cx = 4 padded, Cy 32 / 64)
void launch_wgrad_stem(rfi_ctx* ctx, const WgradArgs& a);
void launch_wgrad(rfi_ctx* ctx, const WgradArgs& a, int impl = IMPL_AUTO);

// The raw output Y of a conv layer as the elementwise kernels read it: float32 [M][C] or, in the bf16 data flow
// (planes.hpp, P = 1), bfloat16 [M][ps].  A `const float*` converts implicitly, so float32 call sites read as before.
// (also the gradient tensors a conv kernel writes: float32, or bfloat16 in the bf16 data flow)
struct YRef {
    const void* p = nullptr;
    int bf16 = 0;
    int64_t ps = 0;                   // elements between pixels; 0: C
    YRef(const float* f) : p(f) {}
    YRef(float* f) : p(f) {}
    YRef(const float* f, int64_t pstride) : p(f), ps(pstride) {}
    YRef(const unsigned short* h, int64_t pstride) : p(h), bf16(1), ps(pstride) {}
    YRef(View v) : p(v.p), ps(v.pstride) {}
    int64_t stride(int C) const { return ps ? ps : C; }
};

// ---------------------------------------------------------------- batch norm
// two-stage statistics of a [M][C] tensor: launch_bn_stats writes fp64 (sum, sumsq) partials
// into partial_ws (bn_stats_ws_floats(C) floats); launch_bn_finalize merges them -> mean,
// biased var, invstd, and scale/shift for the consumers' load transform, and updates the
// running stats `ema_repeats` times (momentum 0.1, unbiased var) when running_mean != null.
size_t bn_stats_ws_floats(int C);
void launch_bn_stats(rfi_ctx* ctx, const float* y, int64_t M, int C, float* partial_ws);
void launch_bn_finalize(rfi_ctx* ctx, const float* partial_ws, int64_t M, int C, const float* gamma,
                        const float* beta, float* running_mean, float* running_var,
                        int ema_repeats, float* mean, float* invstd, float* scale, float* shift,
                        float* var_out, int records = 0,    // records > 0: partials came from a conv epilogue
                        hipEvent_t done = nullptr);          // completes with the kernel (a stop event: other streams may wait for scale / shift)
// eval mode: scale/shift from running statistics
void launch_bn_eval_coeffs(rfi_ctx* ctx, int C, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, float* scale,
                           float* shift);
// backward, pass 1: per-channel sum(dz), sum(dz * xhat) with dz = da * (act > 0)
//   -> c1 = mean(dz), c2 = mean(dz*xhat), dgamma, dbeta   (two launches: partial + finalize)
void launch_bn_bwd_reduce(rfi_ctx* ctx, YRef da, YRef y, int64_t M, int C,
                          const float* scale, const float* shift, const float* mean,
                          const float* invstd, float* partial_ws, float* c1, float* c2,
                          float* dgamma, float* dbeta, float slope = 0.0f);
size_t bn_bwd_ws_floats(int64_t M, int C);
// the same finish when the partial sums came out of the producing conv kernel's epilogue (ConvArgs::bwd_y)
void launch_bn_bwd_finalize_records(rfi_ctx* ctx, const float* partial_ws, int records, int64_t M, int C, float* c1,
                                    float* c2, float* dgamma, float* dbeta);
// backward, pass 2 (in place on da): dy = gamma*invstd * (dz - c1 - xhat*c2); also per-channel
// sum(dy) -> dbias_conv (partials in ws, finished by the same launch pair)
void launch_bn_bwd_apply(rfi_ctx* ctx, YRef da_inout, YRef y, int64_t M, int C,
                         const float* scale, const float* shift, const float* mean,
                         const float* invstd, const float* gamma, const float* c1, const float* c2,
                         float* partial_ws, float* dbias, float slope = 0.0f,
                         unsigned short* planes_out = nullptr, int64_t planes_pstride = 0, int planes_P = 0,
                         hipEvent_t done = nullptr, bool finish_dbias = true, const float* head_dl = nullptr,
                         const float* head_w = nullptr);
// head_dl / head_w: da_inout is output only; the incoming gradient is head_dl[pixel] * head_w[channel] (a one-channel 1x1
// head whose launch_head_bwd skipped writing da)
// finish_dbias = false: the per-block sums of dy stay in partial_ws (channel_sum_ws_floats(M, C) floats) and
// launch_bn_bwd_apply_finish turns them into dbias later, e.g. on another stream (nothing consumes dbias before the optimiser)
void launch_bn_bwd_apply_finish(rfi_ctx* ctx, const float* partial_ws, int64_t M, int C, float* dbias);
// ... or for many layers at once: one table entry per sum (partials of `records` blocks, `stride` doubles apart)
struct FinishSumDesc { const double* partial; int records; int64_t stride; int count; float* out; };
int bn_bwd_apply_records(int64_t M, int C);      // per-block partials bn_bwd_apply leaves for an [M][C] tensor
void launch_finish_channel_sums_batched(rfi_ctx* ctx, const FinishSumDesc* descs_dev, int n, int max_count);
// planes_out != null: dy is written as a plane tensor (planes.hpp; P bf16 pieces per value) instead of in place
// done != null: the event completes with the kernel that writes dy (hipExtLaunchKernel's stop event: the dispatch's own
// completion signal) -- another stream can wait for dy without an event-record packet in this stream's queue, which
// costs the next kernel ~6.5 us of idle queue (rocprofv3 kernel trace, tools/trace_gaps.py)

// ---------------------------------------------------------------- pool / head / loss
// a = relu(y*scale+shift) -> skip view (full res) and 2x2 max-pooled p; skip.p == null or pooled == null: that output is left
// out (two launches on two streams: model.cpp)
void launch_bn_relu_pool(rfi_ctx* ctx, const float* y, int N, int H, int W, int C,
                         const float* scale, const float* shift, MutView skip, float* pooled, float slope = 0.0f);
// da[n,y,x,c] = dskip[n,y,x,c] + (argmax of the 2x2 window of a == (y,x) ? dpool : 0)
void launch_pool_bwd_merge(rfi_ctx* ctx, YRef y, int N, int H, int W, int C,
                           const float* scale, const float* shift, YRef dskip, YRef dpool,
                           float* da, float slope = 0.0f, unsigned short* da16 = nullptr);
// da16 != null (bf16 data flow; C % 4 == 0): the merged gradient is stored as a dense bfloat16 [pixel][C] tensor there
// instead of da, and the sums of launch_pool_bwd_merge_sums are those of the stored values (the same for launch_head_bwd)
// the same pass, also writing the BatchNorm-backward sums of that layer (sum dz, sum dz * xhat) as fp64 records into
// partial_ws (bn_bwd_ws_floats) for launch_bn_bwd_finalize_records; returns the record count, or 0 when the shape does
// not fit the scheme (nothing launched: call launch_pool_bwd_merge and the separate reduction instead)
int launch_pool_bwd_merge_sums(rfi_ctx* ctx, YRef y, int N, int H, int W, int C, const float* scale,
                               const float* shift, const float* mean, const float* invstd, YRef dskip, YRef dpool,
                               float* da, float slope, float* partial_ws, unsigned short* da16 = nullptr);
// logits[m,o] = b[o] + sum_c relu(y*scale+shift)[m,c] * w[o][c]
void launch_head_fwd(rfi_ctx* ctx, YRef y, int64_t M, int C, const float* scale,
                     const float* shift, const float* w, const float* b, int Cout, float* logits,
                     float slope = 0.0f);
// x = sigmoid(z); d *= x (1 - x)  (UNetOverfit's sigmoid head, models/unet.py:196)
void launch_sigmoid_fwd(rfi_ctx* ctx, const float* z, int64_t n, float* x);
void launch_sigmoid_bwd(rfi_ctx* ctx, const float* x, int64_t n, float* d_inout);
// loss sums over all logits: [0]=sum bce, [1]=sum sig*y, [2]=sum sig, [3]=sum y  (double)
void launch_loss_reduce(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count,
                        double* partial_ws, double* sums4, float* loss_out);
size_t loss_ws_doubles(int64_t count);
// dlogits from logits + the 4 sums (BCE mean + dice), in place into dlogits
void launch_loss_bwd(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count,
                     const double* sums4, float* dlogits);
// sigmoid focal loss (builder-defined, not in the reference): mean over elements, alpha < 0 = no alpha
void launch_focal_reduce(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count, float alpha,
                         float gamma, double* partial_ws, float* loss_out);
void launch_focal_bwd(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count, float alpha,
                      float gamma, float* dlogits);
// head backward: da[m,c] = sum_o dlogits[m,o]*w[o][c] (NOT yet relu-masked: bn_bwd does that);
// dw[o][c] = sum_m dlogits[m,o]*act[m,c]; db[o] = sum_m dlogits[m,o]
// bn_records_ws != null (+ the layer's batch mean / invstd): where the shape allows (Cout == 1, C % 4 == 0) the same pass
// leaves the BatchNorm-backward sums of the layer as fp64 records there (bn_bwd_ws_floats; a region DISJOINT from
// partial_ws) and returns their count for launch_bn_bwd_finalize_records; 0: not produced
int launch_head_bwd(rfi_ctx* ctx, YRef y, int64_t M, int C, const float* scale,
                    const float* shift, const float* w, int Cout, const float* dlogits, float* da,
                    float* partial_ws, float* dw, float* db, float slope = 0.0f, const float* bn_mean = nullptr,
                    const float* bn_invstd = nullptr, float* bn_records_ws = nullptr, unsigned short* da16 = nullptr,
                    bool* skip_da = nullptr, bool finish = true);
// finish = false (vector path: C % 4 == 0, aligned): the partial records (bn_bwd_apply_records(M, C) of them, Cout * C + Cout
// doubles each: dw then db) stay in partial_ws for launch_finish_channel_sums_batched
// skip_da: in -- the caller can do without da (it will run launch_bn_bwd_apply in its head form); out -- da was not written
// (only where the BatchNorm-backward sums come out of this pass)
size_t head_bwd_ws_floats(int64_t M, int C, int Cout);
// per-channel sum over pixels of a view (convT bias grad)
void launch_channel_sum(rfi_ctx* ctx, View v, int64_t M, int C, float* partial_ws, float* out, bool finish = true);
void launch_channel_sum(rfi_ctx* ctx, YRef v, int64_t M, int C, float* partial_ws, float* out, bool finish = true);      // (a bfloat16 view too)
// finish = false (16-byte aligned views with C % 4 == 0 only): the per-block partials stay in partial_ws (bn_bwd_apply_records(M, C)
// records, C doubles apart) for launch_finish_channel_sums_batched
size_t channel_sum_ws_floats(int64_t M, int C);

// ---------------------------------------------------------------- layouts
void launch_nchw_to_nhwc(rfi_ctx* ctx, const float* src, int N, int C, int H, int W, float* dst);
void launch_nhwc_to_nchw(rfi_ctx* ctx, const float* src, int N, int C, int H, int W, float* dst);
// forward layout [tap][co][ci] -> dgrad layout [tap'][ci][co]; flip: tap' = taps-1-tap
void launch_weight_to_dgrad(rfi_ctx* ctx, const float* wf, int taps, int Cout, int Cin, int flip,
                            float* wd);
struct RelayoutDesc {          // one conv-like layer of launch_weight_to_dgrad_batched (float offsets)
    int64_t src_off, dst_off;
    int taps, cout, cin, flip;
    int64_t tile0 = 0;         // first 32 x 32 transpose tile of this layer in the launch's grid (filled by relayout_assign_tiles)
};
// prefix of 32 x 32 tiles over the layers; returns the total (= gridDim.x of the batched launch)
int64_t relayout_assign_tiles(RelayoutDesc* descs, int n);
void launch_weight_to_dgrad_batched(rfi_ctx* ctx, const RelayoutDesc* descs_dev, int n, const float* src,
                                    float* dst, double total_bytes, int64_t total_tiles);
void launch_u8_to_f32(rfi_ctx* ctx, const uint8_t* src, int64_t n, float* dst);
// dst[m][0..cp) = src[m][0..c) followed by zeros (channel padding of the network input to a multiple of 4)
// dA[i] = Y[i] > 0 ? dA[i] : 0  (backward of a BN-less Conv -> ReLU; n % 4 == 0)
void launch_relu_bwd(rfi_ctx* ctx, float* dA, const float* Y, int64_t n);
void launch_pad_channels(rfi_ctx* ctx, const float* src, int64_t M, int c, int cp, float* dst);

// ---------------------------------------------------------------- optimiser
// sum of squares of g[0..n) -> *sumsq (double), deterministic two-stage
void launch_sumsq(rfi_ctx* ctx, const float* g, int64_t n, double* partial_ws, double* sumsq);
size_t sumsq_ws_doubles(int64_t n);
struct AdamArgs {
    float* p; float* g; float* m; float* v; int64_t n;
    float beta2, eps, wd, max_norm, grad_scale;
    float one_minus_beta1, one_minus_beta2;   // formed in double, rounded once (as torch does)
    float neg_step, bc2_sqrt;                 // -lr/(1-beta1^t), sqrt(1-beta2^t)
    const double* sumsq;            // device scalar: ||g||^2 BEFORE grad_scale
    float* norm_out;                // device scalar out: ||g*grad_scale||
};
void launch_adam(rfi_ctx* ctx, const AdamArgs& a);

// ---------------------------------------------------------------- preprocessing / metrics
void launch_preprocess(rfi_ctx* ctx, const void* patches, int dtype, int n, int ph, int pw,
                       float* minmax_ws, float* out_nhwc, const rfi_patch_src* table_dev = nullptr, int C = 0,
                       int T = 0);
// gather form (table_dev != null): `patches` are the n_planes x C x T waterfall planes and patch i is
// the ph x pw tile table_dev[i] names; the same map serves the labels and the blank-patch test
// order statistics / real-input branch (order_stats.hip); v: [n][per] doubles on the device
// f32 (all of these): the doubles hold float32 values and every arithmetic result is rounded to float32 -- NumPy's
// float32 arithmetic for float32 input
void launch_patch_median(rfi_ctx* ctx, const double* v, int n, int per, bool absdev, const double* centre,
                         bool finite_only, double* out, int* cnt_out, bool f32 = false);
void launch_scale_by_median(rfi_ctx* ctx, double* v, int n, int per, const double* med, bool f32 = false);
void launch_stretch(rfi_ctx* ctx, double* v, int64_t total, int kind, bool f32 = false);          // 1 SQRT, 2 LOG10 of |v|
void launch_replace_inf(rfi_ctx* ctx, double* v, int n, int per, const double* mad, const int* nfinite);
void launch_mad_flags(rfi_ctx* ctx, const double* v, int n, int per, const double* med, const double* mad,
                      double sigma, uint8_t* flags, bool f32 = false);
void launch_narrow_f32(rfi_ctx* ctx, const double* src, int64_t total, float* dst);
void launch_to_abs_f64(rfi_ctx* ctx, const void* src, int dtype, int64_t total, double* dst);
void launch_synth(rfi_ctx* ctx, unsigned long long seed, int n_samples, int n_pol, int C, int T, double noise,
                  int bandpass, int order, double corr, const rfi_event* events_dev, const int* offsets_dev,
                  int out_dtype, void* planes, uint8_t* flags);
void launch_gather_labels(rfi_ctx* ctx, const uint8_t* flags, const rfi_patch_src* table_dev, int C, int T, int n,
                          int ps, uint8_t* out);
void launch_patch_any_flag(rfi_ctx* ctx, const uint8_t* flags, const rfi_patch_src* table_dev, int C, int T, int n,
                           int ps, unsigned* any_out);
void launch_confusion(rfi_ctx* ctx, const void* pred, int pred_dtype, const void* truth,
                      int truth_dtype, int64_t count, unsigned long long* counts3);
void launch_threshold(rfi_ctx* ctx, const float* logits, int64_t count, float threshold,
                      uint8_t* mask);

// x[i] *= f  (the emulated gradient exchange of the single-GPU tests, rfi_comm_emulate)
void launch_scale_inplace(rfi_ctx* ctx, float* x, int64_t n, float f);
// ---------------------------------------------------------------- detection building blocks (detect_kernels.hip)
// RoIAlign over an NHWC feature map; rois = R x (batch index, x1, y1, x2, y2) float32 in image coordinates
void launch_roi_align_fwd(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, const float* rois, int R, float scale,
                          int PH, int PW, int sampling_ratio, bool aligned, float* out);      // out [R][PH][PW][C]
// gather form of the RoIAlign gradient: RoIs sorted by image index; writes EVERY element of dx (no accumulation, no atomics)
void launch_roi_align_bwd_sorted(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, const float* rois, int R, float scale,
                                 int PH, int PW, int sr, bool aligned, float* dx);
void launch_mask_targets(rfi_ctx* ctx, const unsigned char* masks, int G, int H, int W, const float* rois, int R, int PH, int PW,
                         int sr, unsigned char* out);
void launch_anchor_match_batched(rfi_ctx* ctx, const float* anchors, int64_t n, int64_t anchor_stride, const int* anchor_count,
                                 const float* gt, int B, int Gmax, const int* gt_count, float hi, float lo, bool low_quality,
                                 float* best_ws, signed char* labels, int* matched, float* targets);
void launch_nms_batched(rfi_ctx* ctx, const float* boxes, const int* count, int B, int K, float thr, unsigned char* keep);
void launch_roi_align_bwd(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, const float* rois, int R, float scale,
                          int PH, int PW, int sampling_ratio, bool aligned, float* dx);       // dx [N][H][W][C], zeroed here
// FPN top-down merge: out = lateral + nearest 2x upsampling of top ([N][ceil(H/2)][ceil(W/2)][C])
void launch_fpn_merge_fwd(rfi_ctx* ctx, const float* lateral, const float* top, int N, int H, int W, int C, float* out);
void launch_fpn_merge_bwd_top(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, float* dtop);

// ---------------------------------------------------------------- region proposals (rpn_kernels.hip)
// boxes = decode(anchors, deltas) (box-coder weights 1, dw / dh clamped at log(1000/16)), clipped to [0, clip_w] x [0, clip_h]
// when clip_w > 0; anchors [n_anchors][4] repeat over the n = k * n_anchors delta rows
void launch_box_decode(rfi_ctx* ctx, const float* anchors, int64_t n_anchors, const float* deltas, int64_t n, float clip_h,
                       float clip_w, float* out);
// anchors vs G ground-truth boxes (device pointers): labels int8 (1 / 0 / -1), matched ground-truth index (labels 1, else
// -1) and, if targets != null, the encoded regression targets [n][4]; best_ws: G floats of scratch
void launch_anchor_match(rfi_ctx* ctx, const float* anchors, int64_t n, const float* gt, int G, float hi, float lo, bool low_quality,
                         float* best_ws, signed char* labels, int* matched, float* targets);
// suppression bit matrix [n][ceil(n / 64)] of boxes sorted by descending score (bit j of row i: j > i and IoU > thr)
void launch_nms_mask(rfi_ctx* ctx, const float* boxes, int n, float thr, unsigned long long* mask);
// RPN loss on a head output [P][5 A]: BCE on the sampled anchors + smooth L1 on the positives, both / num_sampled;
// writes the gradient w.r.t. the head output and the two loss terms (device floats)
size_t rpn_loss_ws_doubles();
// Fast R-CNN loss on the box head's output [R][5 K1]: mean cross-entropy + smooth L1 of the ground-truth class's deltas over the
// foreground RoIs / R; labels int32 in [0, K1) (0 = background); writes the gradient and the two terms (partial_ws as rpn_loss)
void launch_fastrcnn_loss(rfi_ctx* ctx, const float* head, int64_t R, int K1, const int* labels, const float* targets, float beta,
                          float* dhead, double* partial_ws, float* loss2_dev);
void launch_rpn_loss(rfi_ctx* ctx, const float* head, int64_t P, int A, const signed char* labels, const float* targets,
                     int64_t num_sampled, float beta, float* dhead, double* partial_ws, float* loss2_dev,
                     const int* num_sampled_dev = nullptr);      // num_sampled_dev != null: the normaliser is read from the device

// ---------------------------------------------------------------- the detector's box bookkeeping on the device (detect_sample.hip)
// keys: 64-bit, one segment of `stride` (a power of two <= 8192) keys per workgroup, sorted ascending in place
void launch_segsort_u64(rfi_ctx* ctx, unsigned long long* keys, int n_segs, int stride);
// sampler keys of labels [B][n] (1 positive, 0 negative, else neither): class << 48 | Philox(seed; i, b, stream, step) << 16 | i;
// entries beyond count[b] (null: n) or n get the largest key
void launch_sample_keys(rfi_ctx* ctx, const signed char* labels, int B, int n, const int* count, unsigned long long seed, unsigned step,
                        unsigned stream0, unsigned long long* keys, int stride);
// RPN sampler on the SORTED keys: `batch` anchors per image with at most max_pos positives keep their label, the rest of the
// labels >= 0 become -1; labels / targets are written per pyramid level ([images][anchors of the level]); *n_sampled += count
void launch_rpn_sample_apply(rfi_ctx* ctx, const unsigned long long* keys_sorted, int B, int n, int stride, int batch, int max_pos,
                             const signed char* labels, const float* targets, int L, const int* level_off, signed char* const* level_labels,
                             float* const* level_targets, int* n_sampled);
// top-k of the objectness logits of one pyramid level: keys of head [B][P][5 A] (descending score, ties by anchor index) ...
void launch_topk_keys(rfi_ctx* ctx, const float* head, int B, int P, int A, unsigned long long* keys, int stride);
// ... and, from the sorted keys, the K best decoded + clipped into slot `lvl` of boxes [B][L][K][4] / scores [B][L][K] (boxes
// under min_size moved behind the others, score -inf) with counts [B][L] of the boxes that remain
void launch_topk_decode(rfi_ctx* ctx, const unsigned long long* keys_sorted, int B, int stride, int P, int A, int K, const float* head,
                        const float* anchors, float clip_h, float clip_w, float min_size, float* boxes, float* scores, int* counts, int L,
                        int lvl);
// per image: the post_nms best kept boxes of its L x K candidates, then its ground-truth boxes -> props [B][Pmax][4], pcount [B]
void launch_proposals_select(rfi_ctx* ctx, const float* boxes, const float* scores, const unsigned char* keep, int B, int L, int K,
                             int post_nms, const float* gt, int Gmax, const int* gt_count, int Pmax, float* props, int* pcount);
// RoI sampler: `batch` proposals per image with at most max_pos foreground -> sel [B][batch] (positives first), nsel / npos [B]
void launch_roi_sample(rfi_ctx* ctx, const signed char* labels, const int* pcount, int B, int Pmax, int batch, int max_pos,
                       unsigned long long seed, unsigned step, unsigned stream0, int* sel, int* nsel, int* npos);
// the sampled RoIs of the batch, image-major and compact, with class labels, targets, matched instance, pyramid level
// (0 + [area >= t1] + [area >= t2] + [area >= t3]), the foreground rows again for the mask branch, counts = (R, Rf)
void launch_roi_compact(rfi_ctx* ctx, const int* sel, const int* nsel, const int* npos, int B, int batch, int Pmax, const float* props,
                        const int* matched, const float* targets, const int* gt_labels, int Gmax, const int* gbase, float t1, float t2,
                        float t3, float* rois, int* cls, float* tgt, int* gt, int* level, int* img_start, float* rois_fg, float* rois_gt,
                        int* level_fg, int* fg_start, int* counts);
// multi-level RoIAlign: RoI r reads level[r] of maps[4] ([N][H0 >> k][W0 >> k][C], spatial scale scale0 / 2^k); the RoI count is
// read from the device (rows beyond it are not written); backward: ADDS every level's gradient into dmaps[k] (gather form:
// the RoIs of image n are rows [img_start[n], img_start[n + 1]))
void launch_roi_align_ml_fwd(rfi_ctx* ctx, const float* const* maps, int N, int H0, int W0, int C, float scale0, const float* rois,
                             const int* level, const int* count_dev, int max_rois, int PH, int PW, int sr, float* out);
void launch_roi_align_ml_bwd(rfi_ctx* ctx, float* const* dmaps, int N, int H0, int W0, int C, float scale0, const float* dout,
                             const float* rois, const int* level, const int* img_start, int max_rois, int PH, int PW, int sr);

// ---------------------------------------------------------------- ResNet-style encoder pieces (resnet_kernels.hip)
void launch_s2d(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, float* out);          // [N,H,W,C] -> [N,H/2,W/2,4C]
void launch_d2s_add(rfi_ctx* ctx, const float* dxp, const float* ds, View extra, int N, int H, int W, int C, float* out);
void launch_bn_add_relu(rfi_ctx* ctx, const float* y, const float* scale, const float* shift, const float* s,
                        const float* s_scale, const float* s_shift, int64_t M, int C, MutView out, MutView out2);
void launch_relu_mask(rfi_ctx* ctx, View da, View da2, View a, View base, int64_t M, int C, float* dz);   // (da + da2) * (a > 0) + base
void launch_add_inplace(rfi_ctx* ctx, float* x, const float* y, int64_t n);
// device-to-device copy as a kernel on the context's stream (hipMemcpyAsync's blit path idles the queue ~55 us per copy)
void launch_copy_d2d(rfi_ctx* ctx, void* dst, const void* src, size_t bytes);
// 3x3 stride-2 filters [9][Cout][Cin] <-> their 2x2 form on the space-to-depth input [4][Cout][4 Cin]
void launch_w_s2d(rfi_ctx* ctx, float* w3, int Cout, int Cin, float* w2, bool to_s2d);

// MaxPool2d(3, 2, 1) of act(y * scale + shift) (scale null: of y) + first-argmax bytes (4 channels per word); its adjoint
// (gather); x[:, ::2, ::2] and its adjoint added into dx
void launch_maxpool3_fwd(rfi_ctx* ctx, const float* y, int N, int H, int W, int C, const float* scale, const float* shift, float* out,
                         unsigned* arg4);
void launch_maxpool3_bwd(rfi_ctx* ctx, const float* dout, const unsigned* arg4, int N, int H, int W, int C, float* da);
void launch_subsample2(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, float* out);
void launch_subsample2_bwd_add(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, float* dx);

// K-packed stem: out [N, OH, OW, Kp] with k = (r R + s) C + c (zeros beyond R R C and outside the image); the filters
// [R R][Cout][C] <-> [Cout][Kp] (to_packed = false writes the packed GRADIENT back into the tap layout)
void launch_im2col(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, int R, int S, int pad, int OH, int OW, int Kp, float* out);
void launch_w_pack(rfi_ctx* ctx, float* w, int taps, int Cout, int C, int Kp, float* wp, bool to_packed);

// generic: out[i] = sum_s slabs[s*n + i]
void launch_reduce_slabs(rfi_ctx* ctx, const float* slabs, int nslabs, int64_t n, float* out);

}  // namespace rfi
