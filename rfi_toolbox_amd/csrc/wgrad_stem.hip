// Weight gradient of the first 3x3 conv of a network (stem): Cx = 3 input channels padded to 4, Cy = 32 or 64 output
// channels, stride 1, pad 1, float32 tensors, the float32-by-3xbf16 arithmetic.
//
//   dW[tap][cy][cx] = sum_{n,y,x} dA[n,y,x,cy] * X[n, y + r - 1, x + s - 1, cx]
//
// The general kernel (wgrad_ws.hip) stages X as a 32-channel block of which 4 channels carry data and runs nine taps of
// full 32 x 32 blocks: 85 us at the very end of the U-Net's backward pass (nothing left to overlap with).  Here the GEMM is
// M = cy, N = (tap, cx) = 36 columns in two 32-column blocks, K = pixels: the Xop fragment of a k-step (16 pixels) comes from
// a halo image of [pixel][4 channels] bf16 per plane through the TRANSPOSING LDS read, whose per-lane row address lets
// every 4-column group of the fragment point at a DIFFERENT tap's pixel -- the im2col lives in the addresses, nothing is
// materialised.  12 MFMAs per k-step and cy block instead of 54; what remains is reading dA once (134 MB at batch 64 x
// 128^2 x 32: HBM bound).
//
// Plain 4-wave workgroups, one 8 x 16-pixel tile at a time (wave w: rows 2 w, 2 w + 1), partial sums in registers across a
// workgroup's tiles, one slab per workgroup, reduce_slabs at the end (fixed order: no float atomics).
#include <algorithm>

#include "ws_common.hpp"

namespace rfi {
namespace {

using namespace ws;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WStemDev {
    const float* x;                   // [N][H][W][4]
    const float* dy;                  // [N][H][W][Cy]
    int N, H, W, Cy;
    float* slab;                      // [gridDim.x][9][Cy][4]
};

constexpr int TH = 8, TW = 16, BM = TH * TW, HH = TH + 2, HW = TW + 2, HP = HH * HW;
constexpr int ROW = 192;              // bytes per pixel of a 32-channel dA block image: 3 planes x 32 channels x 2 B
constexpr int XPLANE = HP * 8;        // bytes of one X plane image: [halo pixel][4 bf16]

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NT>
__global__ __launch_bounds__(256) void wgrad_stem_kernel(WStemDev d) {
    constexpr int Y_BYTES = NT * BM * ROW, X_BYTES = 3 * XPLANE;
    constexpr int RED_FLOATS = NT * 2 * 1024;            // one wave's accumulators
    constexpr int LDS = (Y_BYTES + X_BYTES) > RED_FLOATS * 4 ? (Y_BYTES + X_BYTES) : RED_FLOATS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
    unsigned char* const sY = smem;
    unsigned char* const sX = smem + Y_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fragment addressing (wgrad_ws.hip): 16-lane group gq: columns 16 (gq & 1) .. + 15 of the 32-column block, pixel half
    // kh = gq >> 1; inside the group lane 4 q + pc supplies the address of block row q (pixel q of the 4), 4-column chunk pc
    const int ll = lane & 15, q = ll >> 2, pc = ll & 3, gq = lane >> 4, kh = gq >> 1;
    const int y_lane = (gq & 1) * 32 + pc * 8;           // bytes inside a plane row of the dA image
    int x_tap[2];                                        // halo-pixel offset of this lane's tap in the two column blocks
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int tap = std::min(cb * 8 + (gq & 1) * 4 + pc, 8);     // (columns of taps 9 .. 15 are never stored)
        x_tap[cb] = (tap / 3) * HW + tap % 3;
    }
    f32x16 acc[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][cb][r] = 0.0f;

    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int ntiles = d.N * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ox0 = (t % tiles_x) * TW, oy0 = ((t / tiles_x) % tiles_y) * TH, n = t / (tiles_x * tiles_y);
        __syncthreads();                                 // the previous tile's fragment reads are done
        // ---- stage dA: [pixel][32 NT channels] float32 -> per 32-channel block [pixel][h 64 B | m 64 B | l 64 B]
        for (int idx = tid; idx < BM * 8 * NT; idx += 256) {
            const int c4 = idx % (8 * NT), pix = idx / (8 * NT);
            const int y = oy0 + pix / TW, x = ox0 + pix % TW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y < d.H && x < d.W) v = *reinterpret_cast<const f32x4*>(d.dy + ((size_t)(n * d.H + y) * d.W + x) * d.Cy + c4 * 4);
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v.x, v.y, h0, m0, l0);
            split_pair(v.z, v.w, h1, m1, l1);
            unsigned char* dst = sY + (c4 >> 3) * (BM * ROW) + pix * ROW + (c4 & 7) * 8;
            *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dst + 64) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(dst + 128) = u32x2{l0, l1};
        }
        // ---- stage X: the halo tile, float4 per pixel -> three planes of 8 bytes
        for (int idx = tid; idx < HP; idx += 256) {
            const int iy = oy0 - 1 + idx / HW, ix = ox0 - 1 + idx % HW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
                v = *reinterpret_cast<const f32x4*>(d.x + ((size_t)(n * d.H + iy) * d.W + ix) * 4);
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v.x, v.y, h0, m0, l0);
            split_pair(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2*>(sX + idx * 8) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(sX + XPLANE + idx * 8) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(sX + 2 * XPLANE + idx * 8) = u32x2{l0, l1};
        }
        __syncthreads();
        // ---- wave w: k-steps (tile rows) 2 w and 2 w + 1
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = 2 * wave + rb;
            const int t0 = row * TW + 8 * kh + q, t1 = t0 + 4;                       // this lane's pixel rows of the two reads
            const int h0 = row * HW + 8 * kh + q, h1 = h0 + 4;                       // ... as halo pixels of tap (0, 0)
            bf16x8 bf[2][3];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bf[cb][p] = tr_frag(sX + p * XPLANE + (h0 + x_tap[cb]) * 8, sX + p * XPLANE + (h1 + x_tap[cb]) * 8);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x8 af[3];
                const unsigned char* yimg = sY + nt * (BM * ROW) + y_lane;
#pragma unroll
                for (int p = 0; p < 3; ++p) af[p] = tr_frag(yimg + t0 * ROW + p * 64, yimg + t1 * ROW + p * 64);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[nt][cb] = mma3(af, bf[cb], acc[nt][cb]);
            }
        }
    }
    // ---- the four waves add their accumulators through LDS (fixed order), wave 0 writes the workgroup's slab
    float* s_red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s_red[((nt * 2 + cb) * 16 + r) * 64 + lane] = acc[nt][cb][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nt][cb][r] += s_red[((nt * 2 + cb) * 16 + r) * 64 + lane];
        }
    }
    if (wave == 0) {
        // C row = cy: (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of the block; column lane & 31 = 4 (tap - 8 cb) + cx
        float* slab = d.slab + (size_t)blockIdx.x * 9 * d.Cy * 4;
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int tap = cb * 8 + (li >> 2), cx = li & 3;
            if (tap < 9) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cy = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (cy < d.Cy) slab[((size_t)tap * d.Cy + cy) * 4 + cx] = acc[nt][cb][r];
                    }
            }
        }
    }
}

}  // namespace

bool wgrad_stem_eligible(const WgradArgs& a) {
    if (!(a.R == 3 && a.S == 1 && a.pad == 1 && a.Cx == 4 && a.xop.pstride == 4 && (a.Cy == 32 || a.Cy == 64) && a.yop.pstride == a.Cy)) return false;
    if (a.Hx != a.H || a.Wx != a.W || a.xf_x.scale || a.xf_y.scale) return false;
    if (a.sy != 4 || a.sx != 1 || a.tap_stride != (int64_t)4 * a.Cy) return false;
    if ((reinterpret_cast<uintptr_t>(a.xop.p) | reinterpret_cast<uintptr_t>(a.yop.p)) & 15) return false;
    return a.slab != nullptr && a.slab_floats >= (size_t)64 * 9 * a.Cy * 4;
}

void launch_wgrad_stem(rfi_ctx* ctx, const WgradArgs& a) {
    RFI_REQUIRE(wgrad_stem_eligible(a) && a.bf16x3, "wgrad_stem: shape or arithmetic not eligible");
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int64_t stride = (int64_t)9 * a.Cy * 4;
    int GX = std::min(ntiles, a.Cy == 32 ? 1024 : 768);                          // 4 (3) workgroups per CU
    GX = (int)std::min<int64_t>(GX, (int64_t)(a.slab_floats / (size_t)stride));
    WStemDev d{a.xop.p, a.yop.p, a.N, a.H, a.W, a.Cy, a.slab};
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * 9.0 * 3 * a.Cy;
        const double bytes = 4.0 * ((double)a.N * a.H * a.W * (4 + a.Cy) + 9.0 * 4 * a.Cy);
        std::string label;
        if (ctx->profiling)
            label = "wgrad_stem N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" + std::to_string(a.W) + " cx4 cy" + std::to_string(a.Cy) +
                    " split" + std::to_string(GX) + " 3xbf16";
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
        if (a.Cy == 32) hipLaunchKernelGGL(wgrad_stem_kernel<1>, dim3(GX), dim3(256), 0, ctx->stream, d);
        else hipLaunchKernelGGL(wgrad_stem_kernel<2>, dim3(GX), dim3(256), 0, ctx->stream, d);
        check_launch("wgrad_stem");
    }
    launch_reduce_slabs(ctx, a.slab, GX, stride, a.dw);
}

}  // namespace rfi
