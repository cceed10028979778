// Wave-specialised weight gradient of the 3x3 (and 2x2 / 1x1, stride 1) convolutions for FLOAT32 operand tensors, in the
// float32-by-3xbf16 arithmetic (P = 3) or with bf16 operands (P = 1).
//
//   dW[tap][cy][cx] = sum_{n,y,x} T(Yop)[n,y,x,cy] * T(Xop)[n, y+r-pad, x+s-pad, cx]
//
// GEMM view, tiling, fragment addressing and slab scheme of wgrad_split.hip (one 64 x 64 channel tile per workgroup, 8 x 8
// pixel tiles = four k-steps of 16 pixels, nine accumulators per wave, transposing LDS reads).  What differs is WHO
// stages: waves 4-7 (producers) load the two operand tiles of pixel tile k + 1 as float32, apply the load transforms,
// split every value once and write the plane images into the OTHER half of a double-buffered LDS, while waves 0-3
// (consumers) run nothing but ds_read_b64_tr_b16 + MFMA on tile k -- wgrad_split.hip staged synchronously and relied on
// a second workgroup per CU to fill the gaps (matrix pipe busy 0.57).  One workgroup per CU also means HALF the partial
// slabs (256 workgroups instead of 512), i.e. half the bytes reduce_slabs has to move.
#include <algorithm>

#include "ws_common.hpp"

namespace rfi {
namespace {

using namespace ws;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct WWsDev {
    WgradArgs a;
    int nsplit;
    int64_t slab_stride;
    unsigned x_bytes, y_bytes;
};

template <int R, int BYB, int BXB, int TH, int TW, int P>
struct WWCfg {
    static constexpr int NTAP = R * R;
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH + R - 1, HW = TW + R - 1, HP = HH * HW;
    static constexpr int BLOCKS = BYB * BXB;
    static constexpr int WP = 4 / BLOCKS;                       // consumer waves that share one channel block (split the k-steps)
    static constexpr int KS = BM / 16, KS_W = KS / WP;
    static constexpr int ROW = P * 64;                          // bytes per pixel of a 32-channel block image: P planes x 32 channels
    static constexpr int YQ = BYB * 8, XQ = BXB * 8;            // float4 groups per pixel
    static constexpr int Y_ITEMS = (BM * YQ + 255) / 256, X_ITEMS = (HP * XQ + 255) / 256;
    static constexpr int Y_BYTES = BYB * BM * ROW, X_BYTES = BXB * HP * ROW;
    static constexpr int STAGE = Y_BYTES + X_BYTES;
    static constexpr int TC = (NTAP % 3 == 0) ? 3 : NTAP;
    static constexpr int RED_BYTES = (WP > 1) ? BLOCKS * TC * 4096 : 0;
    static constexpr int LDS_BYTES = 2 * STAGE > RED_BYTES ? 2 * STAGE : RED_BYTES;
    static_assert(BLOCKS == 1 || BLOCKS == 2 || BLOCKS == 4, "1, 2 or 4 channel blocks");
    static_assert(BM % 16 == 0 && KS % WP == 0 && TW % 4 == 0, "tile must split into k-steps of 16 pixels");
    static_assert(256 % YQ == 0 && 256 % XQ == 0, "a producer thread keeps one channel group for all its items");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int P>
__device__ __forceinline__ f32x16 mma(const bf16x8 (&a)[P], const bf16x8 (&b)[P], f32x16 acc) {
    if constexpr (P == 3) return mma3(a, b, acc);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}
// one operand fragment: this lane's 8 pixels (k = 8 h + 0..7) of its channel, from two transposing reads of 4 pixel rows
// each.  `p0` / `p1`: byte addresses of THIS lane's row of the two 4x16 blocks
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// barrier behind this wave's LDS writes (the producers have left by then: a terminated wave is not waited for)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int R, int BYB, int BXB, int TH, int TW, int P>
__global__ __launch_bounds__(512) void wgrad_ws_kernel(WWsDev d) {
    using C = WWCfg<R, BYB, BXB, TH, TW, P>;
    const WgradArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cy0 = blockIdx.y * BYB * 32, cx0 = blockIdx.z * BXB * 32;
    const int split = blockIdx.x;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    const int my_tiles = split < ntiles ? (ntiles - split + d.nsplit - 1) / d.nsplit : 0;   // (>= 1: nsplit <= ntiles)

    if (wave >= 4) {
        // =============================================================== producers
        const int ptid = tid - 256;
        const int yq = ptid % C::YQ, xq = ptid % C::XQ;
        const int cyq = cy0 + yq * 4, cxq = cx0 + xq * 4;
        const bool y_cok = cyq < a.Cy, x_cok = cxq < a.Cx;            // Cx, Cy % 4 == 0 (launch precondition)
        f32x4 ysc = {1.f, 1.f, 1.f, 1.f}, ysh = {0.f, 0.f, 0.f, 0.f}, xsc = ysc, xsh = ysh;
        if (a.xf_y.scale && y_cok) {
            ysc = *reinterpret_cast<const f32x4*>(a.xf_y.scale + cyq);
            ysh = *reinterpret_cast<const f32x4*>(a.xf_y.shift + cyq);
        }
        if (a.xf_x.scale && x_cok) {
            xsc = *reinterpret_cast<const f32x4*>(a.xf_x.scale + cxq);
            xsh = *reinterpret_cast<const f32x4*>(a.xf_x.shift + cxq);
        }
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.yop.p), 0, (int)d.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xop.p), 0, (int)d.x_bytes, 0x00020000);
        // LDS byte offset of this thread's 4 channels inside a pixel row of its 32-channel block image
        const int y_lds = (yq >> 3) * (C::BM * C::ROW) + (yq & 7) * 8, x_lds = (xq >> 3) * (C::HP * C::ROW) + (xq & 7) * 8;
        u32x4 yreg[C::Y_ITEMS], xreg[C::X_ITEMS];
        unsigned yvalid = 0, xvalid = 0;
        // buffer loads: an item outside the image (or the channel range) reads offset 2^31, i.e. zero
        auto load_tile = [&](int tile) {
            const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
            const int oy0 = ty_i * TH, ox0 = tx_i * TW;
            const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
            yvalid = 0;
            xvalid = 0;
#pragma unroll
            for (int it = 0; it < C::Y_ITEMS; ++it) {
                const int pix = (ptid + it * 256) / C::YQ;
                const int y = oy0 + pix / TW, x = ox0 + pix % TW;
                const bool ok = pix < C::BM && y < a.H && x < a.W && y_cok;
                const unsigned off = ok ? (unsigned)(((n * a.H + y) * a.W + x) * a.yop.pstride + cyq) * 4u : 0x80000000u;
                yreg[it] = __builtin_amdgcn_raw_buffer_load_b128(yrs, off, 0, 0);
                yvalid |= (ok ? 1u : 0u) << it;
            }
#pragma unroll
            for (int it = 0; it < C::X_ITEMS; ++it) {
                const int pix = (ptid + it * 256) / C::XQ;
                const int iy = iy0 + pix / C::HW, ix = ix0 + pix % C::HW;
                const bool ok = pix < C::HP && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx && x_cok;
                const unsigned off = ok ? (unsigned)(((n * a.Hx + iy) * a.Wx + ix) * a.xop.pstride + cxq) * 4u : 0x80000000u;
                xreg[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
                xvalid |= (ok ? 1u : 0u) << it;
            }
        };
        // transform, split ONCE into (h, m, l) and write the planes of the thread's 4 channels (8 bytes each)
        auto put = [&](f32x4 v, unsigned char* dst) {
            if constexpr (P == 3) {
                unsigned h0, m0, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(dst + 64) = u32x2{m0, m1};
                *reinterpret_cast<u32x2*>(dst + 128) = u32x2{l0, l1};
            } else {                                  // bf16 operands: one rounding (RNE) per element at staging
                *reinterpret_cast<u32x2*>(dst) = u32x2{cvt_pair(v.x, v.y), cvt_pair(v.z, v.w)};
            }
        };
        auto xform = [&](f32x4 v, const f32x4& sc, const f32x4& sh, const InXform& xf, bool valid) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            if (xf.scale) {
                v = v * sc + sh;
                if (xf.relu) v = __builtin_elementwise_max(v, v * xf.slope);       // slope 0 = ReLU
                v = valid ? v : zero;                                              // zero padding AFTER the transform
            }
            return v;
        };
        auto store_tile = [&](int buf) {
            unsigned char* const sY = smem + buf * C::STAGE;
            unsigned char* const sX = sY + C::Y_BYTES;
#pragma unroll
            for (int it = 0; it < C::Y_ITEMS; ++it) {
                const int pix = (ptid + it * 256) / C::YQ;
                const f32x4 v = xform(__builtin_bit_cast(f32x4, yreg[it]), ysc, ysh, a.xf_y, (yvalid >> it) & 1u);
                if (pix < C::BM) put(v, sY + y_lds + pix * C::ROW);
            }
#pragma unroll
            for (int it = 0; it < C::X_ITEMS; ++it) {
                const int pix = (ptid + it * 256) / C::XQ;
                const f32x4 v = xform(__builtin_bit_cast(f32x4, xreg[it]), xsc, xsh, a.xf_x, (xvalid >> it) & 1u);
                if (pix < C::HP) put(v, sX + x_lds + pix * C::ROW);
            }
        };
        load_tile(split);
        // k = -1 is the prologue (tile 0); every later iteration stages tile k + 1 while the consumers multiply tile k
        for (int k = -1; k < my_tiles; ++k) {
            if (k + 1 < my_tiles) {
                store_tile((k + 1) & 1);                 // (waits for the loads of tile k + 1, issued an iteration ago)
                if (k + 2 < my_tiles) load_tile(split + (k + 2) * d.nsplit);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            wg_barrier();
        }
    } else {
        // =============================================================== consumers
        const int blk = wave % C::BLOCKS, ps = wave / C::BLOCKS;
        const int by = blk / BXB, bx = blk % BXB;
        // fragment addressing.  16-lane group g = lane >> 4: channels 16 (g & 1) .. + 15 of the 32-block, pixel half
        // h = g >> 1 (k = 8 h ..); inside the group lane 4 q + p supplies the address of block row q (pixel q of the 4),
        // channels 4 p .. 4 p + 3
        const int ll = lane & 15, q = ll >> 2, pc = ll & 3, gq = lane >> 4;
        const int lane_off = (gq & 1) * 32 + pc * 8;                  // bytes inside a plane row
        const int kh = gq >> 1;
        auto xpix = [&](int t) { return (t / TW) * C::HW + (t % TW); };          // halo pixel of tile pixel t (tap 0)
        f32x16 acc[C::NTAP];
#pragma unroll
        for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        wg_barrier();                                    // tile 0 is staged
        for (int k = 0; k < my_tiles; ++k) {
            const unsigned char* const sY = smem + (k & 1) * C::STAGE;
            const unsigned char* const yimg = sY + by * (C::BM * C::ROW) + lane_off;
            const unsigned char* const ximg = sY + C::Y_BYTES + bx * (C::HP * C::ROW) + lane_off;
            // software pipeline over (k-step, tap): the Xop fragment of the NEXT tap is read from LDS before the MFMAs of the
            // current one are issued; the fences pin "reads of the next tap, then MFMAs of this one"
            bf16x8 af[P], bfr[2][P];
            auto load_a = [&](int kk) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
#pragma unroll
                for (int p = 0; p < P; ++p) af[p] = tr_frag(yimg + t0 * C::ROW + p * 64, yimg + t1 * C::ROW + p * 64);
            };
            auto load_b = [&](int kk, int tap, bf16x8 (&bf)[P]) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
                const int x0 = xpix(t0), x1 = xpix(t1);
                const int toff = ((tap / R) * C::HW + (tap % R)) * C::ROW;
#pragma unroll
                for (int p = 0; p < P; ++p)
                    bf[p] = tr_frag(ximg + x0 * C::ROW + toff + p * 64, ximg + x1 * C::ROW + toff + p * 64);
            };
            load_b(0, 0, bfr[0]);
#pragma unroll
            for (int kk = 0; kk < C::KS_W; ++kk) {
                load_a(kk);
#pragma unroll
                for (int tap = 0; tap < C::NTAP; ++tap) {
                    constexpr int NT_ = C::NTAP;
                    const int cur = (kk * NT_ + tap) & 1;
                    if (tap + 1 < NT_) load_b(kk, tap + 1, bfr[cur ^ 1]);
                    else if (kk + 1 < C::KS_W) load_b(kk + 1, 0, bfr[cur ^ 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[tap] = mma<P>(af, bfr[cur], acc[tap]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            wg_barrier();                                // every consumer is done with this tile's buffers; tile k + 1 is staged
        }
        // ---- waves that split the tile's k-steps (WP > 1) add their accumulators through LDS, TC taps at a time.  The
        // producers are past their last barrier (they wrote nothing after it): the staging area is free
        if constexpr (C::WP > 1) {
            constexpr int TC = C::TC;
            float* s_red = reinterpret_cast<float*>(smem) + blk * TC * 1024;
#pragma unroll
            for (int t0 = 0; t0 < C::NTAP; t0 += TC) {
#pragma unroll
                for (int w = 1; w < C::WP; ++w) {
                    lds_barrier();
                    if (ps == w) {
#pragma unroll
                        for (int t = 0; t < TC; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) s_red[(t * 16 + r) * 64 + lane] = acc[t0 + t][r];
                    }
                    lds_barrier();
                    if (ps == 0) {
#pragma unroll
                        for (int t = 0; t < TC; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[t0 + t][r] += s_red[(t * 16 + r) * 64 + lane];
                    }
                }
            }
        }
        // ---- the workgroup's partial slab: rows (reg) = cy, cols (lane & 31) = cx
        if (ps == 0) {
            float* slab = a.slab + (size_t)split * d.slab_stride;
            const int li = lane & 31, lh = lane >> 5;
            const int cx = cx0 + bx * 32 + li;
#pragma unroll
            for (int tap = 0; tap < C::NTAP; ++tap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cy = cy0 + by * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (cy < a.Cy && cx < a.Cx)
                        slab[(int64_t)tap * a.tap_stride + (int64_t)cy * a.sy + (int64_t)cx * a.sx] = acc[tap][r];
                }
            }
        }
    }
}

struct Plan { int nsplit; int64_t slab_stride; };

template <int R, int BYB, int BXB, int TH, int TW>
Plan plan_cfg(const WgradArgs& a) {
    static const int wgs = getenv("RFI_WGRAD_WS_WGS") ? atoi(getenv("RFI_WGRAD_WS_WGS")) : 256;    // one workgroup per CU
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int chunks = (int)cdiv(a.Cy, 32 * BYB) * (int)cdiv(a.Cx, 32 * BXB);
    int nsplit = (int)cdiv(wgs, chunks);
    if (nsplit > ntiles) nsplit = ntiles;
    if (nsplit < 1) nsplit = 1;
    return Plan{nsplit, (int64_t)R * R * a.tap_stride};
}

template <int R, int BYB, int BXB, int TH, int TW, int P>
void launch_cfg(rfi_ctx* ctx, const WgradArgs& a) {
    using C = WWCfg<R, BYB, BXB, TH, TW, P>;
    const Plan p = plan_cfg<R, BYB, BXB, TH, TW>(a);
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)p.nsplit * p.slab_stride, "wgrad_ws: slab workspace too small");
    WWsDev d{a, p.nsplit, p.slab_stride, (unsigned)((int64_t)a.N * a.Hx * a.Wx * a.xop.pstride * 4),
             (unsigned)((int64_t)a.N * a.H * a.W * a.yop.pstride * 4)};
    dim3 grid(p.nsplit, (unsigned)cdiv(a.Cy, 32 * BYB), (unsigned)cdiv(a.Cx, 32 * BXB));
    const size_t lds = C::LDS_BYTES;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_kernel<R, BYB, BXB, TH, TW, P>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cy * a.Cx * R * R;
        std::string label;
        if (ctx->profiling)
            label = "wgrad_ws R" + std::to_string(R) + " N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" +
                    std::to_string(a.W) + " cx" + std::to_string(a.Cx) + " cy" + std::to_string(a.Cy) + " split" +
                    std::to_string(p.nsplit) + (P == 3 ? " 3xbf16" : " bf16");
        const double bytes = 4.0 * ((double)a.N * a.Hx * a.Wx * a.Cx + (double)a.N * a.H * a.W * a.Cy + (double)R * R * a.Cx * a.Cy);
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
        hipLaunchKernelGGL((wgrad_ws_kernel<R, BYB, BXB, TH, TW, P>), grid, dim3(512), lds, ctx->stream, d);
        check_launch("wgrad_ws");
    }
    launch_reduce_slabs(ctx, a.slab, p.nsplit, p.slab_stride, a.dw);
}

template <int R>
void select(rfi_ctx* ctx, const WgradArgs& a) {
    const bool y2 = a.Cy > 32, x2 = a.Cx > 32;
    const bool p1 = a.bf16 && !a.bf16x3;
#define RFI_WW(BYB_, BXB_, TH_, TW_)                                                \
    do {                                                                            \
        if (p1) launch_cfg<R, BYB_, BXB_, TH_, TW_, 1>(ctx, a);                     \
        else launch_cfg<R, BYB_, BXB_, TH_, TW_, 3>(ctx, a);                        \
        return;                                                                     \
    } while (0)
    if (y2 && x2) RFI_WW(2, 2, 8, 8);
    if (y2) RFI_WW(2, 1, 8, 8);
    if (x2) RFI_WW(1, 2, 8, 8);
    if (a.W >= 16) RFI_WW(1, 1, 8, 16);
    RFI_WW(1, 1, 16, 8);
#undef RFI_WW
}

}  // namespace

// the shapes of wgrad_split.hip; 32-bit byte offsets with 2^31 as the "reads zero" offset of the buffer loads
bool wgrad_ws_eligible(const WgradArgs& a) {
    if (a.Cx % 4 || a.Cy % 4 || a.xop.pstride % 4 || a.yop.pstride % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.xop.p) & 15) || (reinterpret_cast<uintptr_t>(a.yop.p) & 15)) return false;
    if ((int64_t)a.N * a.Hx * a.Wx * a.xop.pstride * 4 >= (int64_t)1 << 31 || (int64_t)a.N * a.H * a.W * a.yop.pstride * 4 >= (int64_t)1 << 31) return false;
    if (a.S != 1) return false;
    return (a.R == 3 && a.pad == 1) || (a.R == 2 && a.pad == 1) || (a.R == 1 && a.pad == 0);
}
void launch_wgrad_ws(rfi_ctx* ctx, const WgradArgs& a) {
    RFI_REQUIRE(wgrad_ws_eligible(a) && (a.bf16 || a.bf16x3), "wgrad_ws: shape or arithmetic not eligible");
    if (a.R == 3) select<3>(ctx, a);
    else if (a.R == 2) select<2>(ctx, a);
    else select<1>(ctx, a);
}

}  // namespace rfi
