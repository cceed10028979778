// Weight gradient of the 3x3 convolutions in the float32-by-3xbf16 arithmetic for FLOAT32 operand tensors (the
// default arithmetic of the models: activations and gradients live in HBM as float32).
//
//   dW[tap][cy][cx] = sum_{n,y,x} T(Yop)[n,y,x,cy] * T(Xop)[n, y+r-1, x+s-1, cx]
//
// Same GEMM view, workgroup tiling, accumulators and slab reduction as wgrad_planes.hip.  What differs is the
// staging: a tile of each operand is loaded as float32 (16 bytes per lane), the producing layer's BatchNorm-apply +
// activation is applied in flight (InXform), every element is split ONCE into its three bf16 pieces (h, m, l) and
// the pieces are written to LDS as [32-channel block][pixel][plane][32 channels] images; the fragments ("8
// consecutive pixels of one channel per lane") then come out of ds_read_b64_tr_b16 with no further VALU work.
// Round 1's kernel read float32 from LDS and split every FRAGMENT in registers -- one operand fragment per tap per
// k-step per wave, ten times the conversion work of splitting at staging time; its matrix pipe was busy 40 % of
// the cycles.
#include <algorithm>

#include "planes.hpp"

namespace rfi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ unsigned cvt_pair(float a, float b) {       // RNE, v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pair(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pair(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pair(sa, sb);
}

struct SWgradDev {
    WgradArgs a;
    int nsplit;
    int64_t slab_stride;
};

template <int R, int S, int BYB, int BXB, int TH, int TW, int P_>
struct SWCfg {
    static constexpr int P = P_;
    static constexpr int NT = 256;
    static constexpr int NTAP = R * R;
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH * S + R - S, HW = TW * S + R - S, HP = HH * HW;
    static constexpr int BLOCKS = BYB * BXB;
    static constexpr int WP = 4 / BLOCKS;
    static constexpr int KS = BM / 16;
    static constexpr int KS_W = KS / WP;
    static constexpr int ROW = P * 64;
    static constexpr int YQ = BYB * 8, XQ = BXB * 8;           // float4 groups per pixel
    static constexpr int Y_ITEMS = (BM * YQ + NT - 1) / NT, X_ITEMS = (HP * XQ + NT - 1) / NT;
    static constexpr int Y_BYTES = BYB * BM * ROW, X_BYTES = BXB * HP * ROW;
    static constexpr int TC = (NTAP % 3 == 0) ? 3 : NTAP;
    static constexpr int RED_BYTES = (WP > 1) ? BLOCKS * TC * 4096 : 0;
    static constexpr int LDS_BYTES = (Y_BYTES + X_BYTES) > RED_BYTES ? (Y_BYTES + X_BYTES) : RED_BYTES;
    static_assert(BLOCKS == 1 || BLOCKS == 2 || BLOCKS == 4, "1, 2 or 4 channel blocks");
    static_assert(BM % 16 == 0 && KS % WP == 0 && TW % 4 == 0, "tile must split into k-steps of 16 pixels");
    static_assert(NT % YQ == 0 && NT % XQ == 0, "a thread keeps one channel group for all its items");
};

template <int P>
__device__ __forceinline__ f32x16 mma(const bf16x8 (&a)[P], const bf16x8 (&b)[P], f32x16 acc) {
    if constexpr (P == 3) {            // pieces: [0] = h, [1] = m, [2] = l; small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// one operand fragment: this lane's 8 pixels (k = 8 h + 0..7) of its channel, from two transposing reads of
// 4 pixel rows each.  `p0` / `p1`: byte addresses of THIS lane's row of the two 4x16 blocks
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}


template <int R, int S, int BYB, int BXB, int TH, int TW, int P>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(SWgradDev d) {
    using C = SWCfg<R, S, BYB, BXB, TH, TW, P>;
    const WgradArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sY = smem;
    unsigned char* const sX = smem + C::Y_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave % C::BLOCKS, ps = wave / C::BLOCKS;
    const int by = blk / BXB, bx = blk % BXB;
    const int cy0 = blockIdx.y * BYB * 32, cx0 = blockIdx.z * BXB * 32;
    const int split = blockIdx.x;

    // ---- staging descriptors: item `it` of a thread is float4 number tid + it*256 of an operand tile; all items of a
    // thread carry the same 4 channels, so ONE scale/shift pair per operand lives in registers for the whole kernel
    const int yq = tid % C::YQ, xq = tid % C::XQ;
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
    // LDS byte offset of this thread's 4 channels inside a pixel row of its 32-channel block image
    const int y_lds = (yq >> 3) * (C::BM * C::ROW) + (yq & 7) * 8, x_lds = (xq >> 3) * (C::HP * C::ROW) + (xq & 7) * 8;

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    f32x4 yreg[C::Y_ITEMS], xreg[C::X_ITEMS];
    unsigned yvalid = 0, xvalid = 0;
    // loads are UNCONDITIONAL (out-of-range items read offset 0 and are zeroed when written to LDS)
    auto load_tile = [&](int tile) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = ty_i * TH, ox0 = tx_i * TW;
        const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
        yvalid = 0;
        xvalid = 0;
#pragma unroll
        for (int it = 0; it < C::Y_ITEMS; ++it) {
            const int pix = (tid + it * 256) / C::YQ;
            const int y = oy0 + pix / TW, x = ox0 + pix % TW;
            const bool ok = pix < C::BM && y < a.H && x < a.W && y_cok;
            const int off = ok ? ((n * a.H + y) * a.W + x) * a.yop.pstride + cyq : 0;
            yreg[it] = *reinterpret_cast<const f32x4*>(a.yop.p + off);
            yvalid |= (ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const int pix = (tid + it * 256) / C::XQ;
            const int iy = iy0 + pix / C::HW, ix = ix0 + pix % C::HW;
            const bool ok = pix < C::HP && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx && x_cok;
            const int off = ok ? ((n * a.Hx + iy) * a.Wx + ix) * a.xop.pstride + cxq : 0;
            xreg[it] = *reinterpret_cast<const f32x4*>(a.xop.p + off);
            xvalid |= (ok ? 1u : 0u) << it;
        }
    };
    // transform, split ONCE into (h, m, l) and write the three planes of the thread's 4 channels (8 bytes each)
    auto put = [&](f32x4 v, unsigned char* dst) {
        if constexpr (P == 3) {
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v.x, v.y, h0, m0, l0);
            split_pair(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dst + 64) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(dst + 128) = u32x2{l0, l1};
        } else {                                  // bf16 compute mode: one rounding (RNE) per element at staging
            *reinterpret_cast<u32x2*>(dst) = u32x2{cvt_pair(v.x, v.y), cvt_pair(v.z, v.w)};
        }
    };
    auto store_tile = [&]() {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < C::Y_ITEMS; ++it) {
            f32x4 v = yreg[it];
            if (a.xf_y.scale) {
                v = v * ysc + ysh;
                if (a.xf_y.relu) v = __builtin_elementwise_max(v, v * a.xf_y.slope);   // slope 0 = ReLU
            }
            v = ((yvalid >> it) & 1u) ? v : zero;
            const int pix = (tid + it * 256) / C::YQ;
            if (pix < C::BM) put(v, sY + y_lds + pix * C::ROW);
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            f32x4 v = xreg[it];
            if (a.xf_x.scale) {
                v = v * xsc + xsh;
                if (a.xf_x.relu) v = __builtin_elementwise_max(v, v * a.xf_x.slope);
            }
            v = ((xvalid >> it) & 1u) ? v : zero;
            const int pix = (tid + it * 256) / C::XQ;
            if (pix < C::HP) put(v, sX + x_lds + pix * C::ROW);
        }
    };

    // ---- fragment addressing.  16-lane group g = lane >> 4: channels 16 (g & 1) .. + 15 of the 32-block, pixel
    // half h = g >> 1 (k = 8 h ..); inside the group lane 4 q + p supplies the address of block row q (pixel q of
    // the 4), channels 4 p .. 4 p + 3.
    const int ll = lane & 15, q = ll >> 2, pc = ll & 3, gq = lane >> 4;
    const int lane_off = (gq & 1) * 32 + pc * 8;                  // bytes inside a plane row
    const int kh = gq >> 1;
    // tile pixel of (k-step ks, read s in {0,1}) for this lane: t = ks * 16 + 8 kh + 4 s + q
    auto ypix = [&](int t) { return t; };                                            // Y image is [pixel of the tile]
    auto xpix = [&](int t) { return ((t / TW) * S) * C::HW + (t % TW) * S; };         // halo pixel of tile pixel t (tap 0)

    f32x16 acc[C::NTAP];
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const unsigned char* const yimg = sY + by * (C::BM * C::ROW) + lane_off;
    const unsigned char* const ximg = sX + bx * (C::HP * C::ROW) + lane_off;
    const int my_tiles = split < ntiles ? (ntiles - split + d.nsplit - 1) / d.nsplit : 0;
    for (int k = 0; k < my_tiles; ++k) {
        // synchronous staging: the tile's loads are not kept in flight under the previous tile's MFMAs (44 more
        // registers per lane on top of 144 accumulators would spill); the CU's other workgroup computes meanwhile
        load_tile(split + k * d.nsplit);
        store_tile();
        __syncthreads();
        // software pipeline over (k-step, tap): the Xop fragment of the NEXT tap is read from LDS before the MFMAs of
        // the current one are issued (hipcc otherwise sinks each transposing read to just before its use and the matrix
        // pipe idles for an LDS round trip per tap); the fences pin "reads of the next tap, then MFMAs of this one"
        {
            bf16x8 af[P], bfr[2][P];
            auto load_a = [&](int kk) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
#pragma unroll
                for (int p = 0; p < P; ++p)
                    af[p] = tr_frag(yimg + ypix(t0) * C::ROW + p * 64, yimg + ypix(t1) * C::ROW + p * 64);
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
        }
        __syncthreads();                             // everybody is done reading before the next tile is written
    }

    // ---- waves that split the tile's k-steps (WP > 1) add their accumulators through LDS, TC taps at a time
    if constexpr (C::WP > 1) {
        constexpr int TC = C::TC;
        float* s_red = reinterpret_cast<float*>(smem) + blk * TC * 1024;
#pragma unroll
        for (int t0 = 0; t0 < C::NTAP; t0 += TC) {
#pragma unroll
            for (int w = 1; w < C::WP; ++w) {
                __syncthreads();
                if (ps == w) {
#pragma unroll
                    for (int t = 0; t < TC; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s_red[(t * 16 + r) * 64 + lane] = acc[t0 + t][r];
                }
                __syncthreads();
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

struct Plan { int nsplit; int64_t slab_stride; };

template <int R, int S, int BYB, int BXB, int TH, int TW>
Plan plan_cfg(const WgradArgs& a) {
    static const int wgs = getenv("RFI_WGRAD_WGS") ? atoi(getenv("RFI_WGRAD_WGS")) : 512;    // tuning experiments
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int chunks = (int)cdiv(a.Cy, 32 * BYB) * (int)cdiv(a.Cx, 32 * BXB);
    int nsplit = (int)cdiv(wgs, chunks);             // two workgroups per CU in total
    if (nsplit > ntiles) nsplit = ntiles;
    if (nsplit < 1) nsplit = 1;
    return Plan{nsplit, (int64_t)R * R * a.tap_stride};
}

template <int R, int S, int BYB, int BXB, int TH, int TW, int P>
void launch_cfg(rfi_ctx* ctx, const WgradArgs& a) {
    using C = SWCfg<R, S, BYB, BXB, TH, TW, P>;
    const Plan p = plan_cfg<R, S, BYB, BXB, TH, TW>(a);
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)p.nsplit * p.slab_stride, "wgrad: slab workspace too small");
    SWgradDev d{a, p.nsplit, p.slab_stride};
    dim3 grid(p.nsplit, (unsigned)cdiv(a.Cy, 32 * BYB), (unsigned)cdiv(a.Cx, 32 * BXB));
    const size_t lds = C::LDS_BYTES;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_kernel<R, S, BYB, BXB, TH, TW, P>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cy * a.Cx * R * R;
        std::string label;
        if (ctx->profiling)
            label = "wgrad R" + std::to_string(R) + " N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" +
                    std::to_string(a.W) + " cx" + std::to_string(a.Cx) + " cy" + std::to_string(a.Cy) + " split" +
                    std::to_string(p.nsplit) + (P == 3 ? " 3xbf16" : " bf16") + " split-at-staging";
        const double bytes = 4.0 * ((double)a.N * a.Hx * a.Wx * a.Cx + (double)a.N * a.H * a.W * a.Cy + (double)R * R * a.Cx * a.Cy);
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
        hipLaunchKernelGGL((wgrad_split_kernel<R, S, BYB, BXB, TH, TW, P>), grid, dim3(256), lds, ctx->stream, d);
        check_launch("wgrad_split");
    }
    launch_reduce_slabs(ctx, a.slab, p.nsplit, p.slab_stride, a.dw);
}

template <int R, int S>
Plan select(rfi_ctx* ctx, const WgradArgs& a, bool launch) {
    const bool y2 = a.Cy > 32, x2 = a.Cx > 32;
    const bool p1 = a.bf16 && !a.bf16x3;
#define RFI_SW(BYB_, BXB_, TH_, TW_)                                                  \
    do {                                                                              \
        if (launch) {                                                                 \
            if (p1) launch_cfg<R, S, BYB_, BXB_, TH_, TW_, 1>(ctx, a);                \
            else launch_cfg<R, S, BYB_, BXB_, TH_, TW_, 3>(ctx, a);                   \
        }                                                                             \
        return plan_cfg<R, S, BYB_, BXB_, TH_, TW_>(a);                               \
    } while (0)
    if (y2 && x2) RFI_SW(2, 2, 8, 8);
    if (y2) RFI_SW(2, 1, 8, 8);
    if (x2) RFI_SW(1, 2, 8, 8);
    if (a.W >= 16) RFI_SW(1, 1, 8, 16);
    RFI_SW(1, 1, 16, 8);
#undef RFI_SW
}
Plan select_r(rfi_ctx* ctx, const WgradArgs& a, bool launch) {
    if (a.R == 3) return select<3, 1>(ctx, a, launch);
    if (a.R == 2) return select<2, 1>(ctx, a, launch);
    return select<1, 1>(ctx, a, launch);
}

}  // namespace

// 3x3 (pad 1), the 2x2 (pad 1) form of a stride-2 3x3 conv on its space-to-depth input, and 1x1 convolutions
bool wgrad_split_eligible(const WgradArgs& a) {
    if (a.Cx % 4 || a.Cy % 4 || a.xop.pstride % 4 || a.yop.pstride % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.xop.p) & 15) || (reinterpret_cast<uintptr_t>(a.yop.p) & 15)) return false;
    return (a.R == 3 && a.S == 1 && a.pad == 1) || (a.R == 2 && a.S == 1 && a.pad == 1) || (a.R == 1 && a.S == 1 && a.pad == 0);
}
size_t wgrad_split_slab_floats(const WgradArgs& a) {
    const Plan p = select_r(nullptr, a, false);
    return (size_t)p.nsplit * p.slab_stride;
}
void launch_wgrad_split(rfi_ctx* ctx, const WgradArgs& a) { select_r(ctx, a, true); }

}  // namespace rfi
