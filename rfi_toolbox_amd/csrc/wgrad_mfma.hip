// Weight gradient of the conv-like contraction on the gfx950 matrix cores (fp32 MFMA 32x32x2).
//
//   dW[tap][cy][cx] = sum_{n,y,x} T(Yop)[n,y,x,cy] * T(Xop)[n, y*S+r-pad, x*S+s-pad, cx]
//
// GEMM view: the reduction runs over PIXELS (K = N*H*W, up to 10^6) and the output is tiny
// (taps*Cy*Cx), so the work is split over pixels: a workgroup is persistent over a strided set of
// TH x TW spatial tiles, keeps its taps x 32 x 32 accumulators in registers for its whole life and
// writes ONE partial slab at the end; a second launch sums the slabs in fixed order (bitwise
// reproducible, no float atomics).  Per tile the Yop tile and the Xop HALO tile are staged through
// registers into LDS as [pixel][channel] rows; MFMA operands are ds_read_b32 (lane = channel,
// k = pixel), so one A read feeds all taps.  Waves of a workgroup split either the (cy,cx) 32x32
// blocks of a BY x BX channel tile or, when that tile is a single block, the tile's pixels.
#include <algorithm>

#include "planes.hpp"

namespace rfi {

void launch_wgrad_direct(rfi_ctx* ctx, const WgradArgs& a);
size_t wgrad_direct_slab_floats(const WgradArgs& a);
// wgrad_split.hip: the 3 x bf16 arithmetic with the operands split once at staging time (3x3 layers)
bool wgrad_split_eligible(const WgradArgs& a);
size_t wgrad_split_slab_floats(const WgradArgs& a);
void launch_wgrad_split(rfi_ctx* ctx, const WgradArgs& a);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// float32 emulated by three bf16 pieces (see conv_mfma.hip): h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 { bf16x8 h, m, l; };
__device__ __forceinline__ unsigned cvt_pair(float a, float b) {
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
__device__ __forceinline__ Split3 split8(const float (&v)[8]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
    const u32x4 H = {h[0], h[1], h[2], h[3]}, M = {m[0], m[1], m[2], m[3]}, L = {l[0], l[1], l[2], l[3]};
    return Split3{__builtin_bit_cast(bf16x8, H), __builtin_bit_cast(bf16x8, M), __builtin_bit_cast(bf16x8, L)};
}
__device__ __forceinline__ f32x16 mfma_3xbf16(const Split3& a, const Split3& b, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ s16x4 pack_bf16(float a, float b, float c, float d) {   // RNE, 2 x v_cvt_pk_bf16_f32
    const f32x2 lo = {a, b}, hi = {c, d};
    u32x2 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
    return __builtin_bit_cast(s16x4, r);
}

template <int R, int S, int BY, int BX, int TH, int TW>
struct WCfg {
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH * S + R - S;
    static constexpr int HW = TW * S + R - S;
    static constexpr int HP = HH * HW;
    static constexpr int NTAP = R * R;
    static constexpr int BYP = BY + 4, BXP = BX + 4;
    static constexpr int BLOCKS = (BY / 32) * (BX / 32);   // 32x32 channel blocks per workgroup
    static constexpr int WP = 4 / BLOCKS;                   // waves splitting the tile's pixels
    static constexpr int KSTEPS = BM / 2 / WP;              // k-steps (2 pixels each) per wave per tile
    static constexpr int Y_ITEMS = (BM * (BY / 4) + 255) / 256;
    static constexpr int X_ITEMS = (HP * (BX / 4) + 255) / 256;
    static constexpr int TC = (NTAP % 3 == 0) ? 3 : NTAP;  // taps per cross-wave reduction round
    static constexpr int STAGE_FLOATS = BM * BYP + HP * BXP;
    static constexpr int RED_FLOATS = (WP > 1) ? BLOCKS * TC * 1024 : 0;
    static constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
    static_assert(BLOCKS == 1 || BLOCKS == 2 || BLOCKS == 4, "1, 2 or 4 channel blocks");
    static_assert((BM / 2) % WP == 0, "pixels must split evenly over waves");
};

struct WgradDev {
    WgradArgs a;
    int nsplit;          // workgroups along the pixel split
    int64_t slab_stride; // floats per slab
};

// BF: bf16 compute mode -- same fp32 tiles in LDS; a lane gathers its channel's values of 4 consecutive
// pixels (lane half lh: pixels 4 lh .. 4 lh + 3 of an 8-pixel tile row), rounds them to bf16 and ONE
// v_mfma_f32_32x32x8_bf16 per tap (fp32 accumulate) covers the 8 pixels four fp32 k-steps covered.
// PREC: 0 float32 MFMA, 1 bf16 operands, 2 float32 emulated by 3 x bf16 pieces (lane half lh takes one whole
// 8-pixel tile row; six v_mfma_f32_32x32x16_bf16 per tap and 16 pixels)
template <int R, int S, int BY, int BX, int TH, int TW, int PREC>
__global__ __launch_bounds__(256) void wgrad_igemm_kernel(WgradDev d) {
    using C = WCfg<R, S, BY, BX, TH, TW>;
    const WgradArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_y = smem;
    float* s_x = s_y + C::BM * C::BYP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int blk = wave % C::BLOCKS, ps = wave / C::BLOCKS;
    const int by = blk / (BX / 32), bx = blk % (BX / 32);
    const int cy0 = blockIdx.y * BY, cx0 = blockIdx.z * BX;
    const int split = blockIdx.x;

    // ---- staging descriptors.  Item `it` of a thread is float4 number tid + it*256 of the tile;
    // 256 is a multiple of BY/4 and BX/4, so ALL items of a thread carry the same 4 channels: one
    // scale/shift pair per operand lives in registers for the whole kernel.
    constexpr int YQ = BY / 4, XQ = BX / 4;
    const int yq = tid % YQ, xq = tid % XQ;
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
    // tile-independent pixel coordinates of every item (packed row<<16 | col)
    int y_rc[C::Y_ITEMS], x_rc[C::X_ITEMS];
#pragma unroll
    for (int it = 0; it < C::Y_ITEMS; ++it) {
        const int pix = (tid + it * 256) / YQ;
        y_rc[it] = ((pix / TW) << 16) | (pix % TW);
    }
#pragma unroll
    for (int it = 0; it < C::X_ITEMS; ++it) {
        const int pix = (tid + it * 256) / XQ;
        x_rc[it] = ((pix / C::HW) << 16) | (pix % C::HW);
    }
    const int y_lds0 = (tid / YQ) * C::BYP + yq * 4, x_lds0 = (tid / XQ) * C::BXP + xq * 4;

    f32x4 yreg[C::Y_ITEMS], xreg[C::X_ITEMS];
    unsigned yvalid = 0, xvalid = 0;

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;

    // loads are UNCONDITIONAL (out-of-range items read offset 0 and are zeroed when written to LDS):
    // the compiler issues them back to back and they stay in flight under the MFMAs
    auto load_tile = [&](int tile) {
        const int tx_i = tile % tiles_x;
        const int ty_i = (tile / tiles_x) % tiles_y;
        const int n = tile / (tiles_x * tiles_y);
        const int oy0 = ty_i * TH, ox0 = tx_i * TW;
        const int ybase = (n * a.H + oy0) * a.W + ox0;                     // pixel index of the tile origin
        const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
        yvalid = 0;
        xvalid = 0;
#pragma unroll
        for (int it = 0; it < C::Y_ITEMS; ++it) {
            const int r = y_rc[it] >> 16, c = y_rc[it] & 0xffff;
            const bool ok = (tid + it * 256 < C::BM * YQ) && oy0 + r < a.H && ox0 + c < a.W && y_cok;
            const int off = ok ? (ybase + r * a.W + c) * a.yop.pstride + cyq : 0;
            yreg[it] = *reinterpret_cast<const f32x4*>(a.yop.p + off);
            yvalid |= (ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const int iy = iy0 + (x_rc[it] >> 16), ix = ix0 + (x_rc[it] & 0xffff);
            const bool ok = (tid + it * 256 < C::HP * XQ) && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx && x_cok;
            const int off = ok ? ((n * a.Hx + iy) * a.Wx + ix) * a.xop.pstride + cxq : 0;
            xreg[it] = *reinterpret_cast<const f32x4*>(a.xop.p + off);
            xvalid |= (ok ? 1u : 0u) << it;
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
            if ((it + 1) * 256 <= C::BM * YQ || tid + it * 256 < C::BM * YQ)
                *reinterpret_cast<f32x4*>(s_y + y_lds0 + it * (256 / YQ) * C::BYP) = v;
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            f32x4 v = xreg[it];
            if (a.xf_x.scale) {
                v = v * xsc + xsh;
                if (a.xf_x.relu) v = __builtin_elementwise_max(v, v * a.xf_x.slope);   // slope 0 = ReLU
            }
            v = ((xvalid >> it) & 1u) ? v : zero;
            if ((it + 1) * 256 <= C::HP * XQ || tid + it * 256 < C::HP * XQ)
                *reinterpret_cast<f32x4*>(s_x + x_lds0 + it * (256 / XQ) * C::BXP) = v;
        }
    };

    f32x16 acc[C::NTAP];
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // software pipeline over this workgroup's tiles: tile t+1's loads fly under tile t's MFMAs
    const int my_tiles = split < ntiles ? (ntiles - split + d.nsplit - 1) / d.nsplit : 0;
    if (my_tiles > 0) load_tile(split);
    for (int k = 0; k < my_tiles; ++k) {
        store_tile();
        __syncthreads();
        const int next = split + (k + 1 < my_tiles ? k + 1 : k) * d.nsplit;   // last tile re-read once
        load_tile(next);
        // ---- MFMA over this wave's share of the tile's pixels.  Operands of k-step ks+1 are read
        // from LDS into a second register set before the MFMAs of k-step ks are issued, so the
        // matrix pipe never waits on an LDS round trip (one wave per SIMD here: nobody else hides it)
        // Every operand address is `per-lane base + compile-time offset` (the k-loop is fully unrolled and
        // a wave's pixel range starts on a tile row), so the 10 ds_reads of a k-step carry immediate
        // offsets and cost no VALU work: they issue inside the 16 quad-cycles the last MFMA of the
        // previous k-step still occupies the pipe.
        static_assert((C::BM / C::WP) % TW == 0 && TW % 2 == 0, "a wave's pixels must start on a tile row");
        const int pw0 = ps * C::KSTEPS * 2;                          // first pixel of this wave
        if constexpr (PREC == 2) {
            static_assert(TW == 8 && (C::KSTEPS % 8) == 0, "3 x bf16 path: pairs of 8-pixel tile rows per wave");
            const float* yr = s_y + (pw0 + lh * 8) * C::BYP + by * 32 + li;
            const float* xr0 = s_x + (((pw0 / TW + lh) * S) * C::HW) * C::BXP + bx * 32 + li;
#pragma unroll
            for (int g = 0; g < C::KSTEPS / 8; ++g) {                  // two tile rows (16 pixels) per group
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = yr[(g * 16 + t) * C::BYP];
                const Split3 a3 = split8(v);
                const float* xr = xr0 + (g * 2 * S) * C::HW * C::BXP;
#pragma unroll
                for (int tap = 0; tap < C::NTAP; ++tap) {
                    const float* xp = xr + ((tap / R) * C::HW + (tap % R)) * C::BXP;
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = xp[t * S * C::BXP];
                    acc[tap] = mfma_3xbf16(a3, split8(v), acc[tap]);
                }
            }
            __syncthreads();
            continue;
        }
        if constexpr (PREC == 1) {
            static_assert(TW == 8 && (C::KSTEPS % 4) == 0, "bf16 path: 8-pixel tile rows, whole rows per wave");
            const float* ya4 = s_y + (pw0 + lh * 4) * C::BYP + by * 32 + li;
            const float* xb4 = s_x + (((pw0 / TW) * S) * C::HW + lh * 4 * S) * C::BXP + bx * 32 + li;
#pragma unroll
            for (int g = 0; g < C::KSTEPS / 4; ++g) {                  // one 8-pixel tile row per group
                const float* yp = ya4 + g * 8 * C::BYP;
                const s16x4 a4 = pack_bf16(yp[0], yp[C::BYP], yp[2 * C::BYP], yp[3 * C::BYP]);
                const float* xr = xb4 + (g * S) * C::HW * C::BXP;
#pragma unroll
                for (int tap = 0; tap < C::NTAP; ++tap) {
                    const float* xp = xr + ((tap / R) * C::HW + (tap % R)) * C::BXP;
                    const s16x4 b4 = pack_bf16(xp[0], xp[S * C::BXP], xp[2 * S * C::BXP], xp[3 * S * C::BXP]);
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[tap], 0, 0, 0);
                }
            }
            __syncthreads();
            continue;
        }
        const float* ya = s_y + (pw0 + lh) * C::BYP + by * 32 + li;
        const float* xb0 = s_x + (((pw0 / TW) * S) * C::HW + lh * S) * C::BXP + bx * 32 + li;
        float av[2], bv[2][C::NTAP];
        auto lds_operands = [&](int ks, float& a_, float (&b_)[C::NTAP]) {
            const int pk = ks * 2;                                     // pixel pair of this k-step (lane half adds 1)
            a_ = ya[pk * C::BYP];
            const float* xb = xb0 + (((pk / TW) * S) * C::HW + (pk % TW) * S) * C::BXP;
#pragma unroll
            for (int tap = 0; tap < C::NTAP; ++tap) b_[tap] = xb[((tap / R) * C::HW + (tap % R)) * C::BXP];
        };
        lds_operands(0, av[0], bv[0]);
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < C::KSTEPS) lds_operands(ks + 1, av[cur ^ 1], bv[cur ^ 1]);
            // hipcc otherwise sinks each ds_read to just before its MFMA (lgkmcnt(0) per MFMA);
            // the fences keep "reads of step ks+1, then MFMAs of step ks" in program order
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < C::NTAP; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][tap], acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- waves that split the tile's pixels (WP > 1) first add their accumulators together through
    // LDS (the staging area is free now: the tile loop ended on a barrier), TC taps at a time, so a
    // workgroup always emits exactly one partial slab
    if constexpr (C::WP > 1) {
        constexpr int TC = C::TC;                                  // 3 x 4 KB (R=3) or 4 x 4 KB (R=2) per block
        float* s_red = smem + blk * TC * 1024;
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
    // ---- write the workgroup's partial: rows (reg) = cy, cols (lane&31) = cx
    if (ps == 0) {
        float* slab = a.slab + (size_t)split * d.slab_stride;
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

struct Plan {
    int nsplit, wp;
    int64_t slab_stride;
};

template <int R, int S, int BY, int BX, int TH, int TW>
Plan plan_cfg(const WgradArgs& a, int cus) {
    using C = WCfg<R, S, BY, BX, TH, TW>;
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int chunks = (int)cdiv(a.Cy, BY) * (int)cdiv(a.Cx, BX);
    // 144 accumulator + ~170 working registers per lane: ONE workgroup per CU is resident, so one
    // workgroup per CU in total (a second round would only repeat the prologue/epilogue)
    int nsplit = (int)cdiv((int64_t)cus, chunks);
    if (nsplit > ntiles) nsplit = ntiles;
    if (nsplit < 1) nsplit = 1;
    Plan p;
    p.nsplit = nsplit;
    p.wp = 1;                     // pixel-split waves are summed inside the workgroup
    p.slab_stride = (int64_t)C::NTAP * a.tap_stride;
    return p;
}

template <int R, int S, int BY, int BX, int TH, int TW, int PREC = 0>
void launch_cfg(rfi_ctx* ctx, const WgradArgs& a) {
    if constexpr (PREC == 0) {
        if (a.bf16) return launch_cfg<R, S, BY, BX, TH, TW, 1>(ctx, a);
        if constexpr (WCfg<R, S, BY, BX, TH, TW>::KSTEPS % 8 == 0) {
            if (a.bf16x3) return launch_cfg<R, S, BY, BX, TH, TW, 2>(ctx, a);
        }
    }
    using C = WCfg<R, S, BY, BX, TH, TW>;
    const Plan p = plan_cfg<R, S, BY, BX, TH, TW>(a, 256);
    const int nslabs = p.nsplit * p.wp;
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)nslabs * p.slab_stride, "wgrad: slab workspace too small");
    WgradDev d{a, p.nsplit, p.slab_stride};
    dim3 grid(p.nsplit, (unsigned)cdiv(a.Cy, BY), (unsigned)cdiv(a.Cx, BX));
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&wgrad_igemm_kernel<R, S, BY, BX, TH, TW, PREC>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cy * a.Cx * R * R;
        std::string label;
        if (ctx->profiling)
            label = "wgrad R" + std::to_string(R) + " N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" +
                    std::to_string(a.W) + " cx" + std::to_string(a.Cx) + " cy" + std::to_string(a.Cy) + " split" +
                    std::to_string(p.nsplit) + (PREC == 1 ? " bf16" : (PREC == 2 ? " 3xbf16" : ""));
        const double bytes = 4.0 * ((double)a.N * a.Hx * a.Wx * a.Cx + (double)a.N * a.H * a.W * a.Cy + (double)R * R * a.Cx * a.Cy);
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
        hipLaunchKernelGGL((wgrad_igemm_kernel<R, S, BY, BX, TH, TW, PREC>), grid, dim3(256), lds, ctx->stream, d);
        check_launch("wgrad_igemm");
    }
    launch_reduce_slabs(ctx, a.slab, nslabs, p.slab_stride, a.dw);
}

enum { SEL_PLAN = 0, SEL_LAUNCH = 1 };

template <int R, int S, int TH, int TW>
Plan select(rfi_ctx* ctx, const WgradArgs& a, int what, int cus) {
    // channel tile: 64 where the dimension has more than 32 channels, else 32
    const bool y64 = a.Cy > 32, x64 = a.Cx > 32;
#define RFI_WG(BY_, BX_)                                                        \
    do {                                                                        \
        if (what == SEL_LAUNCH) launch_cfg<R, S, BY_, BX_, TH, TW>(ctx, a);     \
        return plan_cfg<R, S, BY_, BX_, TH, TW>(a, cus);                        \
    } while (0)
    if (y64 && x64) RFI_WG(64, 64);
    if (y64) RFI_WG(64, 32);
    if (x64) RFI_WG(32, 64);
    RFI_WG(32, 32);
#undef RFI_WG
}

bool wgrad_mfma_eligible(const WgradArgs& a) {
    if (a.Cx % 4 || a.Cy % 4 || a.xop.pstride % 4 || a.yop.pstride % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.xop.p) & 15) || (reinterpret_cast<uintptr_t>(a.yop.p) & 15)) return false;
    return (a.R == 3 && a.S == 1 && a.pad == 1) || (a.R == 2 && a.S == 2 && a.pad == 0);
}

Plan dispatch(rfi_ctx* ctx, const WgradArgs& a, int what, int cus) {
    if (a.R == 3) {
        // channel tiles with fewer than 4 blocks split the tile's PIXELS over the waves: give them a
        // 16x8 tile so each wave still has 16+ k-steps between barriers
        const bool splits_pixels = !(a.Cy > 32 && a.Cx > 32);
        if (splits_pixels && a.H >= 16) return select<3, 1, 16, 8>(ctx, a, what, cus);
        return select<3, 1, 8, 8>(ctx, a, what, cus);
    }
    return select<2, 2, 4, 8>(ctx, a, what, cus);
}

}  // namespace

size_t wgrad_slab_floats(const WgradArgs& a, int impl) {
    size_t need = wgrad_direct_slab_floats(a);
    if ((impl == IMPL_PLANES_X3 || impl == IMPL_PLANES_BF16) && a.R == 3 && a.S == 1)
        return std::max(need, pwgrad_slab_floats_f32(a));
    if (impl == IMPL_MFMA_BF16 || impl == IMPL_MFMA_BF16X3) impl = IMPL_MFMA;
    if (impl != IMPL_DIRECT && wgrad_split_eligible(a)) need = std::max(need, wgrad_split_slab_floats(a));
    if (impl != IMPL_DIRECT) need = std::max(need, wgrad_ws_slab_floats(a));      // (the transposed conv's shape: its own plan)
    if (impl != IMPL_DIRECT && wgrad_mfma_eligible(a)) {
        const Plan p = dispatch(nullptr, a, SEL_PLAN, 256);
        // plan with the largest CU count we may meet so the workspace always suffices
        const size_t m = (size_t)p.nsplit * p.wp * p.slab_stride;
        if (m > need) need = m;
    }
    return need;
}

void launch_wgrad(rfi_ctx* ctx, const WgradArgs& a_in, int impl) {
    WgradArgs a = a_in;
    if (impl == IMPL_PLANES_X3 || impl == IMPL_PLANES_BF16) {      // plane kernel through temporary plane copies
        RFI_REQUIRE(a.R == 3 && a.S == 1, "wgrad: the plane kernel covers 3x3 stride-1 convolutions");
        launch_pwgrad_from_f32(ctx, a, impl == IMPL_PLANES_X3 ? 3 : 1);
        return;
    }
    if (impl == IMPL_MFMA_BF16) {
        a.bf16 = true;
        impl = IMPL_MFMA;
    }
    if (impl == IMPL_MFMA_BF16X3) {
        a.bf16x3 = true;
        impl = IMPL_MFMA;
    }
    RFI_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cx > 0 && a.Cy > 0, "wgrad: empty shape");
    RFI_REQUIRE((int64_t)a.N * a.Hx * a.Wx * a.xop.pstride < (int64_t)1 << 31 &&
                    (int64_t)a.N * a.H * a.W * a.yop.pstride < (int64_t)1 << 31,
                "wgrad: tensor too large for 32-bit element offsets");
    const bool ok = wgrad_mfma_eligible(a) || ((a.bf16 || a.bf16x3) && wgrad_split_eligible(a));
    if (impl == IMPL_MFMA) RFI_REQUIRE(ok, "wgrad: shape/alignment not eligible for the MFMA kernel");
    if (impl == IMPL_DIRECT || !ok) {
        launch_wgrad_direct(ctx, a);
        return;
    }
    static const bool no_ws = getenv("RFI_NO_WGRAD_WS") != nullptr;       // A/B runs: round 2's kernels
    static const bool no_stem = getenv("RFI_NO_STEM") != nullptr;
    if (!no_ws && !no_stem && a.bf16x3 && wgrad_stem_eligible(a)) {
        launch_wgrad_stem(ctx, a);    // Cx = 4: (tap, channel) packed into the GEMM's N (wgrad_stem.hip)
        return;
    }
    if (!no_ws && (a.bf16x3 || a.bf16) && wgrad_ws_eligible(a) && (wgrad_split_eligible(a) || (a.R == 2 && a.S == 2))) {
        launch_wgrad_ws(ctx, a);      // (its slab plan -- 256 workgroups -- fits inside wgrad_split's, which sized the workspace)
        return;
    }
    static const bool old_x3 = getenv("RFI_OLD_WGRAD") != nullptr;       // round 1's split-per-fragment kernel (A/B runs)
    // R = 2 / S = 1 and R = 1 (ResNet-style encoder) exist only in the split-at-staging kernel (P = 1: bf16 mode)
    const bool only_split = (a.R == 2 && a.S == 1) || a.R == 1;
    if (only_split) RFI_REQUIRE((a.bf16 || a.bf16x3) && wgrad_split_eligible(a),
                                "wgrad: 2x2 stride-1 and 1x1 weight gradients need the bf16 or 3 x bf16 arithmetic and 4-channel alignment");
    if (((a.bf16x3 && !old_x3) || only_split) && wgrad_split_eligible(a)) {
        launch_wgrad_split(ctx, a);
        return;
    }
    dispatch(ctx, a, SEL_LAUNCH, 256);   // MI355X: 256 CUs (the plan fixes the slab workspace size)
}

}  // namespace rfi
