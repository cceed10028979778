// Weight gradient of the conv-like contraction on PLANE tensors (planes.hpp) for the gfx950 matrix cores.
//
//   dW[tap][cy][cx] = sum_{n,y,x} Yop[n,y,x,cy] * Xop[n, y*S+r-pad, x*S+s-pad, cx]
//
// GEMM view: the reduction runs over PIXELS (K = N*H*W), the output is tiny (taps*Cy*Cx).  A workgroup owns a
// BYB x BXB grid of 32x32 channel blocks and is persistent over a strided set of TH x TW spatial tiles; every
// wave keeps its taps x 32 x 32 accumulators in registers for the whole kernel and the workgroup writes ONE
// partial slab at the end (summed in fixed order by reduce_slabs: bitwise reproducible, no float atomics).
//
// Both operands need "8 consecutive PIXELS of one channel" per lane (k = pixel), i.e. the transpose of the
// NHWC plane layout.  Nothing is transposed by hand: the Yop tile and the Xop HALO tile are staged by LDS-DMA as
// [32-channel block][pixel][plane][32 channels] images (rows of P*64 bytes) and the fragments are read with
// ds_read_b64_tr_b16, the gfx950 transposing LDS read (4 pixel rows x 16 channels per 16-lane group, delivered
// channel-per-lane): conflict free for these row lengths (q*192 mod 256 and q*64 mod 256 hit four distinct
// 64-byte quarters).  No operand is split, rounded or shuffled in registers: P = 3 reads the three bf16 pieces of
// the float32 values straight from the plane tensors (six MFMAs per 32x32x16 block product), P = 1 reads bf16.
// One Yop fragment serves all taps; an Xop fragment is re-read per tap at a shifted pixel offset.
#include <algorithm>

#include "planes.hpp"

namespace rfi {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef const __attribute__((address_space(1))) void gbl_void;

struct PWgradDev {
    PWgradArgs a;
    int nsplit;                  // workgroups along the pixel split
    int64_t slab_stride;         // floats per slab
    unsigned x_zero[2], y_zero;  // byte offsets of zero areas at the end of the operand tensors
    int nkx;                     // Xop chunks over both segments
};

template <int R, int S, int BYB, int BXB, int TH, int TW, int P>
struct PWCfg {
    static constexpr int NT = 256;
    static constexpr int NTAP = R * R;
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH * S + R - S, HW = TW * S + R - S, HP = HH * HW;
    static constexpr int BLOCKS = BYB * BXB;
    static constexpr int WP = 4 / BLOCKS;                    // waves splitting the tile's k-steps
    static constexpr int KS = BM / 16;                       // k-steps (16 pixels) per tile
    static constexpr int KS_W = KS / WP;
    static constexpr int ROW = P * 64;                       // bytes per pixel of a 32-channel block image
    static constexpr int Y_SLOTS = BYB * BM * P * 4, X_SLOTS = BXB * HP * P * 4;
    static constexpr int Y_ITEMS = (Y_SLOTS + NT - 1) / NT, X_ITEMS = (X_SLOTS + NT - 1) / NT;
    static constexpr int Y_BYTES = Y_ITEMS * NT * 16, X_BYTES = X_ITEMS * NT * 16;
    static constexpr int TC = (NTAP % 3 == 0) ? 3 : NTAP;    // taps per cross-wave reduction round
    static constexpr int RED_BYTES = (WP > 1) ? BLOCKS * TC * 4096 : 0;
    static constexpr int LDS_BYTES = (Y_BYTES + X_BYTES) > RED_BYTES ? (Y_BYTES + X_BYTES) : RED_BYTES;
    static_assert(BLOCKS == 1 || BLOCKS == 2 || BLOCKS == 4, "1, 2 or 4 channel blocks");
    static_assert(BM % 16 == 0 && KS % WP == 0 && TW % 4 == 0, "tile must split into k-steps of 16 pixels");
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

// TG tap groups: a workgroup accumulates NTAP / TG taps (TG = 3: one filter row).  The pixel range is then split
// TG times less for the same number of workgroups, i.e. TG times fewer partial slabs are written and reduced -- at the
// price of staging each tile TG times, which the deep layers (small maps, 64 x 64-channel blocks) can afford.
template <int R, int S, int BYB, int BXB, int TH, int TW, int P, int TG>
__global__ __launch_bounds__(256, 2) void pwgrad_kernel(PWgradDev d) {
    using C = PWCfg<R, S, BYB, BXB, TH, TW, P>;
    constexpr int NTG = C::NTAP / TG;
    static_assert(C::NTAP % TG == 0, "tap groups must divide the taps");
    const PWgradArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const sY = smem;
    unsigned char* const sX = smem + C::Y_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = wave % C::BLOCKS, ps = wave / C::BLOCKS;
    const int by = blk / BXB, bx = blk % BXB;
    const int yb0 = blockIdx.y * BYB, xb0 = blockIdx.z * BXB;        // first 32-channel block of the workgroup
    const int split = blockIdx.x / TG, tap0 = (blockIdx.x % TG) * NTG;

    // ---- DMA descriptors: slot s = it * 256 + tid of an operand image [block][pixel][plane][4 pieces of 8 ch]
    // y_d / x_d: pixel row << 20 | pixel column << 8 | (chunk-in-operand * P + plane) * 2 + half  (or ~0: zero)
    unsigned y_d[C::Y_ITEMS], x_d[C::X_ITEMS];
#pragma unroll
    for (int it = 0; it < C::Y_ITEMS; ++it) {
        const int s = it * C::NT + tid;
        const int c4 = s & 3, plane = (s >> 2) % P, pix = (s / (4 * P)) % C::BM, b = s / (4 * P * C::BM);
        const int chunk = (yb0 + b) * 2 + (c4 >> 1);
        y_d[it] = (s < C::Y_SLOTS && chunk < a.yop.nchunks)
                      ? ((unsigned)(pix / TW) << 20) | ((unsigned)(pix % TW) << 8) | (unsigned)((chunk * P + plane) * 2 + (c4 & 1))
                      : 0xffffffffu;
    }
    unsigned x_seg = 0;                              // bit it: slot it reads segment 1
#pragma unroll
    for (int it = 0; it < C::X_ITEMS; ++it) {
        const int s = it * C::NT + tid;
        const int c4 = s & 3, plane = (s >> 2) % P, pix = (s / (4 * P)) % C::HP, b = s / (4 * P * C::HP);
        int chunk = (xb0 + b) * 2 + (c4 >> 1);
        const bool ok = s < C::X_SLOTS && chunk < d.nkx;
        if (ok && chunk >= a.xop[0].nchunks) {
            chunk -= a.xop[0].nchunks;
            x_seg |= 1u << it;
        }
        x_d[it] = ok ? ((unsigned)(pix / C::HW) << 20) | ((unsigned)(pix % C::HW) << 8) | (unsigned)((chunk * P + plane) * 2 + (c4 & 1))
                     : 0xffffffffu;
    }

    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    // per-segment operand constants in registers: indexing the kernel argument by the (per-lane) segment number made
    // every Xop DMA wait for a vector load of its own base / stride / zero offset -- and with it for every DMA in flight
    const unsigned char* const xbase0 = reinterpret_cast<const unsigned char*>(a.xop[0].p);
    const unsigned char* const xbase1 = reinterpret_cast<const unsigned char*>(a.xop[1].p);
    const unsigned xps0 = (unsigned)a.xop[0].pstride * 2u, xps1 = (unsigned)a.xop[1].pstride * 2u;
    const unsigned xz0 = d.x_zero[0], xz1 = d.x_zero[1];
    auto issue_tile = [&](int tile) {
        const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = ty_i * TH, ox0 = tx_i * TW;
        const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
        const unsigned char* yb = reinterpret_cast<const unsigned char*>(a.yop.p);
        const unsigned yps = (unsigned)a.yop.pstride * 2u;
#pragma unroll
        for (int it = 0; it < C::Y_ITEMS; ++it) {
            const int y = oy0 + (int)(y_d[it] >> 20), x = ox0 + (int)((y_d[it] >> 8) & 0xfff);
            const bool ok = y_d[it] != 0xffffffffu && y < a.H && x < a.W;
            const unsigned off = ok ? (unsigned)((n * a.H + y) * a.W + x) * yps + (y_d[it] & 0xff) * 16u : d.y_zero;
            __builtin_amdgcn_global_load_lds((gbl_void*)(yb + off), (lds_void*)(sY + (it * C::NT + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const bool seg = (x_seg >> it) & 1;
            const unsigned char* xb = seg ? xbase1 : xbase0;
            const unsigned xps = seg ? xps1 : xps0;
            const int iy = iy0 + (int)(x_d[it] >> 20), ix = ix0 + (int)((x_d[it] >> 8) & 0xfff);
            const bool ok = x_d[it] != 0xffffffffu && (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx;
            const unsigned off = ok ? (unsigned)((n * a.Hx + iy) * a.Wx + ix) * xps + (x_d[it] & 0xff) * 16u : (seg ? xz1 : xz0);
            __builtin_amdgcn_global_load_lds((gbl_void*)(xb + off), (lds_void*)(sX + (it * C::NT + wave * 64) * 16), 16, 0, 0);
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

    f32x16 acc[NTG];
    int toffs[NTG];                                   // LDS byte offset of this workgroup's taps inside a halo image
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
        toffs[t] = (((tap0 + t) / R) * C::HW + ((tap0 + t) % R)) * C::ROW;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    }

    const unsigned char* const yimg = sY + by * (C::BM * C::ROW) + lane_off;
    const unsigned char* const ximg = sX + bx * (C::HP * C::ROW) + lane_off;
    const int my_tiles = split < ntiles ? (ntiles - split + d.nsplit - 1) / d.nsplit : 0;
    for (int k = 0; k < my_tiles; ++k) {
        issue_tile(split + k * d.nsplit);
        __syncthreads();                             // s_waitcnt vmcnt(0) + barrier: the images have landed
        if constexpr (P == 3) {       // (the pipelined form below spills here: 144 accumulators + two fragment sets of 3 planes)
#pragma unroll
            for (int kk = 0; kk < C::KS_W; ++kk) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
                bf16x8 af[P];
    #pragma unroll
                for (int p = 0; p < P; ++p)
                    af[p] = tr_frag(yimg + ypix(t0) * C::ROW + p * 64, yimg + ypix(t1) * C::ROW + p * 64);
                const int x0 = xpix(t0), x1 = xpix(t1);
    #pragma unroll
                for (int tap = 0; tap < NTG; ++tap) {
                    const int toff = toffs[tap];
                    bf16x8 bf[P];
    #pragma unroll
                    for (int p = 0; p < P; ++p)
                        bf[p] = tr_frag(ximg + x0 * C::ROW + toff + p * 64, ximg + x1 * C::ROW + toff + p * 64);
                    acc[tap] = mma<P>(af, bf, acc[tap]);
                }
            }
        } else {
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
                    const int toff = toffs[tap];
    #pragma unroll
                    for (int p = 0; p < P; ++p)
                        bf[p] = tr_frag(ximg + x0 * C::ROW + toff + p * 64, ximg + x1 * C::ROW + toff + p * 64);
                };
                load_b(0, 0, bfr[0]);
    #pragma unroll
                for (int kk = 0; kk < C::KS_W; ++kk) {
                    load_a(kk);
    #pragma unroll
                    for (int tap = 0; tap < NTG; ++tap) {
                        constexpr int NT_ = NTG;
                        const int cur = (kk * NT_ + tap) & 1;
                        if (tap + 1 < NT_) load_b(kk, tap + 1, bfr[cur ^ 1]);
                        else if (kk + 1 < C::KS_W) load_b(kk + 1, 0, bfr[cur ^ 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[tap] = mma<P>(af, bfr[cur], acc[tap]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        __syncthreads();                             // everybody is done reading before the next tile is staged
    }

    // ---- waves that split the tile's k-steps (WP > 1) add their accumulators through LDS, TC taps at a time
    if constexpr (C::WP > 1) {
        constexpr int TC = (NTG % 3 == 0) ? 3 : NTG;          // (<= C::TC: the scratch is sized for that)
        float* s_red = reinterpret_cast<float*>(smem) + blk * TC * 1024;
#pragma unroll
        for (int t0 = 0; t0 < NTG; t0 += TC) {
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
    // ---- the workgroup's partial slab: rows (reg) = cy, cols (lane & 31) = cx; padding channels are skipped
    if (ps == 0) {
        float* slab = a.slab + (size_t)split * d.slab_stride;
        const int li = lane & 31, lh = lane >> 5;
        const int cxp = (xb0 + bx) * 32 + li;                    // channel index in the padded chunk list
        const int pad0 = a.xop[0].nchunks * 16;
        const int seg = cxp >= pad0 ? 1 : 0;
        const int cl = seg ? cxp - pad0 : cxp;
        const bool xok = cl < a.seg_c[seg];
        const int cx = (seg ? a.seg_c[0] : 0) + cl;
        // layout entries beyond the true channels (behind the LAST segment) are written as zeros
        const bool xpad = !xok && seg == a.nseg - 1 && cx < a.cx_layout;
#pragma unroll
        for (int tap = 0; tap < NTG; ++tap) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cy = (yb0 + by) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cy < a.Cy && (xok || xpad))
                    slab[(int64_t)(tap0 + tap) * a.tap_stride + (int64_t)cy * a.sy + (int64_t)cx * a.sx] = xok ? acc[tap][r] : 0.0f;
            }
        }
    }
}

struct Plan { int nsplit; int64_t slab_stride; };

// tap groups of a launch: one filter row per workgroup where the output has >= 16 channel blocks of 64 x 64 (the deep
// layers: their maps are small, so staging them three times costs less than two thirds of the slab traffic saves)
template <int BYB, int BXB>
int tap_groups(const PWgradArgs& a, int nkx) {
    if (a.R != 3) return 1;
    static const int force = getenv("RFI_PWGRAD_TG") ? atoi(getenv("RFI_PWGRAD_TG")) : 0;     // A/B runs: 1 or 3
    if (force == 1 || force == 3) return force;
    const int chunks = (int)cdiv(plane_chunks(a.Cy), 2 * BYB) * (int)cdiv(nkx, 2 * BXB);
    // ... and few pixels: with many (the 256 x 256 maps of a 1024 x 1024 sample) the slabs are a small share and staging every tile
    // three times makes the kernel LDS-DMA bound (resnet1024: 560 TFLOP/s with tap groups, -2.5 % step time without them there; measured
    // thresholds 8 k / 32 k / none: 12.72 / 12.83 / 13.04 ms per step there, 3.561 / 3.573 / 3.583 ms on the U-Net at 64 x 128 x 128)
    static const int64_t max_px = getenv("RFI_PWGRAD_TG_PIXELS") ? atoll(getenv("RFI_PWGRAD_TG_PIXELS")) : 8192;
    return chunks >= 16 && (int64_t)a.N * a.H * a.W <= max_px ? 3 : 1;
}
template <int R, int S, int BYB, int BXB, int TH, int TW>
Plan plan_cfg(const PWgradArgs& a, int nkx) {
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int chunks = (int)cdiv(plane_chunks(a.Cy), 2 * BYB) * (int)cdiv(nkx, 2 * BXB) * tap_groups<BYB, BXB>(a, nkx);
    int nsplit = (int)cdiv(512, chunks);             // two workgroups per CU in total
    if (nsplit > ntiles) nsplit = ntiles;
    if (nsplit < 1) nsplit = 1;
    return Plan{nsplit, (int64_t)R * R * a.tap_stride};
}

template <int R, int S, int BYB, int BXB, int TH, int TW, int P, int TG = 0>
void launch_cfg(rfi_ctx* ctx, PWgradDev& d) {
    if constexpr (TG == 0) {                          // pick the tap grouping of this launch
        if constexpr (R == 3) {
            if (tap_groups<BYB, BXB>(d.a, d.nkx) == 3) return launch_cfg<R, S, BYB, BXB, TH, TW, P, 3>(ctx, d);
        }
        return launch_cfg<R, S, BYB, BXB, TH, TW, P, 1>(ctx, d);
    }
    constexpr int TGK = TG == 0 ? 1 : TG;
    using C = PWCfg<R, S, BYB, BXB, TH, TW, P>;
    const PWgradArgs& a = d.a;
    const Plan p = plan_cfg<R, S, BYB, BXB, TH, TW>(a, d.nkx);
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)p.nsplit * p.slab_stride, "pwgrad: slab workspace too small");
    d.nsplit = p.nsplit;
    d.slab_stride = p.slab_stride;
    dim3 grid(p.nsplit * TGK, (unsigned)cdiv(plane_chunks(a.Cy), 2 * BYB), (unsigned)cdiv(d.nkx, 2 * BXB));
    const size_t lds = C::LDS_BYTES;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pwgrad_kernel<R, S, BYB, BXB, TH, TW, P, TGK>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops
                                               : 2.0 * a.N * a.H * a.W * (double)a.Cy * (a.seg_c[0] + a.seg_c[1]) * R * R;
        std::string label;
        if (ctx->profiling)
            label = "pwgrad R" + std::to_string(R) + " N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" +
                    std::to_string(a.W) + " cx" + std::to_string(a.seg_c[0] + a.seg_c[1]) + " cy" + std::to_string(a.Cy) +
                    " split" + std::to_string(p.nsplit) + (TGK > 1 ? "x3taps" : "") + (P == 3 ? " 3xbf16" : " bf16");
        const double cx = a.seg_c[0] + a.seg_c[1];
        const double bytes = 2.0 * P * ((double)a.N * a.Hx * a.Wx * cx + (double)a.N * a.H * a.W * a.Cy) + 4.0 * R * R * cx * a.Cy;
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
        hipLaunchKernelGGL((pwgrad_kernel<R, S, BYB, BXB, TH, TW, P, TGK>), grid, dim3(256), lds, ctx->stream, d);
        check_launch("pwgrad");
    }
    launch_reduce_slabs(ctx, a.slab, p.nsplit, p.slab_stride, a.dw);
}

enum { SEL_PLAN = 0, SEL_LAUNCH = 1 };

template <int P, int R = 3, int S = 1>
Plan select(rfi_ctx* ctx, PWgradDev& d, int what) {
    const PWgradArgs& a = d.a;
    const bool y2 = plane_chunks(a.Cy) > 2, x2 = d.nkx > 2;       // more than one 32-channel block
#define RFI_PW(BYB_, BXB_, TH_, TW_)                                                  \
    do {                                                                              \
        if (what == SEL_LAUNCH) launch_cfg<R, S, BYB_, BXB_, TH_, TW_, P>(ctx, d);    \
        return plan_cfg<R, S, BYB_, BXB_, TH_, TW_>(a, d.nkx);                        \
    } while (0)
    if (y2 && x2) RFI_PW(2, 2, 8, 8);
    if (y2) RFI_PW(2, 1, 8, 8);
    if (x2) RFI_PW(1, 2, 8, 8);
    if (a.W >= 16) RFI_PW(1, 1, 8, 16);
    RFI_PW(1, 1, 16, 8);
#undef RFI_PW
}

// the shape classes of a launch: 3x3 stride 1 pad 1 (both arithmetics); bfloat16 flow also 3x3 stride 2 pad 1 and 1x1 stride 2
// (the ResNet-style encoder's stage transitions, on the full-resolution Xop: the halo tile has the stride)
int shape_class(const PWgradArgs& a) {
    if (a.R == 3 && a.S == 1 && a.pad == 1) return 0;
    if (a.P == 1 && a.R == 3 && a.S == 2 && a.pad == 1) return 1;
    if (a.P == 1 && a.R == 1 && a.S == 2 && a.pad == 0) return 2;
    if (a.P == 1 && a.R == 2 && a.S == 2 && a.pad == 0) return 3;      // ConvTranspose2d(k2, s2): Xop = the output gradient
    return -1;
}
Plan select_any(rfi_ctx* ctx, PWgradDev& d, int what) {
    switch (shape_class(d.a)) {
        case 1: return select<1, 3, 2>(ctx, d, what);
        case 2: return select<1, 1, 2>(ctx, d, what);
        case 3: return select<1, 2, 2>(ctx, d, what);
        default: return d.a.P == 3 ? select<3>(ctx, d, what) : select<1>(ctx, d, what);
    }
}

void fill_dev(const PWgradArgs& a, PWgradDev& d) {
    d.a = a;
    if (a.nseg == 1) { d.a.xop[1] = PlaneSeg{a.xop[0].p, a.xop[0].pstride, 0}; d.a.seg_c[1] = 0; }
    d.nkx = d.a.xop[0].nchunks + d.a.xop[1].nchunks;
    const int64_t xpix = (int64_t)a.N * a.Hx * a.Wx, ypix = (int64_t)a.N * a.H * a.W;
    for (int s = 0; s < 2; ++s) {
        const int64_t bytes = xpix * d.a.xop[s].pstride * 2;
        RFI_REQUIRE(bytes + 64 < ((int64_t)1 << 32), "pwgrad: Xop too large for 32-bit byte offsets");
        d.x_zero[s] = (unsigned)bytes;
    }
    const int64_t ybytes = ypix * a.yop.pstride * 2;
    RFI_REQUIRE(ybytes + 64 < ((int64_t)1 << 32), "pwgrad: Yop too large for 32-bit byte offsets");
    d.y_zero = (unsigned)ybytes;
    d.nsplit = 1;
    d.slab_stride = 0;
}

}  // namespace

size_t pwgrad_slab_floats(const PWgradArgs& a) {
    PWgradDev d;
    fill_dev(a, d);
    const Plan p = select_any(nullptr, d, SEL_PLAN);
    return (size_t)p.nsplit * p.slab_stride;
}

void launch_pwgrad(rfi_ctx* ctx, const PWgradArgs& a) {
    RFI_REQUIRE(a.P == 1 || a.P == 3, "pwgrad: planes must be 1 or 3");
    RFI_REQUIRE(shape_class(a) >= 0, "pwgrad: 3x3 stride 1 pad 1; bfloat16 flow also 3x3 stride 2 pad 1, 1x1 stride 2, 2x2 stride 2 pad 0");
    RFI_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cy > 0 && a.seg_c[0] > 0, "pwgrad: empty shape");
    RFI_REQUIRE(std::max(a.yop.nchunks, std::max(a.xop[0].nchunks, a.nseg > 1 ? a.xop[1].nchunks : 0)) * a.P * 2 <= 256,
                "pwgrad: more than 42 (P = 3) / 128 (P = 1) chunks per operand segment");
    PWgradDev d;
    fill_dev(a, d);
    select_any(ctx, d, SEL_LAUNCH);
}

// Bridge for callers that hold float32 NHWC tensors (the kernel-level C ABI)
void launch_pwgrad_from_f32(rfi_ctx* ctx, const WgradArgs& w, int P) {
    RFI_REQUIRE(w.R == 3 && w.S == 1 && w.pad == 1, "pwgrad bridge: 3x3 stride-1 only");
    const int64_t xpix = (int64_t)w.N * w.Hx * w.Wx, ypix = (int64_t)w.N * w.H * w.W;
    const size_t xe = plane_elems(xpix, w.Cx, P), ye = plane_elems(ypix, w.Cy, P);
    bf16_t* xp = static_cast<bf16_t*>(ctx->alloc(xe * 2 + 64));
    bf16_t* yp = static_cast<bf16_t*>(ctx->alloc(ye * 2 + 64));
    struct Free {
        rfi_ctx* c; void* a; void* b;
        ~Free() { (void)hipStreamSynchronize(c->stream); try { c->release(a); c->release(b); } catch (...) {} }
    } fr{ctx, xp, yp};
    RFI_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(xp) + xe * 2, 0, 64, ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(yp) + ye * 2, 0, 64, ctx->stream));
    const int64_t xs = (int64_t)plane_chunks(w.Cx) * P * 16, ys = (int64_t)plane_chunks(w.Cy) * P * 16;
    launch_act_split(ctx, w.xop, xpix, w.Cx, w.xf_x, P, xp, xs);
    launch_act_split(ctx, w.yop, ypix, w.Cy, w.xf_y, P, yp, ys);
    PWgradArgs a;
    a.xop[0] = PlaneSeg{xp, xs, plane_chunks(w.Cx)};
    a.nseg = 1; a.seg_c[0] = w.Cx;
    a.yop = PlaneSeg{yp, ys, plane_chunks(w.Cy)};
    a.Cy = w.Cy; a.P = P;
    a.N = w.N; a.H = w.H; a.W = w.W; a.Hx = w.Hx; a.Wx = w.Wx;
    a.dw = w.dw; a.tap_stride = w.tap_stride; a.sy = w.sy; a.sx = w.sx;
    a.slab = w.slab; a.slab_floats = w.slab_floats;
    a.algo_flops = w.algo_flops;
    launch_pwgrad(ctx, a);
}
size_t pwgrad_slab_floats_f32(const WgradArgs& w) {
    PWgradArgs a;
    a.xop[0].nchunks = plane_chunks(w.Cx);
    a.xop[0].pstride = (int64_t)a.xop[0].nchunks * 48;
    a.nseg = 1; a.seg_c[0] = w.Cx;
    a.yop.nchunks = plane_chunks(w.Cy);
    a.yop.pstride = (int64_t)a.yop.nchunks * 48;
    a.Cy = w.Cy;
    a.N = w.N; a.H = w.H; a.W = w.W; a.Hx = w.Hx; a.Wx = w.Wx;
    a.tap_stride = w.tap_stride;
    return pwgrad_slab_floats(a);
}

}  // namespace rfi
