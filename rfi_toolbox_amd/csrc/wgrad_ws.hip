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
//
// [r4] Two more template parameters.  ST: the stride of the convolution -- ST = 2 with R = 2 / pad 0 is the weight gradient
// of the k2 / s2 TRANSPOSED conv (Xop = the gradient on the 2H x 2W grid, Yop = the layer's input on the H x W grid: tap
// (r, s) pairs Yop pixel (y, x) with Xop pixel (2y + r, 2x + s); the staged Xop tile is the 2TH x 2TW block under the Yop
// tile, no halo).  PWV: the number of producer waves.  A staged element of a ONE-tap contraction (R = 1, or R = 2 / ST = 2:
// nothing is shared between taps) feeds few MFMAs, so those shapes are bound by the producers' split, and a producer wave
// issues one vector instruction per ~14 cycles beside an MFMA wave -- latency, not issue bandwidth: with PWV = 8 every SIMD
// hosts two producer waves (768 threads, 168 registers) and the split of a tile takes about two thirds as long.
#include <algorithm>
#include <cstdio>
#include <vector>

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
    unsigned long long* stamps;       // RFI_DIAG_STAMPS build: per-wave cycle sums
};

template <int R, int BYB, int BXB, int TH, int TW, int P, int ST = 1, int PWV = 4>
struct WWCfg {
    static constexpr int NTAP = R * R;
    static constexpr int BM = TH * TW;
    static constexpr int NPT = 64 * PWV;                       // producer threads
    static constexpr int HH = ST * (TH - 1) + R, HW = ST * (TW - 1) + R, HP = HH * HW;
    static constexpr int BLOCKS = BYB * BXB;
    static constexpr int WP = 4 / BLOCKS;                       // consumer waves that share one channel block (split the k-steps)
    static constexpr int KS = BM / 16, KS_W = KS / WP;
    static constexpr int ROW = P * 64;                          // bytes per pixel of a 32-channel block image: P planes x 32 channels
    static constexpr int YQ = BYB * 8, XQ = BXB * 8;            // float4 groups per pixel
    static constexpr int Y_ITEMS = (BM * YQ + NPT - 1) / NPT, X_ITEMS = (HP * XQ + NPT - 1) / NPT;
    static constexpr int Y_BYTES = BYB * BM * ROW, X_BYTES = BXB * HP * ROW;
    static constexpr int STAGE = Y_BYTES + X_BYTES;
    static constexpr int TC = (NTAP % 3 == 0) ? 3 : NTAP;
    static constexpr int RED_BYTES = (WP > 1) ? BLOCKS * TC * 4096 : 0;
    static constexpr int TR_FLOATS = 32 * 33;                   // per consumer wave: the transposing slab store's scratch
    static constexpr int EPI_BYTES = RED_BYTES + 4 * TR_FLOATS * 4;
    static constexpr int LDS_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
    static_assert(BLOCKS == 1 || BLOCKS == 2 || BLOCKS == 4, "1, 2 or 4 channel blocks");
    static_assert(BM % 16 == 0 && KS % WP == 0 && TW % 4 == 0, "tile must split into k-steps of 16 pixels");
    static_assert(NPT % YQ == 0 && NPT % XQ == 0, "a producer thread keeps one channel group for all its items");
    static_assert(PWV == 4 || PWV == 8 || PWV == 12, "4 consumer waves + 4, 8 or 12 producer waves");
    static_assert(ST == 1 || (ST == 2 && R == 2), "stride 2: the transposed conv's 2x2 taps");
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

// XM: the load transforms -- 1 Xop = relu(x * scale + shift), Yop plain (the 3x3 layers of the U-Net: compile-time, no
// tests); 2 whatever the arguments say (either operand, any activation, or none)
template <int R, int BYB, int BXB, int TH, int TW, int P, int XM, int ST = 1, int PWV = 4>
__global__ __launch_bounds__(256 + 64 * PWV) void wgrad_ws_kernel(WWsDev d) {
    using C = WWCfg<R, BYB, BXB, TH, TW, P, ST, PWV>;
    const WgradArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RFI_DIAG_STAMPS
    unsigned long long st_[4] = {0, 0, 0, 0};
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cy0 = blockIdx.y * BYB * 32, cx0 = blockIdx.z * BXB * 32;
    const int split = blockIdx.x;
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    const int my_tiles = split < ntiles ? (ntiles - split + d.nsplit - 1) / d.nsplit : 0;   // (>= 1: nsplit <= ntiles)

    if (wave >= 4) {
        // =============================================================== producers
        // The producers are the bottleneck of this kernel (cycle stamps, round 3: 9.3 k cycles per 8 x 8 tile against
        // 6.9 k of MFMA issue) and they issue one vector instruction per ~14 cycles beside the MFMA waves, so this code
        // counts instructions: everything that does not depend on the tile (pixel coordinates inside the tile, byte offsets
        // relative to its first pixel, LDS addresses) is computed once; a tile costs one scalar base and one add per item
        // (interior tiles) or the bounds tests (edge tiles); the transforms are written per element, unfused and unpacked.
        const int ptid = tid - 256;
        constexpr int YSTEP = C::NPT / C::YQ, XSTEP = C::NPT / C::XQ;      // pixels between a thread's consecutive items
        const int yq = ptid % C::YQ, xq = ptid % C::XQ;
        const int ypix0 = ptid / C::YQ, xpix0 = ptid / C::XQ;
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
        // an item that does not exist (beyond the tile's pixels or the tensor's channels) gets NOWHERE as its relative
        // offset: base + NOWHERE lies beyond the buffer for every tile base (bases are > -2^24 and < 0x7f000000: the launch
        // checks both), so the load returns zero
        constexpr unsigned NOWHERE = 0x81000000u, OUTSIDE = 0x80000000u;
        unsigned y_rel[C::Y_ITEMS], x_rel[C::X_ITEMS];
        int y_dyx[C::Y_ITEMS], x_dyx[C::X_ITEMS];        // row << 16 | column of the item's pixel inside the tile / halo tile
#pragma unroll
        for (int it = 0; it < C::Y_ITEMS; ++it) {
            const int pix = ypix0 + it * YSTEP, dy = pix / TW, dx = pix % TW;
            y_rel[it] = (pix < C::BM && y_cok) ? (unsigned)((dy * a.W + dx) * a.yop.pstride + cyq) * 4u : NOWHERE;
            y_dyx[it] = pix < C::BM ? (dy << 16) | dx : 0x7fff7fff;
        }
#pragma unroll
        for (int it = 0; it < C::X_ITEMS; ++it) {
            const int pix = xpix0 + it * XSTEP, dy = pix / C::HW, dx = pix % C::HW;
            x_rel[it] = (pix < C::HP && x_cok) ? (unsigned)((dy * a.Wx + dx) * a.xop.pstride + cxq) * 4u : NOWHERE;
            x_dyx[it] = pix < C::HP ? (dy << 16) | dx : 0x7fff7fff;
        }
        // LDS byte address of this thread's 4 channels of item 0 (item it: + it * STEP * ROW, an immediate)
        const int y_lds = (yq >> 3) * (C::BM * C::ROW) + (yq & 7) * 8 + ypix0 * C::ROW;
        const int x_lds = (xq >> 3) * (C::HP * C::ROW) + (xq & 7) * 8 + xpix0 * C::ROW;
        struct TileRegs {
            u32x4 yreg[C::Y_ITEMS], xreg[C::X_ITEMS];
            unsigned yvalid = ~0u, xvalid = ~0u;         // per-item "inside the image" bits of the tile in the registers
            bool yfull = true, xin = true;               // ... all set (wave-uniform)
        };
        // (a second tile of loads in flight, in a second register set, was measured for the one-tap shapes: +-0 -- the producers
        // do not wait for their loads -- and taken out again)
        TileRegs tr0;
        auto load_tile = [&](int tile, TileRegs& tr) {
            u32x4 (&yreg)[C::Y_ITEMS] = tr.yreg;
            u32x4 (&xreg)[C::X_ITEMS] = tr.xreg;
            unsigned &yvalid = tr.yvalid, &xvalid = tr.xvalid;
            bool &yfull = tr.yfull, &xin = tr.xin;
            const int tx_i = tile % tiles_x, ty_i = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
            const int oy0 = ty_i * TH, ox0 = tx_i * TW;
            const int iy0 = oy0 * ST - a.pad, ix0 = ox0 * ST - a.pad;
            const unsigned ybase = (unsigned)(((n * a.H + oy0) * a.W + ox0) * a.yop.pstride) * 4u;
            const unsigned xbase = (unsigned)(((n * a.Hx + iy0) * a.Wx + ix0) * a.xop.pstride) * 4u;     // (may be "negative")
            yfull = oy0 + TH <= a.H && ox0 + TW <= a.W;
            xin = iy0 >= 0 && ix0 >= 0 && iy0 + C::HH <= a.Hx && ix0 + C::HW <= a.Wx;
            if (yfull) {
                yvalid = ~0u;
#pragma unroll
                for (int it = 0; it < C::Y_ITEMS; ++it) yreg[it] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ybase + y_rel[it], 0, 0);
            } else {
                yvalid = 0;
#pragma unroll
                for (int it = 0; it < C::Y_ITEMS; ++it) {
                    const bool ok = oy0 + (y_dyx[it] >> 16) < a.H && ox0 + (y_dyx[it] & 0xffff) < a.W;
                    yreg[it] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? ybase + y_rel[it] : OUTSIDE, 0, 0);
                    yvalid |= (ok ? 1u : 0u) << it;
                }
            }
            if (xin) {
                xvalid = ~0u;
#pragma unroll
                for (int it = 0; it < C::X_ITEMS; ++it) xreg[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xbase + x_rel[it], 0, 0);
            } else {
                xvalid = 0;
#pragma unroll
                for (int it = 0; it < C::X_ITEMS; ++it) {
                    const bool ok = (unsigned)(iy0 + (x_dyx[it] >> 16)) < (unsigned)a.Hx && (unsigned)(ix0 + (x_dyx[it] & 0xffff)) < (unsigned)a.Wx;
                    xreg[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xbase + x_rel[it] : OUTSIDE, 0, 0);
                    xvalid |= (ok ? 1u : 0u) << it;
                }
            }
        };
        // split ONCE into (h, m, l) and write the planes of the thread's 4 channels (8 bytes each)
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
        // the load transform of one item.  MASK: the tile touches the image border -- zero padding AFTER the transform
        // (`valid`: the item's pixel is inside).  Per element, unfused (the values every other consumer of the tensor
        // computes), no packed fp32 forms (slower beside the MFMA waves)
        auto xform = [&](f32x4 v, const f32x4& sc, const f32x4& sh, const InXform& xf, auto mask_c, unsigned valid) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)valid, 0, 1);          // 0 or ~0
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e] * sc[e] + sh[e];
                if constexpr (XM == 1) {
                    asm("v_max_f32 %0, 0, %1" : "=v"(t) : "v"(t));                        // ReLU
                } else {
                    if (xf.relu) {
                        const float u = t * xf.slope;                                       // slope 0 = ReLU
                        asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(u));
                    }
                }
                if constexpr (decltype(mask_c)::value) t = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & m);
                v[e] = t;
            }
            return v;
        };
        auto store_tile = [&](int buf, const TileRegs& tr) {
            const u32x4 (&yreg)[C::Y_ITEMS] = tr.yreg;
            const u32x4 (&xreg)[C::X_ITEMS] = tr.xreg;
            const unsigned yvalid = tr.yvalid, xvalid = tr.xvalid;
            const bool xin = tr.xin;
            unsigned char* const sY = smem + buf * C::STAGE + y_lds;
            unsigned char* const sX = smem + buf * C::STAGE + C::Y_BYTES + x_lds;
            const bool do_y = XM == 2 && a.xf_y.scale != nullptr, do_x = XM == 1 || (XM == 2 && a.xf_x.scale != nullptr);
#pragma unroll
            for (int it = 0; it < C::Y_ITEMS; ++it) {
                f32x4 v = __builtin_bit_cast(f32x4, yreg[it]);
                if (do_y) v = xform(v, ysc, ysh, a.xf_y, std::true_type{}, yvalid >> it);
                if (it * YSTEP + YSTEP <= C::BM || ypix0 + it * YSTEP < C::BM) put(v, sY + it * YSTEP * C::ROW);
            }
            if (do_x && !xin) {
#pragma unroll
                for (int it = 0; it < C::X_ITEMS; ++it) {
                    const f32x4 v = xform(__builtin_bit_cast(f32x4, xreg[it]), xsc, xsh, a.xf_x, std::true_type{}, xvalid >> it);
                    if (it * XSTEP + XSTEP <= C::HP || xpix0 + it * XSTEP < C::HP) put(v, sX + it * XSTEP * C::ROW);
                }
            } else {
#pragma unroll
                for (int it = 0; it < C::X_ITEMS; ++it) {
                    f32x4 v = __builtin_bit_cast(f32x4, xreg[it]);
                    if (do_x) v = xform(v, xsc, xsh, a.xf_x, std::false_type{}, 0u);
                    if (it * XSTEP + XSTEP <= C::HP || xpix0 + it * XSTEP < C::HP) put(v, sX + it * XSTEP * C::ROW);
                }
            }
        };
        // phase k = -1 is the prologue (tile 0); every later phase stages tile k + 1 while the consumers multiply tile k
        auto phase = [&](int k, TileRegs& tr) {
            WS_T(t0);
            if (k + 1 < my_tiles) {
                store_tile((k + 1) & 1, tr);             // (waits for the loads of tile k + 1, issued a phase ago)
                if (k + 2 < my_tiles) load_tile(split + (k + 2) * d.nsplit, tr);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            WS_T(t1);
            wg_barrier();
            WS_T(t2);
            WS_ACC(0, t0, t1);
            WS_ACC(1, t1, t2);
        };
        load_tile(split, tr0);
        for (int k = -1; k < my_tiles; ++k) phase(k, tr0);     // (my_tiles + 1 phases, as many barriers as the consumers pass)
#ifdef RFI_DIAG_STAMPS
        if (d.stamps && lane == 0 && (PWV == 4 || (wave >= 8 && wave < 12))) {       // (eight records per workgroup: with more producer waves, waves 8-11)
            unsigned long long* o = d.stamps + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (PWV == 4 ? wave : wave - 4)) * 8;
            o[0] = st_[0]; o[1] = st_[1]; o[2] = (unsigned long long)my_tiles;
        }
#endif
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
        auto xpix = [&](int t) { return ST * ((t / TW) * C::HW + (t % TW)); };   // halo pixel of tile pixel t (tap 0)
        f32x16 acc[C::NTAP];
#pragma unroll
        for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        WS_T(tp0);
        wg_barrier();                                    // tile 0 is staged
        WS_T(tp1);
        for (int k = 0; k < my_tiles; ++k) {
            WS_T(t0);
            const unsigned char* const sY = smem + (k & 1) * C::STAGE;
            const unsigned char* const yimg = sY + by * (C::BM * C::ROW) + lane_off;
            const unsigned char* const ximg = sY + C::Y_BYTES + bx * (C::HP * C::ROW) + lane_off;
            // software pipeline over the steps (k-step, tap) of the tile: the Xop fragment of step s + 1 -- and, on the last
            // but one tap of a k-step, the Yop fragment of the next k-step -- is read from LDS WHILE the MFMAs of step s
            // issue: one or two transposing reads behind every MFMA (issued as one burst in front of them, the reads leave
            // the matrix pipe idle for most of the burst; a Yop fragment read at the top of its k-step stalls the first MFMA
            // for the full LDS latency: 7.7 k cycles per tile against 6.9 k of MFMA issue, cycle stamps of round 3)
            bf16x8 af[2][P], bfr[2][P];
            auto load_a = [&](int kk, bf16x8 (&dst)[P]) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
#pragma unroll
                for (int p = 0; p < P; ++p) dst[p] = tr_frag(yimg + t0 * C::ROW + p * 64, yimg + t1 * C::ROW + p * 64);
            };
            auto load_b = [&](int kk, int tap, bf16x8 (&bf)[P]) {
                const int t0 = (ps * C::KS_W + kk) * 16 + 8 * kh + q, t1 = t0 + 4;
                const int x0 = xpix(t0), x1 = xpix(t1);
                const int toff = ((tap / R) * C::HW + (tap % R)) * C::ROW;
#pragma unroll
                for (int p = 0; p < P; ++p)
                    bf[p] = tr_frag(ximg + x0 * C::ROW + toff + p * 64, ximg + x1 * C::ROW + toff + p * 64);
            };
            constexpr int NT_ = C::NTAP, NS = C::KS_W * NT_, TA = NT_ > 1 ? NT_ - 2 : 0, NM = P == 3 ? 6 : 1;
            load_b(0, 0, bfr[0]);
            load_a(0, af[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < C::KS_W; ++kk)
#pragma unroll
            for (int tap = 0; tap < NT_; ++tap) {
                const int st = kk * NT_ + tap;
                const bool has_b = st + 1 < NS, has_a = tap == TA && kk + 1 < C::KS_W;
                if (has_b) load_b((st + 1) / NT_, (st + 1) % NT_, bfr[(st + 1) & 1]);
                if (has_a) load_a(kk + 1, af[(kk + 1) & 1]);
                acc[tap] = mma<P>(af[kk & 1], bfr[st & 1], acc[tap]);
                const int NR = 2 * P * ((has_b ? 1 : 0) + (has_a ? 1 : 0));
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
                    const int cnt = (i + 1) * NR / NM - i * NR / NM;                      // the LDS reads that fall to it (literals)
                    if (cnt == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else if (cnt == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else if (cnt == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    else if (cnt == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    else if (cnt > 4) __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WS_T(t1);
            wg_barrier();                                // every consumer is done with this tile's buffers; tile k + 1 is staged
            WS_T(t2);
            WS_ACC(0, t0, t1);
            WS_ACC(1, t1, t2);
        }
        WS_T(tp2);
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
        WS_T(tp3);
        // ---- the workgroup's partial slab: rows (reg) = cy, cols (lane & 31) = cx
        if (ps == 0 && a.sy == 1 && a.sx != 1) {
            // cy is the contiguous index of dW (the transposed conv's [tap][cout][cin]): the accumulators hold cx along the
            // lanes, so a direct store scatters 64 four-byte words over 64 lines (cycle stamps: 20 k cycles, a quarter of
            // the kernel).  Each wave transposes its 32 x 32 block through a scratch of its own BEHIND the reduction area
            // (the staging buffers are free: every wave is past the loop's last barrier) and stores 128-byte rows.
            float* slab = a.slab + (size_t)split * d.slab_stride;
            const int li = lane & 31, lh = lane >> 5;
            float* const scr = reinterpret_cast<float*>(smem + C::RED_BYTES) + wave * C::TR_FLOATS;
            const int cy = cy0 + by * 32 + li;
#pragma unroll
            for (int tap = 0; tap < C::NTAP; ++tap) {
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[li * 33 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[tap][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its LDS accesses complete in order)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int cxl = 2 * j + lh, cx = cx0 + bx * 32 + cxl;
                    const float v = scr[cxl * 33 + li];
                    if (cy < a.Cy && cx < a.Cx) slab[(int64_t)tap * a.tap_stride + (int64_t)cy + (int64_t)cx * a.sx] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (ps == 0) {
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
#ifdef RFI_DIAG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WS_T(tp4);
        if (d.stamps && lane == 0) {
            unsigned long long* o = d.stamps + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wave) * 8;
            o[0] = st_[0]; o[1] = st_[1]; o[2] = (unsigned long long)my_tiles;
            o[3] = tp1 - tp0; o[4] = tp2 - tp1; o[5] = tp3 - tp2; o[6] = tp4 - tp3;
        }
#endif
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

template <int R, int BYB, int BXB, int TH, int TW, int P, int XM, int ST = 1, int PWV = 4>
void launch_cfg(rfi_ctx* ctx, const WgradArgs& a) {
    using C = WWCfg<R, BYB, BXB, TH, TW, P, ST, PWV>;
    const Plan p = plan_cfg<R, BYB, BXB, TH, TW>(a);
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)p.nsplit * p.slab_stride, "wgrad_ws: slab workspace too small");
    WWsDev d{a, p.nsplit, p.slab_stride, (unsigned)((int64_t)a.N * a.Hx * a.Wx * a.xop.pstride * 4),
             (unsigned)((int64_t)a.N * a.H * a.W * a.yop.pstride * 4), nullptr};
    dim3 grid(p.nsplit, (unsigned)cdiv(a.Cy, 32 * BYB), (unsigned)cdiv(a.Cx, 32 * BXB));
    const size_t lds = C::LDS_BYTES;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_kernel<R, BYB, BXB, TH, TW, P, XM, ST, PWV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
    {
        const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cy * a.Cx * R * R;
        std::string label;
        if (ctx->profiling)
            label = "wgrad_ws R" + std::to_string(R) + (ST == 2 ? "s2" : "") + " N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" +
                    std::to_string(a.W) + " cx" + std::to_string(a.Cx) + " cy" + std::to_string(a.Cy) + " split" +
                    std::to_string(p.nsplit) + (P == 3 ? " 3xbf16" : " bf16");
        const double bytes = 4.0 * ((double)a.N * a.Hx * a.Wx * a.Cx + (double)a.N * a.H * a.W * a.Cy + (double)R * R * a.Cx * a.Cy);
        ProfScope ps(ctx, FAM_WGRAD_MFMA, flops, bytes, label);
#ifdef RFI_DIAG_STAMPS
        const size_t nw = (size_t)grid.x * grid.y * grid.z * 8;
        RFI_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d.stamps), nw * 64));
        RFI_CHECK_HIP(hipMemsetAsync(d.stamps, 0, nw * 64, ctx->stream));
#endif
        hipLaunchKernelGGL((wgrad_ws_kernel<R, BYB, BXB, TH, TW, P, XM, ST, PWV>), grid, dim3(256 + C::NPT), lds, ctx->stream, d);
        check_launch("wgrad_ws");
#ifdef RFI_DIAG_STAMPS
        std::vector<unsigned long long> hs(nw * 8);
        RFI_CHECK_HIP(hipMemcpyAsync(hs.data(), d.stamps, nw * 64, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        double c[7] = {0, 0, 0, 0, 0, 0, 0}, pr[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < 7; ++i) ((w & 7) < 4 ? c : pr)[i] += (double)hs[w * 8 + i];
        std::fprintf(stderr, "[stamps] wgrad_ws<%d,%d,%d,%d,%d> N%d %dx%d cx%d cy%d grid %ux%ux%u tiles/wg %.1f | cycles per tile: consumer mfma %.0f "
                     "barrier %.0f | producer work %.0f barrier %.0f | per consumer wave: prologue %.0f loop %.0f reduce %.0f slab store %.0f\n", R, BYB, BXB, TH, TW, a.N, a.H, a.W, a.Cx, a.Cy, grid.x, grid.y, grid.z,
                     c[2] / (nw / 2), c[0] / c[2], c[1] / c[2], pr[0] / pr[2], pr[1] / pr[2], c[3] / (nw / 2), c[4] / (nw / 2), c[5] / (nw / 2), c[6] / (nw / 2));
        RFI_CHECK_HIP(hipFree(d.stamps));
#endif
    }
    launch_reduce_slabs(ctx, a.slab, p.nsplit, p.slab_stride, a.dw);
}

template <int R>
void select(rfi_ctx* ctx, const WgradArgs& a) {
    const bool y2 = a.Cy > 32, x2 = a.Cx > 32;
    const bool p1 = a.bf16 && !a.bf16x3;
    // XM = 1 (compile-time transform) for the case that carries the U-Net: 3x3, Xop behind BatchNorm + ReLU, Yop plain
    const bool xm1 = R == 3 && a.xf_x.scale && a.xf_x.relu == 1 && a.xf_x.slope == 0.0f && !a.xf_y.scale;
#define RFI_WW(BYB_, BXB_, TH_, TW_)                                                \
    do {                                                                            \
        if constexpr (R == 3) {                                                     \
            if (xm1) {                                                              \
                if (p1) launch_cfg<R, BYB_, BXB_, TH_, TW_, 1, 1>(ctx, a);          \
                else launch_cfg<R, BYB_, BXB_, TH_, TW_, 3, 1>(ctx, a);             \
                return;                                                             \
            }                                                                       \
        }                                                                           \
        if (p1) launch_cfg<R, BYB_, BXB_, TH_, TW_, 1, 2>(ctx, a);                  \
        else launch_cfg<R, BYB_, BXB_, TH_, TW_, 3, 2>(ctx, a);                     \
        return;                                                                     \
    } while (0)
    if constexpr (R == 1) {
        // one tap: producer-bound (header) -- eight producer waves where both operands have 64-channel tiles
        static const int pwv = getenv("RFI_WGRAD_PWV") ? atoi(getenv("RFI_WGRAD_PWV")) : 8;
        if (y2 && x2 && pwv == 8) {
            if (p1) launch_cfg<1, 2, 2, 8, 8, 1, 2, 1, 8>(ctx, a);
            else launch_cfg<1, 2, 2, 8, 8, 3, 2, 1, 8>(ctx, a);
            return;
        }
    }
    if (y2 && x2) RFI_WW(2, 2, 8, 8);
    if (y2) RFI_WW(2, 1, 8, 8);
    if (x2) RFI_WW(1, 2, 8, 8);
    if (a.W >= 16) RFI_WW(1, 1, 8, 16);
    RFI_WW(1, 1, 16, 8);
#undef RFI_WW
}

// the transposed conv's weight gradient (R = 2, stride 2, pad 0): Yop tiles of 4 x 8 pixels against the 8 x 16 Xop pixels
// under them, 64 x 64 channels (a 32-channel Xop: 8 x 8 Yop pixels), eight producer waves
void select_t2(rfi_ctx* ctx, const WgradArgs& a, Plan* plan_only) {
    const bool p1 = a.bf16 && !a.bf16x3;
    const bool y2 = a.Cy > 32, x2 = a.Cx > 32;
    static const int pwv = getenv("RFI_WGRAD_T2_PWV") ? atoi(getenv("RFI_WGRAD_T2_PWV")) : 8;     // (A/B: 4, 8 or 12 producer waves)
#define RFI_WT(BYB_, BXB_, TH_, TW_)                                                       \
    do {                                                                                   \
        if (plan_only) { *plan_only = plan_cfg<2, BYB_, BXB_, TH_, TW_>(a); return; }      \
        if (p1) launch_cfg<2, BYB_, BXB_, TH_, TW_, 1, 2, 2, 8>(ctx, a);                   \
        else if (pwv == 12) launch_cfg<2, BYB_, BXB_, TH_, TW_, 3, 2, 2, 12>(ctx, a);      \
        else if (pwv == 4) launch_cfg<2, BYB_, BXB_, TH_, TW_, 3, 2, 2, 4>(ctx, a);        \
        else launch_cfg<2, BYB_, BXB_, TH_, TW_, 3, 2, 2, 8>(ctx, a);                      \
        return;                                                                            \
    } while (0)
    if (y2 && x2) RFI_WT(2, 2, 4, 8);
    if (y2) RFI_WT(2, 1, 8, 8);
    if (x2) RFI_WT(1, 2, 4, 8);
    RFI_WT(1, 1, 8, 8);
#undef RFI_WT
}

}  // namespace

// the shapes of wgrad_split.hip; 32-bit byte offsets with 2^31 as the "reads zero" offset of the buffer loads
bool wgrad_ws_eligible(const WgradArgs& a) {
    if (a.Cx % 4 || a.Cy % 4 || a.xop.pstride % 4 || a.yop.pstride % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.xop.p) & 15) || (reinterpret_cast<uintptr_t>(a.yop.p) & 15)) return false;
    // (tile bases + an item's relative offset are formed in 32 bits; NOWHERE in the kernel needs bases in (-2^24, 0x7f000000))
    if ((int64_t)a.N * a.Hx * a.Wx * a.xop.pstride * 4 >= 0x7f000000ll || (int64_t)a.N * a.H * a.W * a.yop.pstride * 4 >= 0x7f000000ll) return false;
    if (((int64_t)a.Wx + 1) * a.xop.pstride * 4 >= (1 << 24)) return false;
    if (a.R == 2 && a.S == 2 && a.pad == 0) {             // the transposed conv's weight gradient (select_t2)
        static const bool no_t2 = getenv("RFI_NO_WGRAD_T2") != nullptr;
        return !no_t2 && a.Hx == 2 * a.H && a.Wx == 2 * a.W;
    }
    if (a.S != 1) return false;
    return (a.R == 3 && a.pad == 1) || (a.R == 2 && a.pad == 1) || (a.R == 1 && a.pad == 0);
}
// slab workspace of the shapes no other kernel's plan covers (launch_wgrad sizes the others by wgrad_split's plan)
size_t wgrad_ws_slab_floats(const WgradArgs& a) {
    if (!(a.R == 2 && a.S == 2) || !wgrad_ws_eligible(a)) return 0;
    Plan p{0, 0};
    select_t2(nullptr, a, &p);
    return (size_t)p.nsplit * p.slab_stride;
}
void launch_wgrad_ws(rfi_ctx* ctx, const WgradArgs& a) {
    RFI_REQUIRE(wgrad_ws_eligible(a) && (a.bf16 || a.bf16x3), "wgrad_ws: shape or arithmetic not eligible");
    if (a.R == 3) select<3>(ctx, a);
    else if (a.R == 2 && a.S == 2) select_t2(ctx, a, nullptr);
    else if (a.R == 2) select<2>(ctx, a);
    else select<1>(ctx, a);
}

}  // namespace rfi
