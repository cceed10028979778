// Wave-specialised 3x3 / stride-1 / pad-1 convolution on float32 NHWC tensors in the float32-by-3xbf16 arithmetic
// (six v_mfma_f32_32x32x16_bf16 per 32x32x16 block product, small terms first: planes.hpp, conv_mfma.hip).
//
//   y[n,oy,ox,co] = bias[co] + sum_{tap=(r,s)} sum_ci T(X)[n, oy+r-1, ox+s-1, ci] * W[tap][co][ci]
//
// One 512-thread workgroup per CU, persistent over an XCD-contiguous range of 256-pixel tiles and one block of
// 32 NTL output channels.  Its eight waves have two ROLES (waves w and w + 4 share a SIMD: every SIMD hosts one of each):
//   * waves 4-7, PRODUCERS: per (tile, 16-channel chunk) item they load the float32 halo tile to registers (buffer
//     loads: pixels outside the image read as zero), apply the producing layer's BatchNorm + (Leaky)ReLU (InXform), split
//     every value ONCE into its three bf16 pieces -- all in registers, while the consumers multiply the previous item --
//     and write the pieces to LDS ([pixel][h 32 B | m 32 B | l 32 B | pad 16 B]: conflict-free ds_read_b128 of 32
//     consecutive pixels) in the short window between the two barriers of an item.  The filter chunk (already in MFMA
//     B-operand order in HBM: planes.hpp, wB with P = 3) goes HBM/L2 -> LDS by LDS-DMA, double-buffered;
//   * waves 0-3, CONSUMERS: nothing but ds_read_b128 + MFMA (a wave owns 64 pixels x 32 NTL channels: 108 NTL MFMAs per
//     item, fragments read one tap ahead), then the epilogue of a finished tile straight from the accumulators (a wave
//     store writes 128 contiguous bytes of two pixels).
// Why the roles: in conv_igemm_kernel<...,2> a wave spends 3.5 k cycles on transform + split + LDS writes and 2.5 k on
// load issue per 4.5 k of MFMA (cycle stamps, round 3); the matrix pipe and the vector ALU of a SIMD issue from
// different waves concurrently, so here the split of item q + 1 runs UNDER the MFMAs of item q.  Vector instructions
// issue about half as fast beside a wave that keeps the matrix pipe full (measured: 14 vs 7 cycles per instruction of
// the split), which is why a staged halo element has to feed 64 output channels (NTL = 2) wherever the layer has them.
#include <algorithm>
#include <type_traits>
#include <vector>

#include <hip/hip_ext.h>

#include "ws_common.hpp"

namespace rfi {
namespace {

using namespace ws;

struct WsDev {
    const float* x;                   // [N][H][W][x_ps] float32
    int x_ps;
    unsigned x_bytes;                 // bytes of the tensor (< 2^31: an offset of 2^31 reads as zero)
    const float* scale;               // InXform of the input (null: identity)
    const float* shift;
    float slope;
    int N, H, W, Cin, Cout;
    const bf16_t* wB;                 // [tap][kc][cb][3 planes][64 lanes][8] bf16 (planes.hpp)
    int nkc, ncb;
    const float* bias;
    float* y;                         // [N][H][W][y_ps] float32
    int y_ps;
    double* stats;                    // [gridDim.x][Cout][2] fp64 (sum y, sum y^2) records, or null
    int wide_epi;                     // 64-channel blocks: 16-byte output stores through an LDS transpose (y_ps % 4 == 0, 16-byte aligned y)
    int diag;                         // RFI_WS_DIAG (timing experiments, wrong results): 1 no halo loads after the first, 2 no filter DMA, 4 no MFMAs
    int prio;                         // RFI_WS_PRIO (tuning): 0 none, 1 consumers at priority 2, 2 producers at priority 1
    unsigned long long* stamps;       // RFI_DIAG_STAMPS build: per-wave cycle sums
};

// bytes per halo pixel in LDS: the P planes of a 16-channel chunk (32 B each) + 16 B of padding (odd 16-byte-slot stride:
// the ds_read_b128 of 32 consecutive pixels is conflict free)
template <int P> constexpr int ROWB = 32 * P + 16;

// ADB: two halo buffers (where the LDS has room: 32-channel blocks) -- the producers write item q + 1 while the consumers
// read item q, one barrier per item; otherwise ONE halo buffer, written between two barriers of an item
template <int TB, int TH, int TW, int NTL, bool ADB, int P>
struct WsCfg {
    static constexpr int HH = TH + 2, HW = TW + 2, HPI = HH * HW, HP = TB * HPI;
    static constexpr int A_BYTES = (HP * ROWB<P> + 1023) & ~1023;
    static constexpr int NPIECE = 9 * NTL * P;                       // 1-KiB pieces of a filter chunk: 9 taps x NTL blocks x P planes
    static constexpr int B_BYTES = NPIECE * 1024;
    static constexpr int B_ITEMS = (NPIECE + 3) / 4;                 // pieces per producer wave
    static constexpr int HALO_ITEMS = (HP * 4 + 255) / 256;          // float4 loads per producer thread and item
    static constexpr int B_OFF = (ADB ? 2 : 1) * A_BYTES;
    static constexpr int STAT_OFF = B_OFF + 2 * B_BYTES;
    static constexpr int STAT_BYTES = 4 * NTL * 64 * 2 * 8;            // per consumer lane and n-block (sum, sumsq): written once, at the end
    // the immediate epilogue (NTL >= 2) transposes half blocks through a per-wave scratch of 16 pixel rows x 36 floats into
    // 16-byte stores; the scratch aliases the statistics area (free until the kernel's last lines)
    static constexpr int EPI_WAVE_BYTES = 16 * 36 * 4;
    static constexpr bool WIDE_OK = STAT_OFF + 4 * EPI_WAVE_BYTES <= 160 * 1024;       // (not beside the 4 x 10 x 10 halo of the 8 x 8 tiles)
    static constexpr int LDS_BYTES = STAT_OFF + ((STAT_BYTES > 4 * EPI_WAVE_BYTES || !WIDE_OK) ? STAT_BYTES : 4 * EPI_WAVE_BYTES);
    static_assert(TB * TH * TW == 256, "a tile is 256 output pixels");
    static_assert(32 % TW == 0 || TW % 32 == 0, "a 32-pixel block covers whole rows or a part of one");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};


// XF: the load transform -- 0 none, 1 relu(x * scale + shift), 2 x * scale + shift followed by max(v, v * slope) (LeakyReLU;
// slope 1: no activation)
// P: 3 float32 by three bf16 pieces (six MFMAs per block product); 1 bf16 operands (the float32-tensor bf16 mode: values
// rounded once, RNE, at staging; one MFMA per block product)
template <int TB, int TH, int TW, int NTL, int XF, bool ADB, int P>
__global__ __launch_bounds__(512) void conv_ws_kernel(WsDev d) {
    using C = WsCfg<TB, TH, TW, NTL, ADB, P>;
    constexpr int RB = ROWB<P>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef RFI_DIAG_STAMPS
    unsigned long long st_[4] = {0, 0, 0, 0};
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- persistent workgroup: blocks b and b + 8 share an XCD (round-robin dispatch), so each XCD label gets one
    // contiguous range of tiles and its workgroups stride through it (neighbouring tiles share halos in one L2)
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH, tiles_b = (d.N + TB - 1) / TB;
    const int ntiles = tiles_b * tiles_y * tiles_x;
    const int GX = gridDim.x;
    int t_begin, t_count, j, gx;
    if (GX >= 8 && (GX & 7) == 0) {
        const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
        j = blockIdx.x >> 3;
        gx = GX >> 3;
        t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_count = q8 + (xcd < r8 ? 1 : 0);
    } else {
        j = blockIdx.x; gx = GX; t_begin = 0; t_count = ntiles;
    }
    const int my_tiles = (j < t_count) ? (t_count - j + gx - 1) / gx : 0;
    const int cb0 = blockIdx.y * NTL, n0 = cb0 * 32;
    double* const st_out = d.stats ? d.stats + ((size_t)blockIdx.x * d.Cout + n0) * 2 : nullptr;
    if (my_tiles == 0) {                             // uniform for the workgroup, before any barrier
        if (st_out && tid < 32 * NTL && n0 + tid < d.Cout) st_out[tid * 2] = st_out[tid * 2 + 1] = 0.0;
        return;
    }
    struct Tile { int n, oy0, ox0; };
    auto tile_of = [&](int k) {
        const int t = t_begin + j + k * gx;
        Tile r;
        r.ox0 = (t % tiles_x) * TW;
        r.oy0 = ((t / tiles_x) % tiles_y) * TH;
        r.n = (t / (tiles_x * tiles_y)) * TB;
        return r;
    };
    const int nkc = d.nkc;
    const int nitems = my_tiles * nkc;
    double* const s_stat = reinterpret_cast<double*>(smem + C::STAT_OFF);

    if (wave >= 4) {
        // =============================================================== producers
        const int ptid = tid - 256, pw = wave - 4;
        const unsigned q4b = (unsigned)(ptid & 3) * 16u;  // byte offset of this thread's float4 in a 16-channel chunk
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)d.x_bytes, 0x00020000);
        // tile-independent halo coordinates of every item: image << 24 | row << 12 | col (invalid: row 0xfff)
        unsigned h_rc[C::HALO_ITEMS];
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it) {
            const int idx = ptid + it * 256, pix = idx >> 2;
            const int b = pix / C::HPI, rem = pix % C::HPI;
            h_rc[it] = idx < C::HP * 4 ? ((unsigned)b << 24) | ((unsigned)(rem / C::HW) << 12) | (unsigned)(rem % C::HW) : (0xfffu << 12);
        }
        unsigned voff[C::HALO_ITEMS];                    // byte offset of the pixel's chunk-0 float4; 2^31: outside the image (reads 0)
        unsigned vmsk[C::HALO_ITEMS];                    // XF: ~0 inside the image, 0 outside (zero padding AFTER the transform)
        auto setup_halo = [&](const Tile& t) {
#pragma unroll
            for (int it = 0; it < C::HALO_ITEMS; ++it) {
                const int n = t.n + (int)(h_rc[it] >> 24);
                const int iy = t.oy0 - 1 + (int)((h_rc[it] >> 12) & 0xfff), ix = t.ox0 - 1 + (int)(h_rc[it] & 0xfff);
                const bool ok = n < d.N && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                voff[it] = ok ? (unsigned)(((n * d.H + iy) * d.W + ix) * d.x_ps) * 4u + q4b : 0x80000000u;
                vmsk[it] = ok ? 0xffffffffu : 0u;
            }
        };
        u32x4 raw[C::HALO_ITEMS];                        // the item being loaded
        f32x4 screg = {1.f, 1.f, 1.f, 1.f}, shreg = {0.f, 0.f, 0.f, 0.f};
        unsigned pl[C::HALO_ITEMS][2 * P];               // its P bf16 planes (4 channels each), ready for LDS
        // the loads of an item issue back to back, the coefficient loads first: the youngest HALO_ITEMS vector-memory
        // operations of the wave are then the halo loads, which the vmcnt arithmetic below relies on
        auto issue_loads = [&](int c0) {
            if constexpr (XF != 0) {
                screg = *reinterpret_cast<const f32x4*>(d.scale + c0 + (ptid & 3) * 4);
                shreg = *reinterpret_cast<const f32x4*>(d.shift + c0 + (ptid & 3) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < C::HALO_ITEMS; ++it) raw[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff[it], c0 * 4, 0);
        };
        auto split_all = [&]() {
#pragma unroll
            for (int it = 0; it < C::HALO_ITEMS; ++it) {
                f32x4 v = __builtin_bit_cast(f32x4, raw[it]);
                if constexpr (XF != 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e] * screg[e] + shreg[e];     // (unfused, -ffp-contract=off: the same values every other consumer of this tensor computes)
                        if constexpr (XF == 1) asm("v_max_f32 %0, 0, %1" : "=v"(t) : "v"(t));
                        else t = __builtin_fmaxf(t, t * d.slope);
                        v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & vmsk[it]);
                    }
                }
                if constexpr (P == 3) {
                    split_pair(v.x, v.y, pl[it][0], pl[it][2], pl[it][4]);
                    split_pair(v.z, v.w, pl[it][1], pl[it][3], pl[it][5]);
                } else {
                    pl[it][0] = cvt_pair(v.x, v.y);
                    pl[it][1] = cvt_pair(v.z, v.w);
                }
            }
        };
        auto write_all = [&](int buf) {
            unsigned char* const sA = smem + buf * C::A_BYTES + (ptid >> 2) * RB + (ptid & 3) * 8;
#pragma unroll
            for (int it = 0; it < C::HALO_ITEMS; ++it) {
                if (ptid + it * 256 < C::HP * 4) {
                    unsigned char* row = sA + it * 64 * RB;
#pragma unroll
                    for (int p = 0; p < P; ++p) *reinterpret_cast<u32x2*>(row + 32 * p) = u32x2{pl[it][2 * p], pl[it][2 * p + 1]};
                }
            }
        };
        // filter chunk kc: piece p = (tap * NTL + nt) * 3 + plane is the 1 KiB block a consumer wave reads as one fragment;
        // in wB the NTL x 3 pieces of a tap are contiguous.  A block beyond the last one (odd block count) re-reads the
        // last valid block: its products land in channels that are never stored
        const unsigned char* const wb = reinterpret_cast<const unsigned char*>(d.wB);
        auto issue_B = [&](int kc, int buf) {
            unsigned char* const sB = smem + C::B_OFF + buf * C::B_BYTES;
#pragma unroll
            for (int i = 0; i < C::B_ITEMS; ++i) {
                const int p = pw + 4 * i;
                if (p < C::NPIECE) {
                    const int tap = p / (P * NTL), r = p - tap * (P * NTL), nt = r / P, plane = r - nt * P;
                    const int cb = cb0 + nt < d.ncb ? cb0 + nt : d.ncb - 1;
                    const unsigned off = (unsigned)((((tap * nkc + kc) * d.ncb + cb) * P + plane) * 1024 + lane * 16);
                    __builtin_amdgcn_global_load_lds((gbl_void*)(wb + off), (lds_void*)(sB + p * 1024), 16, 0, 0);
                }
            }
        };
        if (d.prio == 2) __builtin_amdgcn_s_setprio(1);
        // load cursor: item (lk, lch)
        int lk = 0, lch = 0;
        Tile lt = tile_of(0);
        setup_halo(lt);
        issue_loads(0);
        // q = -1 is the prologue (item 0); every later iteration stages item q + 1 while the consumers multiply item q
        for (int q = -1; q < nitems; ++q) {
            WS_T(t0);
            const bool more = q + 1 < nitems;
            if (more) {
                split_all();                             // (waits for the loads of item q + 1, issued an iteration ago)
                WS_T(t1);
                WS_ACC(2, t0, t1);
                if constexpr (ADB) write_all((q + 1) & 1);         // the consumers left this buffer an item ago
                __builtin_amdgcn_sched_barrier(0);
                if (!(d.diag & 2)) issue_B(lch, (q + 1) & 1);      // its buffer was last read for item q - 1
                __builtin_amdgcn_sched_barrier(0);
                if (q + 2 < nitems) {
                    if (++lch == nkc) {
                        lch = 0;
                        ++lk;
                        lt = tile_of(lk);
                        setup_halo(lt);
                    }
                }
                if (!(d.diag & 1)) issue_loads(lch * 16);          // (the very last item is re-read once: harmless)
                __builtin_amdgcn_sched_barrier(0);
                // the DMA pieces were issued BEFORE the halo loads: all but the youngest HALO_ITEMS operations done = every
                // piece has landed
                WS_T(t2);
                wait_vmcnt<C::HALO_ITEMS>();
                if constexpr (ADB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                WS_T(t3);
                WS_ACC(3, t2, t3);
            }
            WS_T(t4);
            wg_barrier();                                // the consumers are done with item q (ADB: and item q + 1 is staged)
            if constexpr (!ADB) {
                if (more) write_all(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                wg_barrier();                            // item q + 1 is staged
            }
            WS_T(t5);
            WS_ACC(0, t0, t4);
            WS_ACC(1, t4, t5);
        }
    } else {
        // =============================================================== consumers
        if (d.prio == 1) __builtin_amdgcn_s_setprio(2);
        const int cw = wave, li = lane & 31, lh = lane >> 5;
        int a_base[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int p = cw * 64 + mt * 32 + li;
            const int b = p / (TH * TW), r = p % (TH * TW);
            a_base[mt] = ((b * C::HH + r / TW) * C::HW + r % TW) * RB + lh * 16;
        }
        const int b_base = lane * 16;
        float bias[NTL];
        unsigned lane_off[NTL];                          // this lane's part of a store address: 4 lh pixels on + its channel
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int co = n0 + nt * 32 + li;
            bias[nt] = (d.bias && co < d.Cout) ? d.bias[co] : 0.0f;
            asm volatile("" : "+v"(bias[nt]));
            lane_off[nt] = (unsigned)(4 * lh * d.y_ps + co);
        }
        // acc: the tile being multiplied; outr: the finished tile whose epilogue is still owed.  C/D layout of 32x32:
        // col = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (pixel of the block): one wave
        // store covers the same 32 channels of two pixels = two 128-byte runs
        // (DEFER: where the registers allow it -- 32-channel blocks; 64-channel blocks run the epilogue from acc at once)
        constexpr bool DEFER = NTL == 1;
        f32x16 acc[2][NTL], outr[2][NTL];          // (outr: never touched, hence no registers, without DEFER)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
        double d1[NTL], d2[NTL];                         // this lane's share of sum y, sum y^2 (its channels, its pixels)
        float s1[NTL], s2[NTL];                          // ... of the tile whose epilogue is running
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) { d1[nt] = d2[nt] = 0.0; s1[nt] = s2[nt] = 0.0f; }
        constexpr int E = 32 * NTL;                      // accumulator registers (= stores) per tile and lane
        // element e of a FULL tile (inside the image, all channels): no bounds tests, address = a wave-uniform pixel row
        // + the lane's constant part
        auto epi_elem = [&](const f32x16 (&src)[2][NTL], int e, unsigned tile_off) {
            const int nt = e / 32, mt = (e >> 4) & 1, r = e & 15;
            const int p = cw * 64 + mt * 32 + (r & 3) + 8 * (r >> 2);       // (+ 4 lh: the same image row)
            const int b = p / (TH * TW), rr = p % (TH * TW);
            const unsigned uoff = tile_off + (unsigned)(((b * d.H + rr / TW) * d.W + rr % TW) * d.y_ps);
            const float v = src[mt][nt][r] + bias[nt];
            d.y[uoff + lane_off[nt]] = v;
            s1[nt] += v;
            s2[nt] += v * v;
        };
        auto epi_generic = [&](const f32x16 (&src)[2][NTL], const Tile& t) {         // any tile: every element behind its bounds tests
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int co = n0 + nt * 32 + li;
                const bool cok = co < d.Cout;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = cw * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int b = p / (TH * TW), rr = p % (TH * TW);
                        const int n = t.n + b, oy = t.oy0 + rr / TW, ox = t.ox0 + rr % TW;
                        const float v = src[mt][nt][r] + bias[nt];
                        if (cok && n < d.N && oy < d.H && ox < d.W) {
                            d.y[(unsigned)(((n * d.H + oy) * d.W + ox) * d.y_ps + co)] = v;
                            s1[nt] += v;
                            s2[nt] += v * v;
                        }
                    }
                }
            }
        };
        auto epi_finish = [&]() {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                d1[nt] += (double)s1[nt];
                d2[nt] += (double)s2[nt];
                s1[nt] = s2[nt] = 0.0f;
            }
        };
        bool pend = false;                               // outr holds a finished tile (pt)
        Tile pt = tile_of(0);
        // one (tile, chunk) item.  PEND: the epilogue of the previous tile (a FULL one) is spread over the nine tap
        // steps, its stores and sums issuing between the MFMAs instead of in front of them
        // FIRST: chunk 0 of a tile -- the first MFMA of every block takes C = 0 (no zeroing of the accumulators)
        auto run_item = [&](auto pend_c, auto first_c, const unsigned char* sA, const unsigned char* sB) {
            constexpr bool PEND = decltype(pend_c)::value, FIRST = decltype(first_c)::value;
            const unsigned tile_off = (unsigned)(((pt.n * d.H + pt.oy0) * d.W + pt.ox0) * d.y_ps);
            // software pipeline over the taps: the fragments of tap t + 1 are read BEFORE the MFMAs of tap t issue
            bf16x8 afr[2][2][P], bfr[2][NTL][P];
            auto load_frags = [&](int tap, bf16x8 (&af)[2][P], bf16x8 (&bf)[NTL][P]) {
                const int tr = tap / 3, ts = tap % 3;
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                    for (int p = 0; p < P; ++p) bf[nt][p] = *reinterpret_cast<const bf16x8*>(sB + ((tap * NTL + nt) * P + p) * 1024);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int p = 0; p < P; ++p)
                        af[mt][p] = *reinterpret_cast<const bf16x8*>(sA + a_base[mt] + (tr * C::HW + ts) * RB + p * 32);
            };
            load_frags(0, afr[0], bfr[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) load_frags(tap + 1, afr[(tap + 1) & 1], bfr[(tap + 1) & 1]);
                if constexpr (!ADB) {
                    if (tap == 8) {
                        // one halo buffer: the last LDS reads of the item (the fragments of tap 8) were issued under tap 7's
                        // MFMAs; once they have landed the producers may overwrite the halo tile -- they do so UNDER the
                        // MFMAs of tap 8 instead of in a window in which the matrix pipe idles
                        __builtin_amdgcn_sched_barrier(0);
                        WS_T(tb0);
                        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
                        wg_barrier();
                        WS_T(tb1);
                        WS_ACC(3, tb0, tb1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTL; ++nt) {
                        f32x16 c = acc[mt][nt];
                        if (FIRST && tap == 0)
#pragma unroll
                            for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                        if constexpr (P == 3) acc[mt][nt] = mma3(afr[tap & 1][mt], bfr[tap & 1][nt], c);
                        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[tap & 1][mt][0], bfr[tap & 1][nt][0], c, 0, 0, 0);
                    }
                if constexpr (PEND && DEFER) {
#pragma unroll
                    for (int k = 0; k < (E + 8) / 9; ++k) {
                        const int e = tap * E / 9 + k;
                        if (e < (tap + 1) * E / 9) epi_elem(outr, e, tile_off);
                    }
                }
                // issue order of this step: the LDS reads of tap t + 1 (and the pending epilogue's stores and sums) go BETWEEN
                // the MFMAs of tap t, one or two per MFMA -- issued as one burst in front of them they leave the matrix pipe
                // idle for most of the burst (12 reads ~ 100 cycles against one 32-cycle MFMA in flight)
                constexpr int NM = 2 * NTL * (P == 3 ? 6 : 1), NR = P * (2 + NTL);
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // one MFMA
                    if (tap + 1 < 9) {                   // the LDS reads that fall to this MFMA (the builtin wants literals)
                        const int cnt = (i + 1) * NR / NM - i * NR / NM;
                        if (cnt == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        else if (cnt == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else if (cnt == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        else if (cnt >= 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    }
                    if constexpr (PEND && DEFER) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                  // two vector ALU instructions
                        if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // a store every third MFMA
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int ck = 0, cch = 0;                             // compute cursor: tile index, chunk
        Tile ct = tile_of(0);
        wg_barrier();
        if constexpr (!ADB) wg_barrier();                // item 0 is staged
        for (int q = 0; q < nitems; ++q) {
            WS_T(t0);
            const unsigned char* const sA = smem + (ADB ? (q & 1) * C::A_BYTES : 0);
            const unsigned char* const sB = smem + C::B_OFF + (q & 1) * C::B_BYTES + b_base;
            if constexpr (DEFER) {
                if (pend && !(pt.n + TB <= d.N && pt.oy0 + TH <= d.H && pt.ox0 + TW <= d.W && n0 + 32 * NTL <= d.Cout)) {
                    epi_generic(outr, pt);               // a ragged tile: its epilogue in one piece, in front of the item
                    epi_finish();
                    pend = false;
                }
            }
            if (d.diag & 4) {
                if constexpr (!ADB) wg_barrier();
            } else {
                if (pend) {                              // (pend implies chunk 0 of the next tile)
                    run_item(std::true_type{}, std::true_type{}, sA, sB);
                    epi_finish();
                    pend = false;
                } else if (cch == 0) {
                    run_item(std::false_type{}, std::true_type{}, sA, sB);
                } else {
                    run_item(std::false_type{}, std::false_type{}, sA, sB);
                }
            }
            WS_T(t1);
            if (++cch == nkc) {                          // the tile is complete
                if constexpr (DEFER) {                   // its results move to outr, the epilogue follows under the next item's
#pragma unroll                                           // MFMAs (after the loop for the last tile)
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NTL; ++nt) outr[mt][nt] = acc[mt][nt];
                    pt = ct;
                    pend = true;
                } else {
                    if (ct.n + TB <= d.N && ct.oy0 + TH <= d.H && ct.ox0 + TW <= d.W && n0 + 32 * NTL <= d.Cout) {
                        // (no MFMA of this wave is in flight here: two elements per packed fp32 instruction)
                        const unsigned tile_off = (unsigned)(((ct.n * d.H + ct.oy0) * d.W + ct.ox0) * d.y_ps);
                        f32x2 s1v[NTL], s2v[NTL];
#pragma unroll
                        for (int nt = 0; nt < NTL; ++nt) s1v[nt] = s2v[nt] = f32x2{0.0f, 0.0f};
                        // WIDE: the values leave through this wave's LDS scratch as 16-byte stores of 4 channels of one pixel (a
                        // wave store = 8 pixels x 128 bytes): 16 store instructions per tile and lane instead of 64 -- the epilogue
                        // is bound by store ISSUE (~70 cycles each).  Same values, same statistics arithmetic, another route out
                        // (compile-time for the float32 64-channel blocks only: in the 128-channel bf16-operand kernels the second
                        // route costs registers -- 336 bytes of scratch per lane, measured 2 % slower on the ResNet-encoder model)
                        constexpr bool WIDE_CT = C::WIDE_OK && NTL == 2 && P == 3;
                        const bool wide = WIDE_CT && d.wide_epi;
                        float* const s_ep = reinterpret_cast<float*>(smem + C::STAT_OFF + cw * C::EPI_WAVE_BYTES);
                        if (WIDE_CT && wide) {
                            // quarter blocks (4 accumulator registers = 8 pixels x 32 channels) alternate between the two halves
                            // of the scratch: the 16-byte read of quarter qi - 1 is issued in front of quarter qi's arithmetic
                            // and LDS writes and its global store behind them, so no LDS round trip is waited for
                            constexpr int NQ = 8 * NTL;
                            const int pl = lane >> 3, c4 = (lane & 7) * 4;
                            f32x4 rq = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                            for (int qi = 0; qi <= NQ; ++qi) {
                                if (qi > 0) rq = *reinterpret_cast<const f32x4*>(s_ep + ((qi - 1) & 1) * 288 + pl * 36 + c4);
                                if (qi < NQ) {
                                    const int nt = qi >> 3, mt = (qi >> 2) & 1, r0 = (qi & 3) * 4;
#pragma unroll
                                    for (int k = 0; k < 4; k += 2) {
                                        const f32x2 v = f32x2{acc[mt][nt][r0 + k], acc[mt][nt][r0 + k + 1]} + f32x2{bias[nt], bias[nt]};
                                        s_ep[(qi & 1) * 288 + (k + 4 * lh) * 36 + li] = v[0];
                                        s_ep[(qi & 1) * 288 + (k + 1 + 4 * lh) * 36 + li] = v[1];
                                        s1v[nt] += v;
                                        s2v[nt] = __builtin_elementwise_fma(v, v, s2v[nt]);
                                    }
                                }
                                if (qi > 0) {
                                    const int pq = qi - 1, nt = pq >> 3, mt = (pq >> 2) & 1;
                                    const int p = cw * 64 + mt * 32 + (pq & 3) * 8 + pl;
                                    const int b = p / (TH * TW), rr = p % (TH * TW);
                                    *reinterpret_cast<f32x4*>(d.y + tile_off + (unsigned)(((b * d.H + rr / TW) * d.W + rr % TW) * d.y_ps) +
                                                              (unsigned)(n0 + nt * 32 + c4)) = rq;
                                }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < E; e += 2) {
                                const int nt = e / 32, mt = (e >> 4) & 1, r = e & 15;
                                const f32x2 v = f32x2{acc[mt][nt][r], acc[mt][nt][r + 1]} + f32x2{bias[nt], bias[nt]};
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    const int p = cw * 64 + mt * 32 + ((r + k) & 3) + 8 * ((r + k) >> 2);
                                    const int b = p / (TH * TW), rr = p % (TH * TW);
                                    d.y[tile_off + (unsigned)(((b * d.H + rr / TW) * d.W + rr % TW) * d.y_ps) + lane_off[nt]] = v[k];
                                }
                                s1v[nt] += v;
                                s2v[nt] = __builtin_elementwise_fma(v, v, s2v[nt]);
                            }
                        }
#pragma unroll
                        for (int nt = 0; nt < NTL; ++nt) {
                            s1[nt] += s1v[nt].x + s1v[nt].y;
                            s2[nt] += s2v[nt].x + s2v[nt].y;
                        }
                    } else {
                        epi_generic(acc, ct);
                    }
                    epi_finish();
                }
                cch = 0;
                ++ck;
                if (ck < my_tiles) ct = tile_of(ck);
            }
            WS_T(t4);
            // ADB: every consumer is done with this item's halo tile and item q + 1 is staged in the other one.  One buffer:
            // the "done" barrier sits in front of tap 8 (run_item); this one says that item q + 1 is staged
            wg_barrier();
            WS_T(t5);
            WS_ACC(0, t0, t1);
            WS_ACC(2, t1, t4);
            WS_ACC(1, t4, t5);
        }
        if constexpr (DEFER) {
            if (pend) {
                epi_generic(outr, pt);
                epi_finish();
            }
        }
        if (st_out) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                s_stat[((cw * NTL + nt) * 64 + lane) * 2] = d1[nt];
                s_stat[((cw * NTL + nt) * 64 + lane) * 2 + 1] = d2[nt];
            }
        }
    }
#ifdef RFI_DIAG_STAMPS
    if (d.stamps && lane == 0) {
        unsigned long long* o = d.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
        for (int i = 0; i < 4; ++i) o[i] = st_[i];
        o[4] = (unsigned long long)nitems;
    }
#endif
    if (st_out) {
        __syncthreads();
        if (tid < 32 * NTL && n0 + tid < d.Cout) {       // channel tid: lanes cl and cl + 32 of the four consumer waves
            const int nt = tid >> 5, cl = tid & 31;
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    t1 += s_stat[((w * NTL + nt) * 64 + h * 32 + cl) * 2];
                    t2 += s_stat[((w * NTL + nt) * 64 + h * 32 + cl) * 2 + 1];
                }
            st_out[tid * 2] = t1;
            st_out[tid * 2 + 1] = t2;
        }
    }
}

template <int TB, int TH, int TW, int NTL, int XF, int P>
void launch_ws(rfi_ctx* ctx, ConvArgs& a, WsDev& d) {
    constexpr bool ADB = NTL == 1 || P == 1;     // two halo buffers wherever the LDS has room
    using C = WsCfg<TB, TH, TW, NTL, ADB, P>;
    const int ntiles = (int)(cdiv(a.N, TB) * cdiv(a.H, TH) * cdiv(a.W, TW));
    const int ycols = (int)cdiv(d.ncb, NTL);
    // one workgroup per CU: about 256 in total, a multiple of 8 along x, tiles spread evenly over the workgroups of
    // each XCD label
    const int gmax = std::max(8, 256 / ycols);
    const int tx = (int)cdiv(ntiles, 8);
    const int per = (int)cdiv(tx, std::max(1, gmax / 8));
    int GX = 8 * (int)cdiv(tx, per);
    if (ntiles < 8) GX = ntiles;
    if (a.stats && GX <= a.stats_max_records) a.stats_records = GX;
    else { a.stats = nullptr; a.stats_records = 0; }
    d.stats = a.stats;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<TB, TH, TW, NTL, XF, ADB, P>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    });
#ifdef RFI_DIAG_STAMPS
    {
        const size_t nw = (size_t)GX * ycols * 8;
        RFI_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d.stamps), nw * 64));
        RFI_CHECK_HIP(hipMemsetAsync(d.stamps, 0, nw * 64, ctx->stream));
        hipLaunchKernelGGL((conv_ws_kernel<TB, TH, TW, NTL, XF, ADB, P>), dim3(GX, ycols), dim3(512), C::LDS_BYTES, ctx->stream, d);
        std::vector<unsigned long long> hs(nw * 8);
        RFI_CHECK_HIP(hipMemcpyAsync(hs.data(), d.stamps, nw * 64, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        double c[5] = {0, 0, 0, 0, 0}, p[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < 5; ++i) ((w & 7) < 4 ? c : p)[i] += (double)hs[w * 8 + i];
        std::fprintf(stderr, "[stamps] conv_ws<%d,%d,%d,%d,%d> N%d %dx%d %d->%d grid %dx%d items/wg %.1f | cycles per item: consumer mfma %.0f "
                     "epilogue %.0f barriers %.0f (+ in front of tap 8: %.0f) | producer work %.0f (split %.0f, dma wait %.0f) barriers+write %.0f\n", TB, TH, TW, NTL, XF,
                     a.N, a.H, a.W, a.Cin, a.Cout, GX, ycols, c[4] / (nw / 2), c[0] / c[4], c[2] / c[4], c[1] / c[4], c[3] / c[4], p[0] / p[4],
                     p[2] / p[4], p[3] / p[4], p[1] / p[4]);
        RFI_CHECK_HIP(hipFree(d.stamps));
        d.stamps = nullptr;
        return;
    }
#endif
    if (a.done) {
        hipExtLaunchKernelGGL((conv_ws_kernel<TB, TH, TW, NTL, XF, ADB, P>), dim3(GX, ycols), dim3(512), C::LDS_BYTES, ctx->stream, nullptr,
                              a.done, 0, d);
        a.done_used = true;
    } else {
        hipLaunchKernelGGL((conv_ws_kernel<TB, TH, TW, NTL, XF, ADB, P>), dim3(GX, ycols), dim3(512), C::LDS_BYTES, ctx->stream, d);
    }
    check_launch("conv_ws");
}

template <int NTL, int XF, int P>
void dispatch_ws(rfi_ctx* ctx, ConvArgs& a, WsDev& d) {
    if (a.W >= 32) launch_ws<1, 8, 32, NTL, XF, P>(ctx, a, d);
    else if (a.W >= 16) launch_ws<1, 16, 16, NTL, XF, P>(ctx, a, d);
    else launch_ws<4, 8, 8, NTL, XF, P>(ctx, a, d);
}
template <int NTL, int P>
void dispatch_xf(rfi_ctx* ctx, ConvArgs& a, WsDev& d, int xf) {
    if (xf == 0) dispatch_ws<NTL, 0, P>(ctx, a, d);
    else if (xf == 1) dispatch_ws<NTL, 1, P>(ctx, a, d);
    else dispatch_ws<NTL, 2, P>(ctx, a, d);
}

}  // namespace

bool conv_ws_eligible(const ConvArgs& a) {
    if (!(a.R == 3 && a.S == 1 && a.pad == 1 && a.zgroups == 1 && a.fold == 0)) return false;
    if (a.Hin != a.H || a.Win != a.W || a.Hout != a.H || a.Wout != a.W) return false;
    if (a.osy != 1 || a.osx != 1 || a.ooy != 0 || a.oox != 0) return false;
    if (a.y16 || a.bwd_y || a.W < 8 || a.H < 8) return false;
    if (a.Cin % 16 != 0 || a.x.pstride % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x.p) & 15)) return false;
    if (a.xf.scale && ((reinterpret_cast<uintptr_t>(a.xf.scale) & 15) || (reinterpret_cast<uintptr_t>(a.xf.shift) & 15))) return false;
    // 32-bit byte offsets with 2^31 as the "reads zero" offset of the buffer loads; 32-bit element offsets of the output
    if ((int64_t)a.N * a.H * a.W * a.x.pstride * 4 >= (int64_t)1 << 31 || (int64_t)a.N * a.H * a.W * a.y.pstride >= (int64_t)1 << 31) return false;
    return true;
}

// wB: the layer's filters in B-operand order with P planes (launch_weights_to_wb; one K segment of Cin channels).
// P = 3: the float32-by-3xbf16 arithmetic; P = 1: bf16 operands (ConvArgs::bf16 on float32 tensors)
void launch_conv_ws(rfi_ctx* ctx, ConvArgs& a, const bf16_t* wB, int P) {
    RFI_REQUIRE(conv_ws_eligible(a), "conv_ws: shape not eligible");
    RFI_REQUIRE(P == 1 || P == 3, "conv_ws: planes must be 1 or 3");
    WsDev d;
    d.x = a.x.p; d.x_ps = a.x.pstride;
    d.x_bytes = (unsigned)((int64_t)a.N * a.H * a.W * a.x.pstride * 4);
    d.scale = a.xf.scale; d.shift = a.xf.shift;
    d.slope = a.xf.relu == 0 ? 1.0f : a.xf.slope;
    d.N = a.N; d.H = a.H; d.W = a.W; d.Cin = a.Cin; d.Cout = a.Cout;
    d.wB = wB;
    d.nkc = plane_chunks(a.Cin); d.ncb = (a.Cout + 31) / 32;
    RFI_REQUIRE((int64_t)wb_elems(9, a.Cout, a.Cin, 0, P) * 2 < ((int64_t)1 << 32), "conv_ws: filter tensor too large");
    d.bias = a.bias;
    d.y = a.y.p; d.y_ps = a.y.pstride;
    d.stats = nullptr;
    d.stamps = nullptr;
    static const int prio = getenv("RFI_WS_PRIO") ? atoi(getenv("RFI_WS_PRIO")) : 0;
    d.prio = prio;
    static const int diag = getenv("RFI_WS_DIAG") ? atoi(getenv("RFI_WS_DIAG")) : 0;
    d.diag = diag;
    static const bool no_wide = getenv("RFI_WS_NARROW_EPI") != nullptr;          // A/B runs: dword stores from the accumulators
    d.wide_epi = !no_wide && a.y.pstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.y.p) & 15) == 0;
    const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cout * 9 * a.Cin;
    std::string label;
    if (ctx->profiling)
        label = "conv_ws N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" + std::to_string(a.W) + " " +
                std::to_string(a.Cin) + "->" + std::to_string(a.Cout) + (a.xf.scale ? " xf" : "") + (P == 3 ? " 3xbf16" : " bf16");
    const double bytes = 4.0 * ((double)a.N * a.H * a.W * a.Cin + 9.0 * a.Cin * a.Cout) + 4.0 * a.N * a.H * a.W * a.Cout;
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, bytes, label);
    static const int ntl_max = getenv("RFI_WS_NTL") ? atoi(getenv("RFI_WS_NTL")) : 4;          // A/B runs
    const int xf = !a.xf.scale ? 0 : (a.xf.relu == 1 || (a.xf.relu == 2 && a.xf.slope == 0.0f)) ? 1 : 2;
    // wider channel blocks cut the producers' work per MFMA (64 channels for the split arithmetic; 128 where one MFMA
    // stands for a block product); they are used where the grid still covers the chip (tiles x blocks >= ~256 workgroups)
    const int64_t tiles = (a.W >= 16 ? (int64_t)a.N : cdiv(a.N, 4)) * cdiv(a.H, a.W >= 32 ? 8 : a.W >= 16 ? 16 : 8) * cdiv(a.W, a.W >= 32 ? 32 : a.W >= 16 ? 16 : 8);
    int ntl = 1;
    if (a.Cout > 32 && ntl_max >= 2 && tiles * cdiv(d.ncb, 2) >= 224) ntl = 2;
    if (P == 1 && a.Cout > 64 && ntl_max >= 4 && tiles * cdiv(d.ncb, 4) >= 224) ntl = 4;
    if (P == 3) {
        if (ntl == 2) dispatch_xf<2, 3>(ctx, a, d, xf);
        else dispatch_xf<1, 3>(ctx, a, d, xf);
    } else {
        if (ntl == 4) dispatch_xf<4, 1>(ctx, a, d, xf);
        else if (ntl == 2) dispatch_xf<2, 1>(ctx, a, d, xf);
        else dispatch_xf<1, 1>(ctx, a, d, xf);
    }
}

}  // namespace rfi
