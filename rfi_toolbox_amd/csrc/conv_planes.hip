// NHWC implicit-GEMM "conv-like" contraction on PLANE tensors (planes.hpp) for the gfx950 matrix cores.
//
//   y[n,oy,ox,co] = bias[co] + sum_{tap=(r,s)} sum_ci X[n, oy*S+r-pad, ox*S+s-pad, ci] * W[tap][co][ci]
//
// GEMM view: M = N*H*W output pixels, N = Cout, K = taps * Cin, computed with v_mfma_f32_32x32x16_bf16:
//   P = 1  bfloat16 activations and filters, float32 accumulate (one MFMA per 32x32x16 block product);
//   P = 3  float32 carried as three bf16 pieces (h, m, l): six MFMAs per block product, small terms first
//          (l*h, m*m, h*l, m*h, h*m, h*h; the three products below 2^-24 of a*b are dropped) -- float32-level
//          accuracy at 2.7x the native float32 MFMA rate.
// Both operands arrive in the layout the instruction wants, so staging is pure data movement:
//   * the input HALO tile ((TH*S+R-S) x (TW*S+R-S) pixels x one 16-channel chunk x P planes) goes HBM/L2 -> LDS
//     by LDS-DMA (global_load_lds_dwordx4): no registers, no VALU apart from one 32-bit offset per 16-byte
//     piece; halo pixels outside the image point at a zero area at the end of the tensor.  A pixel row in LDS
//     is 2P data slots + 1 pad slot of 16 bytes (odd slot stride: the ds_read_b128 of 64 consecutive pixels is
//     conflict free); every tap reads the same tile at a shifted pixel offset, so the input crosses L2 -> LDS
//     once per chunk, not taps times;
//   * the filter tile of the chunk ([tap][32-channel block][plane][64 lanes x 16 B], already in B-operand
//     order in HBM, planes.hpp) is copied the same way and read back lane-linearly.
// One workgroup = WM x WN waves, each owning MT x NTL blocks of 32 pixels x 32 channels; workgroups are
// persistent over an XCD-contiguous range of spatial tiles (neighbouring tiles share halos and filters in one
// L2).  Two workgroups per CU alternate between their DMA wait and their MFMA phase; staging is single-buffered for
// P = 3 and for the two-chunk layers, double-buffered (the next item's DMA in flight under this item's MFMAs; fragment
// reads in inline asm, because hipcc drains every outstanding LDS-DMA before a compiler-visible LDS read) elsewhere.
// The epilogue transposes the accumulators through LDS into 16-byte stores of whole pixel rows, adds the bias and
// folds the BatchNorm statistics of the output (sum y, sum y^2 per channel; fp64 per workgroup record).
#include <algorithm>
#include <type_traits>
#include <vector>

#include "planes.hpp"

namespace rfi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct PConvDev {
    PConvArgs a;
    int nkc, ncb;                    // K chunks (all segments), 32-channel output blocks
    unsigned x_zero[2];              // byte offset of >= 64 zero bytes inside each segment's allocation
    unsigned wb_zero;                // ... and inside the filter allocation
    int diag;                        // RFI_PCONV_DIAG: 1 skip the MFMA phase, 2 skip the DMA (timing experiments only)
    unsigned long long* stamps;      // RFI_DIAG_STAMPS build: per-wave phase cycle sums
};

template <int R, int S, int TH, int TW, int WM, int WN, int MT, int NTL, int P, int G, int PAD = 1>
struct PCfg {
    static constexpr int NW = WM * WN, NT = NW * 64;
    static constexpr int BM = TH * TW, BN = WN * NTL * 32;
    static constexpr int HH = TH * S + R - S, HW = TW * S + R - S, HP = HH * HW;
    static constexpr int NTAP = R * R;
    static constexpr int RS = 2 * P + PAD;                  // 16-byte slots per halo pixel (2P data + PAD pad: with the pad
                                                            // the b128 reads of 64 consecutive pixels are conflict free,
                                                            // without it they are 2-way but 1/(2P+1) less is staged)
    static constexpr int ROWB = RS * 16;
    static constexpr int A_SLOTS = HP * RS;
    static constexpr int A_ITEMS = (A_SLOTS + NT - 1) / NT;
    static constexpr int A_BYTES = A_ITEMS * NT * 16;
    static constexpr int NCBL = BN / 32;
    static constexpr int B_SLOTS = NTAP * NCBL * P * 64;
    static constexpr int B_ITEMS = (B_SLOTS + NT - 1) / NT;
    static constexpr int B_BYTES = B_ITEMS * NT * 16;
    // [A | B | stats]; the per-wave transpose scratch of the epilogue (16 pixel rows x 36 floats) aliases the staging
    // area (the epilogue runs between two barriers, when nobody stages or reads)
    static constexpr int EPI_WAVE_BYTES = 16 * 36 * 4;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int STAT_OFF = STAGE_BYTES > NW * EPI_WAVE_BYTES ? STAGE_BYTES : NW * EPI_WAVE_BYTES;
    static constexpr int STAT_DOUBLES = NW * NTL * 32 * 2;  // [wave][n-tile][channel][sum, sumsq]
    static constexpr int LDS_BYTES = STAT_OFF + STAT_DOUBLES * 8;
    // double-buffered staging (DB kernels): [A0 | A1 | B0 | B1 | stats]
    static constexpr int STAT_OFF_DB = 2 * STAGE_BYTES > NW * EPI_WAVE_BYTES ? 2 * STAGE_BYTES : NW * EPI_WAVE_BYTES;
    static constexpr int LDS_BYTES_DB = STAT_OFF_DB + STAT_DOUBLES * 8;
    static_assert(BM == WM * MT * 32, "tile pixels must equal the waves' m-tiles");
    static_assert(32 % TW == 0 || TW % 32 == 0, "TW must divide or be a multiple of 32");
    static_assert(HW < 4096 && HH < 4096, "halo coordinates are packed in 12 bits");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
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

// Two workgroups per CU alternate between their DMA wait and their MFMA phase (staging is single-buffered; the
// other workgroup's MFMAs, LDS reads and output stores fill the wait).  A workgroup walks its tiles in GROUPS of
// G: for each K chunk the filter tile is staged ONCE and used by the G tiles of the group (G sets of
// accumulators), so the filter traffic per MFMA drops G-fold.
// OM: what the output tensor holds -- 0 float32, 1 float32 holding bf16-rounded values (round_y), 2 bfloat16 (y16).
// BWD / OM are template parameters: the epilogue is instruction-issue bound on the shallow layers (a runtime switch for
// BWD alone cost every launch 3 %).
// DB: two staging buffers per operand -- the DMA of the next (tile, chunk) item is in flight while this one is
// multiplied.  Fragment reads are inline asm there: hipcc puts s_waitcnt vmcnt(0) in front of every compiler-visible LDS
// read while an LDS-DMA is outstanding, which would drain the prefetch.
template <int R, int S, int TH, int TW, int WM, int WN, int MT, int NTL, int P, int G, int PAD, bool BWD, int OM, bool DB>
__global__ __launch_bounds__(WM * WN * 64, 2) void pconv_kernel(PConvDev d) {
    using C = PCfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD>;
    const PConvArgs& a = d.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent workgroup: blocks b and b+8 share an XCD (round-robin dispatch), so each XCD label gets
    // one contiguous range of tiles and its workgroups stride through it
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
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
    const int cb0 = blockIdx.y * C::NCBL, n0 = cb0 * 32;
    double* const st_out = a.stats ? a.stats + ((size_t)blockIdx.x * a.Cout + n0) * 2 : nullptr;
    if (my_tiles == 0) {                             // uniform for the workgroup, before any barrier
        if (st_out && tid < C::BN && n0 + tid < a.Cout) st_out[tid * 2] = st_out[tid * 2 + 1] = 0.0;
        return;
    }
    double* const s_stat = reinterpret_cast<double*>(smem + (DB ? C::STAT_OFF_DB : C::STAT_OFF));
    if (st_out)
        for (int i = tid; i < C::STAT_DOUBLES; i += C::NT) s_stat[i] = 0.0;   // visible after the first barrier below

    // ---- DMA descriptors.  Slot s = it * NT + tid of a staging image; a wave-instruction writes the 64
    // consecutive slots [it * NT + wave * 64, +64): LDS address = wave-uniform base + lane * 16.
    unsigned a_rc[C::A_ITEMS];                       // halo row << 16 | piece << 12 | halo column (tile independent)
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int s = it * C::NT + tid, pix = s / C::RS, piece = s % C::RS;
        a_rc[it] = (pix < C::HP && piece < 2 * P) ? ((unsigned)(pix / C::HW) << 16) | ((unsigned)piece << 12) | (unsigned)(pix % C::HW)
                                                  : 0x7fff0000u;
    }
    unsigned b_off[C::B_ITEMS];                      // byte offset of the slot's source in wB for chunk 0
    const unsigned wb_chunk = (unsigned)d.ncb * P * 1024;           // bytes per K chunk
#pragma unroll
    for (int it = 0; it < C::B_ITEMS; ++it) {
        int r = (it * C::NT + tid) >> 6;
        const int plane = r % P; r /= P;
        const int cbl = r % C::NCBL, tap = r / C::NCBL;
        b_off[it] = (tap < C::NTAP && cb0 + cbl < d.ncb)
                        ? (unsigned)(((tap * d.nkc) * d.ncb + cb0 + cbl) * P + plane) * 1024u + (unsigned)lane * 16u
                        : 0xffffffffu;
    }

    struct Tile { int n, oy0, ox0; };
    auto tile_of = [&](int k) {
        const int t = t_begin + j + k * gx;
        Tile r;
        r.ox0 = (t % tiles_x) * TW;
        r.oy0 = ((t / tiles_x) % tiles_y) * TH;
        r.n = t / (tiles_x * tiles_y);
        return r;
    };
    auto issue_A = [&](const Tile& t, int kc, int abuf = 0) {      // halo tile of chunk kc of one of this workgroup's tiles
        const int seg = kc >= a.x[0].nchunks ? 1 : 0;
        const unsigned coff = (unsigned)(seg ? kc - a.x[0].nchunks : kc) * (P * 32);
        const unsigned ps = (unsigned)a.x[seg].pstride * 2u;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x[seg].p);
        const unsigned xz = d.x_zero[seg];
        const int iy0 = t.oy0 * S - a.pad, ix0 = t.ox0 * S - a.pad, nb = t.n * a.Hin;
        unsigned char* dst = smem + abuf * C::A_BYTES;
#pragma unroll
        for (int it = 0; it < C::A_ITEMS; ++it) {
            const int iy = iy0 + (int)(a_rc[it] >> 16), ix = ix0 + (int)(a_rc[it] & 0xfff);
            const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;   // 0x7fff rows fail
            const unsigned off = ok ? (unsigned)((nb + iy) * a.Win + ix) * ps + ((a_rc[it] >> 12) & 0xf) * 16u + coff : xz;
            __builtin_amdgcn_global_load_lds((gbl_void*)(xb + off), (lds_void*)(dst + (it * C::NT + wave * 64) * 16), 16, 0, 0);
        }
    };
    auto issue_B = [&](int kc, int bbuf = 0) {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wB);
        const unsigned koff = (unsigned)kc * wb_chunk;
        unsigned char* dst = smem + (DB ? 2 * C::A_BYTES + bbuf * C::B_BYTES : C::A_BYTES);
#pragma unroll
        for (int it = 0; it < C::B_ITEMS; ++it) {
            const unsigned off = b_off[it] != 0xffffffffu ? b_off[it] + koff : d.wb_zero;
            __builtin_amdgcn_global_load_lds((gbl_void*)(wb + off), (lds_void*)(dst + (it * C::NT + wave * 64) * 16), 16, 0, 0);
        }
    };

    // ---- fragment addresses (bytes)
    int a_base[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wm * MT + mt) * 32 + li;
        const int ty = p / TW, tx = p % TW;
        a_base[mt] = ((ty * S) * C::HW + tx * S) * C::ROWB + lh * 16;
    }
    const int b_base = ((wn * NTL) * P * 64 + lane) * 16;

    f32x16 acc[G][MT][NTL];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][mt][nt][r] = 0.0f;

    const bool vec_out = (a.Cout & 3) == 0 && (a.y_pstride & 3) == 0 &&
                         (a.y16 ? (reinterpret_cast<uintptr_t>(a.y16) & 7) == 0 : (reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    // the bias of this lane's output channels, loaded ONCE: an ordinary global load inside the loop would make
    // the compiler drain the whole vector-memory queue (DMA prefetch and output stores included) at its use
    f32x4 bias4[NTL];
    float bias1[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int co0 = n0 + (wn * NTL + nt) * 32;
        bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        bias1[nt] = 0.0f;
        if (a.bias) {
            const int co = co0 + (lane & 7) * 4;
            const int cz = a.zblocks ? co % (32 * a.zblocks) : co;          // (tap groups share the bias)
            if (vec_out && co < a.Cout) bias4[nt] = *reinterpret_cast<const f32x4*>(a.bias + cz);
            if (!vec_out && co0 + li < a.Cout) bias1[nt] = a.bias[co0 + li];
        }
        // opaque to the compiler from here on: otherwise it re-loads the (invariant) bias at every use instead of
        // keeping 5 registers, and each of those loads drains the vector-memory queue
        asm volatile("" : "+v"(bias4[nt].x), "+v"(bias4[nt].y), "+v"(bias4[nt].z), "+v"(bias4[nt].w), "+v"(bias1[nt]));
    }
    // BatchNorm-backward sums in the epilogue (PConvArgs::bwd_y16): the layer's per-channel coefficients of this lane's
    // four channels, loaded once like the bias
    // (BWD is a template parameter: the extra registers and branches slowed EVERY launch by 3 % as a runtime switch)
    const bool bwd = BWD && a.bwd_y16 != nullptr && a.stats != nullptr && vec_out;
    f32x4 bsc[BWD ? NTL : 1], bsh[BWD ? NTL : 1], bmu[BWD ? NTL : 1], bis[BWD ? NTL : 1];
#pragma unroll
    for (int nt = 0; nt < (BWD ? NTL : 0); ++nt) {
        bsc[nt] = bsh[nt] = bmu[nt] = bis[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int co = n0 + (wn * NTL + nt) * 32 + (lane & 7) * 4;
        if (bwd && co < a.Cout) {
            bsc[nt] = *reinterpret_cast<const f32x4*>(a.bwd_scale + co);
            bsh[nt] = *reinterpret_cast<const f32x4*>(a.bwd_shift + co);
            bmu[nt] = *reinterpret_cast<const f32x4*>(a.bwd_mean + co);
            bis[nt] = *reinterpret_cast<const f32x4*>(a.bwd_invstd + co);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bsc[nt][e]), "+v"(bsh[nt][e]), "+v"(bmu[nt][e]), "+v"(bis[nt][e]));
    }
    // ---- epilogue of one tile.  C/D layout of 32x32: col = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2)
    // + 4 * (lane >> 5) (pixel).  Each 32x32 block is transposed through this wave's private LDS scratch and leaves
    // as 4 x 16-byte-per-lane stores of whole 128-byte pixel rows; BatchNorm statistics are folded on the way.
    // store offsets (elements): a wave-uniform part per (tile, 8-pixel row group) + this lane's part -- pixel (lane >> 3)
    // of the group, channels 4 (lane & 7) .. + 3 of the n-tile
    static_assert(TW % 8 == 0, "an 8-pixel row group must not wrap");
    const unsigned col_st = (unsigned)(a.osx * a.y_pstride), row_st = (unsigned)(a.osy * a.Wout * a.y_pstride);
    unsigned lane_off[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int co = n0 + (wn * NTL + nt) * 32 + (lane & 7) * 4;
        lane_off[nt] = (unsigned)(lane >> 3) * col_st + (unsigned)co;
        if (a.zblocks) {                             // tap group z = 2 a + b of the transposed conv: pixel offset (a, b), channels of the group
            const int z = co / (32 * a.zblocks), cz = co % (32 * a.zblocks);
            lane_off[nt] = (unsigned)(lane >> 3) * col_st + (unsigned)((z >> 1) * a.Wout * a.y_pstride + (z & 1) * a.y_pstride + cz);
        }
    }
    auto epilogue = [&](const Tile& ct, f32x16 (&ac)[MT][NTL], float* s_ep, auto vec_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;   // 16-byte row stores (vec_out) or the scalar fallback: one branch per tile
        const int xs = a.osx * a.y_pstride;
        // tiles that lie inside the image with all their channels (every tile of the U-Net shapes): no bounds tests
        const bool full = ct.oy0 + TH <= a.H && ct.ox0 + TW <= a.W && n0 + C::BN <= a.Cout;
        const unsigned tbase = (unsigned)(((ct.n * a.Hout + ct.oy0 * a.osy + a.ooy) * a.Wout + ct.ox0 * a.osx + a.oox) * a.y_pstride);
        // bwd: the layer's raw outputs at this lane's (pixel, 4 channels) positions, fetched before any store of the
        // tile is issued (the wait at their first use then never has to pass one of this tile's stores)
        u32x2 ypre[BWD ? MT : 1][BWD ? NTL : 1][2][2];
        if constexpr (BWD) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int half = 0; half < 2; ++half)
#pragma unroll
                        for (int ps = 0; ps < 2; ++ps) {
                            const int pp = half * 16 + ps * 8 + (lane >> 3);
                            const int oy = ct.oy0 + ((wm * MT + mt) * 32) / TW + pp / TW, ox = ct.ox0 + ((wm * MT + mt) * 32) % TW + pp % TW;
                            const int co = n0 + (wn * NTL + nt) * 32 + (lane & 7) * 4;
                            ypre[mt][nt][half][ps] = u32x2{0u, 0u};
                            if (co < a.Cout && oy < a.H && ox < a.W)
                                ypre[mt][nt][half][ps] = *reinterpret_cast<const u32x2*>(
                                    a.bwd_y16 + (size_t)((ct.n * a.H + oy) * a.W + ox) * a.bwd_yps + co);
                        }
        }
        if constexpr (VEC) {
            // Quarter blocks (4 accumulator registers = 8 pixel rows x 32 channels) alternate between the two halves of the
            // scratch: the 16-byte read of quarter qi - 1 is issued in front of quarter qi's LDS writes and its arithmetic and
            // store behind them, so no LDS round trip is waited for.  Tiles that lie inside the image with all their channels
            // (every tile of the model shapes) run a copy of the loop WITHOUT bounds tests, and the statistics are accumulated
            // whether or not the launch wants them: the loop is straight-line code -- with a branch per quarter (full tile?
            // statistics? backward sums?) the compiler waited for every LDS operation at each join and copied the statistics
            // registers at every merge; skipping the epilogue altogether (RFI_PCONV_DIAG=4) showed it to be 24 % (512-channel
            // layers) to 42 % (64-channel layers at 1024 x 1024) of the kernel
            constexpr int NQ = NTL * MT * 4;
            const int pl = lane >> 3, g4 = (lane & 7) * 4;
            f32x4 p1[NTL], p2[NTL];
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) p1[nt] = p2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // quarter pq's 8 pixel rows x 4 channels of this lane (rq: its accumulator values, transposed): bias, rounding, the
            // 8- / 16-byte row store, statistics (or the BatchNorm-backward sums)
            auto finish_quarter = [&](auto full_tag, int pq, f32x4 rq) {
                constexpr bool FULL = decltype(full_tag)::value;
                const int nt = pq / (MT * 4), mt = (pq / 4) % MT, q = pq & 3;
                f32x4 v = rq + bias4[nt];
                const int p0 = (wm * MT + mt) * 32 + q * 8;              // first pixel of the row group (wave-uniform)
                const unsigned off = tbase + (unsigned)(p0 / TW) * row_st + (unsigned)(p0 % TW) * col_st + lane_off[nt];
                bool ok = true;
                if constexpr (!FULL) {
                    const int co = n0 + (wn * NTL + nt) * 32 + g4;
                    const int oy = ct.oy0 + (p0 + pl) / TW, ox = ct.ox0 + (p0 + pl) % TW;
                    ok = co < a.Cout && oy < a.H && ox < a.W;
                }
                if (FULL || ok) {
                    if constexpr (OM >= 1) {            // the tensor holds bf16 values (RNE; NaN stays NaN): v_cvt_pk_bf16_f32
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                        const unsigned w0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
                        const unsigned w1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, bf16x2));
                        v = f32x4{__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xffff0000u),
                                  __builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xffff0000u)};
#ifdef RFI_DIAG_STAMPS
                        if (d.diag & 8) asm volatile("" :: "v"(w0), "v"(w1));      // (timing experiment: the epilogue without its stores)
                        else
#endif
                        if constexpr (OM == 2) *reinterpret_cast<u32x2*>(a.y16 + off) = u32x2{w0, w1};
                    }
                    if constexpr (OM != 2) *reinterpret_cast<f32x4*>(a.y + off) = v;
                    if constexpr (BWD) {
                        const u32x2 t = ypre[BWD ? mt : 0][BWD ? nt : 0][q >> 1][q & 1];
                        const f32x4 yv = {__builtin_bit_cast(float, t[0] << 16), __builtin_bit_cast(float, t[0] & 0xffff0000u),
                                          __builtin_bit_cast(float, t[1] << 16), __builtin_bit_cast(float, t[1] & 0xffff0000u)};
                        const f32x4 z = yv * bsc[BWD ? nt : 0] + bsh[BWD ? nt : 0], xh = (yv - bmu[BWD ? nt : 0]) * bis[BWD ? nt : 0];
                        f32x4 dz;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.0f ? v[e] : v[e] * a.bwd_slope;
                        p1[nt] += dz;
                        p2[nt] += dz * xh;
                    } else {
                        p1[nt] += v;
                        p2[nt] += v * v;
                    }
                }
            };
            // (Measured and not kept, round 4: the same loop for full tiles as a hand-counted software pipeline two quarters deep
            // -- inline-asm ds_read_b128 / ds_write_b32 with lgkmcnt(5) instead of the compiler's lgkmcnt(0) per quarter.  Cycle
            // stamps: epilogue 8.1 k -> 7.0 k cycles per 64-channel tile, the co-resident workgroup's MFMA phase 1.64 k -> 1.68 k
            // per item; kernel and step time +-0.)
            auto quarters = [&](auto full_tag) {
                f32x4 rq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int qi = 0; qi <= NQ; ++qi) {
                    if (qi > 0) rq = *reinterpret_cast<const f32x4*>(s_ep + ((qi - 1) & 1) * 288 + pl * 36 + g4);
                    if (qi < NQ) {
                        const int nt = qi / (MT * 4), mt = (qi / 4) % MT, q = qi & 3;
#pragma unroll
                        for (int k = 0; k < 4; ++k) s_ep[(qi & 1) * 288 + (k + 4 * lh) * 36 + li] = ac[mt][nt][q * 4 + k];
                    }
                    if (qi > 0) finish_quarter(full_tag, qi - 1, rq);
                }
            };
            if (full) quarters(std::true_type{});
            else quarters(std::false_type{});
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ac[mt][nt][r] = 0.0f;
            if (st_out) {
                // lanes l, l^8, l^16, l^32 hold the same 4 channels of different pixels: fold them
                // (<= 128 fp32 terms per channel), then continue in fp64 in this wave's LDS slots
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int o = 8; o < 64; o <<= 1) {
                            p1[nt][e] += __shfl_xor(p1[nt][e], o, 64);
                            p2[nt][e] += __shfl_xor(p2[nt][e], o, 64);
                        }
                    }
                    if (lane < 8) {
                        double* dd = s_stat + ((wave * NTL + nt) * 32 + lane * 4) * 2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dd[e * 2] += (double)p1[nt][e];
                            dd[e * 2 + 1] += (double)p2[nt][e];
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int co = n0 + (wn * NTL + nt) * 32 + li;
                const bool cok = co < a.Cout;
                const float bv = bias1[nt];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int oyb = ct.oy0 + ((wm * MT + mt) * 32) / TW;
                    const int oxb = ct.ox0 + ((wm * MT + mt) * 32) % TW;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pp = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int oy = oyb + pp / TW, ox = oxb + pp % TW;
                        if (cok && oy < a.H && ox < a.W)
                            a.y[(unsigned)(((ct.n * a.Hout + oy * a.osy + a.ooy) * a.Wout + a.oox) * a.y_pstride + co) +
                                (unsigned)(ox * xs)] = ac[mt][nt][r] + bv;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) ac[mt][nt][r] = 0.0f;
                }
            }
        }
    };
    auto epilogue_group = [&](int g0, const Tile (&tl)[G]) {
        float* s_ep = reinterpret_cast<float*>(smem + wave * C::EPI_WAVE_BYTES);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g0 + g < my_tiles && !(d.diag & 4)) {
                if (vec_out) epilogue(tl[g], acc[g], s_ep, std::true_type{});
                else if constexpr (OM == 0) epilogue(tl[g], acc[g], s_ep, std::false_type{});
            }
    };

    const unsigned char* const sA = smem;
    const unsigned char* const sB = smem + C::A_BYTES;
#ifdef RFI_DIAG_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long nitem_ = 0;
#define RFI_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define RFI_ACC(i, a_, b_) st_[i] += (b_) - (a_)
#else
#define RFI_T(v)
#define RFI_ACC(i, a_, b_)
#endif
    if constexpr (DB) {
        static_assert(P == 1, "double-buffered staging: bf16 flow");
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NRD = MT + NTL;                       // fragment reads per tap
        const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(smem);      // (low half of a flat LDS address = LDS offset)
        unsigned a_addr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_addr[mt] = lds0 + (unsigned)a_base[mt];
        const unsigned b_addr = lds0 + 2 * C::A_BYTES + (unsigned)b_base;
        for (int g0 = 0; g0 < my_tiles; g0 += G) {
            const int gcount = my_tiles - g0 < G ? my_tiles - g0 : G;
            Tile tl[G];
#pragma unroll
            for (int g = 0; g < G; ++g) tl[g] = tile_of(g0 + (g < gcount ? g : 0));
            issue_A(tl[0], 0, 0);                           // item 0 of the group
            issue_B(0, 0);
            int item = 0;
            for (int kc = 0; kc < d.nkc; ++kc) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (g < gcount) {
                        const bool last = kc == d.nkc - 1 && g == gcount - 1;
                        const bool wrap = g + 1 >= gcount;                  // the next item starts a new chunk (needs its filters)
                        RFI_T(t0);
                        if (!last) {
                            issue_A(wrap ? tl[0] : tl[G > 1 ? g + 1 < G ? g + 1 : 0 : 0], wrap ? kc + 1 : kc, (item + 1) & 1);
#ifdef RFI_DIAG_STAMPS
                            if (wrap && !(d.diag & 2)) issue_B(kc + 1, (kc + 1) & 1);      // (timing experiment: no filter DMA after the first chunk)
#else
                            if (wrap) issue_B(kc + 1, (kc + 1) & 1);
#endif
                        }
                        RFI_T(t1);
                        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RFI_DIAG_STAMPS
                        else if (wrap && !(d.diag & 2)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::A_ITEMS + C::B_ITEMS) : "memory");
#else
                        else if (wrap) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::A_ITEMS + C::B_ITEMS) : "memory");
#endif
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::A_ITEMS) : "memory");
                        RFI_T(t2);
                        __builtin_amdgcn_s_barrier();                       // every wave's pieces of THIS item have landed
                        RFI_T(t3);
                        {
                            const unsigned aoff = (unsigned)(item & 1) * C::A_BYTES, boff = (unsigned)(kc & 1) * C::B_BYTES;
                            u32x4 afr[2][MT], bfr[2][NTL];
                            auto load_frags = [&](auto tap_c, u32x4 (&af)[MT], u32x4 (&bf)[NTL]) {
                                constexpr int tap = decltype(tap_c)::value;
                                constexpr int tr = tap / R, ts = tap % R;
#pragma unroll
                                for (int nt = 0; nt < NTL; ++nt) {
                                    const unsigned ad = b_addr + boff + (unsigned)(nt * 1024);       // (a named local: asm operands alone do not capture)
                                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[nt]) : "v"(ad), "n"(tap * C::NCBL * 1024));
                                }
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) {
                                    const unsigned ad = a_addr[mt] + aoff;
                                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[mt]) : "v"(ad), "n"((tr * C::HW + ts) * C::ROWB));
                                }
                            };
                            auto wait_frags = [&](auto n_c, u32x4 (&af)[MT], u32x4 (&bf)[NTL]) {     // frags usable once <= n newer reads are pending
                                constexpr int n = decltype(n_c)::value;
                                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory");
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(af[mt]));
#pragma unroll
                                for (int nt = 0; nt < NTL; ++nt) asm volatile("" : "+v"(bf[nt]));
                            };
                            auto tap_step = [&](auto tap_c) {
                                constexpr int tap = decltype(tap_c)::value;
                                if constexpr (tap + 1 < C::NTAP) {
                                    load_frags(std::integral_constant<int, tap + 1>{}, afr[(tap + 1) & 1], bfr[(tap + 1) & 1]);
                                    wait_frags(std::integral_constant<int, NRD>{}, afr[tap & 1], bfr[tap & 1]);
                                } else {
                                    wait_frags(std::integral_constant<int, 0>{}, afr[tap & 1], bfr[tap & 1]);
                                }
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                    for (int nt = 0; nt < NTL; ++nt)
                                        acc[g][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                            __builtin_bit_cast(bf16x8, afr[tap & 1][mt]), __builtin_bit_cast(bf16x8, bfr[tap & 1][nt]), acc[g][mt][nt], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            };
                            load_frags(std::integral_constant<int, 0>{}, afr[0], bfr[0]);
                            static_assert(C::NTAP == 9, "3 x 3 taps");
                            tap_step(std::integral_constant<int, 0>{}); tap_step(std::integral_constant<int, 1>{});
                            tap_step(std::integral_constant<int, 2>{}); tap_step(std::integral_constant<int, 3>{});
                            tap_step(std::integral_constant<int, 4>{}); tap_step(std::integral_constant<int, 5>{});
                            tap_step(std::integral_constant<int, 6>{}); tap_step(std::integral_constant<int, 7>{});
                            tap_step(std::integral_constant<int, 8>{});
                        }
                        RFI_T(t4);
                        __builtin_amdgcn_s_barrier();       // every wave is done reading this item's buffers (all reads were waited for)
                        RFI_T(t5);
#ifdef RFI_DIAG_STAMPS
                        RFI_ACC(0, t0, t1); RFI_ACC(1, t1, t2); RFI_ACC(2, t2, t3); RFI_ACC(3, t3, t4); RFI_ACC(4, t4, t5);
                        ++nitem_;
#endif
                        ++item;
                    }
                }
            }
            RFI_T(te0);
            epilogue_group(g0, tl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#ifdef RFI_DIAG_STAMPS
            { RFI_T(te1); RFI_ACC(5, te0, te1); }
#endif
        }
    } else
    for (int g0 = 0; g0 < my_tiles; g0 += G) {
        const int gcount = my_tiles - g0 < G ? my_tiles - g0 : G;
        Tile tl[G];                                  // (the three scalar divisions of tile_of: once per tile, not per chunk)
#pragma unroll
        for (int g = 0; g < G; ++g) tl[g] = tile_of(g0 + (g < gcount ? g : 0));
        for (int kc = 0; kc < d.nkc; ++kc) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < gcount) {
                    RFI_T(t0);
                    if (!(d.diag & 2) || (g0 == 0 && kc == 0 && g == 0)) {
                        issue_A(tl[g], kc);
                        if (g == 0) issue_B(kc);     // the filter tile of the chunk serves every tile of the group
                    }
                    RFI_T(t1);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    RFI_T(t2);
                    __syncthreads();                 // s_waitcnt vmcnt(0) + barrier: every wave's pieces have landed
                    RFI_T(t3);
                    if (!(d.diag & 1)) {
                        // software pipeline over the taps: the fragments of tap t+1 are read from LDS BEFORE the MFMAs
                        // of tap t are issued, so their LDS latency hides under those MFMAs (hipcc otherwise sinks every
                        // ds_read to just before its first use and the matrix pipe idles for one LDS round trip per
                        // fragment: measured 66 % duty in this phase).  The fences pin "reads of t+1, then MFMAs of t".
                        bf16x8 afr[2][MT][P], bfr[2][NTL][P];
                        auto load_frags = [&](int tap, bf16x8 (&af)[MT][P], bf16x8 (&bf)[NTL][P]) {
                            const int tr = tap / R, ts = tap % R;
#pragma unroll
                            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                                for (int p = 0; p < P; ++p)
                                    bf[nt][p] = *reinterpret_cast<const bf16x8*>(sB + b_base + ((tap * C::NCBL + nt) * P + p) * 1024);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int p = 0; p < P; ++p)
                                    af[mt][p] = *reinterpret_cast<const bf16x8*>(sA + a_base[mt] + (tr * C::HW + ts) * C::ROWB + p * 32);
                        };
                        load_frags(0, afr[0], bfr[0]);
#pragma unroll
                        for (int tap = 0; tap < C::NTAP; ++tap) {
                            if (tap + 1 < C::NTAP) load_frags(tap + 1, afr[(tap + 1) & 1], bfr[(tap + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NTL; ++nt)
                                    acc[g][mt][nt] = mma<P>(afr[tap & 1][mt], bfr[tap & 1][nt], acc[g][mt][nt]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    RFI_T(t4);
                    __syncthreads();                 // every wave is done reading the halo tile (and, after the last tile
                    RFI_T(t5);                       // of the group, the filter tile)
#ifdef RFI_DIAG_STAMPS
                    RFI_ACC(0, t0, t1); RFI_ACC(1, t1, t2); RFI_ACC(2, t2, t3); RFI_ACC(3, t3, t4); RFI_ACC(4, t4, t5);
                    ++nitem_;
#endif
                }
            }
        }
        RFI_T(te0);
        epilogue_group(g0, tl);
        // the staging area (aliased by the scratch) is about to be overwritten: every wave must be done with its
        // scratch READS (lgkmcnt), but nobody has to wait for the output stores to retire (a __syncthreads() would
        // add s_waitcnt vmcnt(0): one exposed HBM write latency per group)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef RFI_DIAG_STAMPS
        { RFI_T(te1); RFI_ACC(5, te0, te1); }
#endif
    }
#ifdef RFI_DIAG_STAMPS
    if (d.stamps && lane == 0) {
        unsigned long long* o = d.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * C::NW + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = st_[i];
        o[6] = nitem_;
        o[7] = (unsigned long long)my_tiles;
    }
#endif
    if (st_out) {
        __syncthreads();
        if (tid < C::BN && n0 + tid < a.Cout) {      // channel tid of the tile: waves (m, wn_c) cover it
            const int wn_c = tid / (NTL * 32), nt_c = (tid % (NTL * 32)) / 32, cl = tid % 32;
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const double* dd = s_stat + (((m * WN + wn_c) * NTL + nt_c) * 32 + cl) * 2;
                t1 += dd[0];
                t2 += dd[1];
            }
            st_out[tid * 2] = t1;
            st_out[tid * 2 + 1] = t2;
        }
    }
}

template <int R, int S, int TH, int TW, int WM, int WN, int MT, int NTL, int P, int G, int PAD = 1, bool BWD = false, int OM = -1, bool DB = false>
void launch_cfg(rfi_ctx* ctx, PConvDev& d) {
    if constexpr (OM >= 0 && !DB && P == 1 && R == 3 && S == 1) {      // double-buffered staging from 3 K chunks (bf16 flow; RFI_PCONV_DB=0: off)
        static const int db = getenv("RFI_PCONV_DB") ? atoi(getenv("RFI_PCONV_DB")) : 3;
        if (db > 0 && d.nkc >= db) return launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, BWD, OM, true>(ctx, d);
    }
    if constexpr (OM < 0) {                           // pick the epilogue variant of this launch
        const int om = d.a.y16 ? 2 : d.a.round_y ? 1 : 0;
        const bool b = d.a.bwd_y16 != nullptr;
        if constexpr (R != 3 || S != 1) {            // the other shapes of the bfloat16 flow: bfloat16 in, bfloat16 out
            RFI_REQUIRE(P == 1 && om == 2, "pconv: strided / 1x1 / 2x2 contractions exist for the bfloat16 flow only (bfloat16 output)");
            if constexpr (R == 2 && S == 2) {        // (a transposed conv's input gradient feeds a BatchNorm layer: its backward sums)
                if (b) return launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, true, 2>(ctx, d);
            } else {
                RFI_REQUIRE(!b, "pconv: the BatchNorm-backward epilogue exists for 3x3 stride-1 and 2x2 stride-2 contractions");
            }
            return launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, false, 2>(ctx, d);
        } else if constexpr (P == 1) {
            RFI_REQUIRE(!(b && om == 1), "pconv: the BatchNorm-backward epilogue writes float32 or bfloat16 tensors");
            if (om == 2) return b ? launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, true, 2>(ctx, d)
                                  : launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, false, 2>(ctx, d);
            if (om == 1) return launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, false, 1>(ctx, d);
            return b ? launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, true, 0>(ctx, d)
                     : launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, false, 0>(ctx, d);
        } else {
            RFI_REQUIRE(om == 0 && !b, "pconv: bfloat16 outputs and the BatchNorm-backward epilogue exist for the bfloat16 flow only");
            return launch_cfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, false, 0>(ctx, d);
        }
    } else {
    using C = PCfg<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD>;
    PConvArgs& a = d.a;
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int ychunks = (int)cdiv(d.ncb, C::NCBL);
    const size_t lds = DB ? C::LDS_BYTES_DB : C::LDS_BYTES;
    // persistent grid: about (256 CUs x resident workgroups) workgroups in total, a multiple of 8 along x, tiles
    // spread evenly over the workgroups of each XCD label
    int occ = (int)((160 * 1024) / lds);
    occ = occ < 1 ? 1 : (occ > 2 ? 2 : occ);
    const int gmax = std::max(8, (256 * occ) / ychunks);
    const int tx = (int)cdiv(ntiles, 8);
    const int per = (int)cdiv(tx, std::max(1, gmax / 8));
    int GX = 8 * (int)cdiv(tx, per);
    if (ntiles < 8) GX = ntiles;
    const bool vec_out = (a.Cout & 3) == 0 && (a.y_pstride & 3) == 0 &&
                         (a.y16 ? (reinterpret_cast<uintptr_t>(a.y16) & 7) == 0 : (reinterpret_cast<uintptr_t>(a.y) & 15) == 0);
    RFI_REQUIRE(vec_out || !(a.y16 || a.round_y), "pconv: bf16 output needs Cout % 4 == 0 and an aligned tensor");
    if (a.stats && vec_out && GX <= a.stats_max_records) a.stats_records = GX;
    else { a.stats = nullptr; a.stats_records = 0; }
    RFI_REQUIRE(!a.bwd_y16 || P == 1, "pconv: the BatchNorm-backward epilogue exists for the bfloat16 flow only");
    RFI_REQUIRE(!a.bwd_y16 || (a.Hout == a.H && a.Wout == a.W && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0 &&
                               (a.bwd_yps & 3) == 0 && (reinterpret_cast<uintptr_t>(a.bwd_y16) & 7) == 0),
                "pconv: BatchNorm-backward sums need a dense output grid and an aligned bfloat16 Y");
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pconv_kernel<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, BWD, OM, DB>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
#ifdef RFI_DIAG_STAMPS
    {
        const size_t nw = (size_t)GX * ychunks * C::NW;
        RFI_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d.stamps), nw * 64));
        RFI_CHECK_HIP(hipMemsetAsync(d.stamps, 0, nw * 64, ctx->stream));
        hipLaunchKernelGGL((pconv_kernel<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, BWD, OM, DB>), dim3(GX, ychunks), dim3(C::NT), lds, ctx->stream, d);
        std::vector<unsigned long long> hs(nw * 8);
        RFI_CHECK_HIP(hipMemcpyAsync(hs.data(), d.stamps, nw * 64, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < nw; ++w) for (int i = 0; i < 8; ++i) sm[i] += (double)hs[w * 8 + i];
        std::fprintf(stderr, "[stamps] pconv %dx%dx%d k%d->%d r%ds%d%s%s P%d tile %dx%d grid %dx%d items/wave %.1f tiles/wg %.1f | cycles per item: issue %.0f "
                     "vmwait %.0f bar1 %.0f mfma %.0f bar2 %.0f | epilogue per group %.0f\n", a.N, a.H, a.W, d.nkc * 16, a.Cout, R, S, DB ? " db" : "",
                     BWD ? " bwd" : "", P, TH, TW, GX, ychunks, sm[6] / nw,
                     sm[7] / nw, sm[0] / sm[6], sm[1] / sm[6], sm[2] / sm[6], sm[3] / sm[6], sm[4] / sm[6], sm[5] / (sm[7] / G));
        RFI_CHECK_HIP(hipFree(d.stamps));
        d.stamps = nullptr;
        return;
    }
#endif
    hipLaunchKernelGGL((pconv_kernel<R, S, TH, TW, WM, WN, MT, NTL, P, G, PAD, BWD, OM, DB>), dim3(GX, ychunks), dim3(C::NT), lds, ctx->stream, d);
    check_launch("pconv");
    }
}

// Tile choice by output width.  Measured and NOT kept (round 2, gpurun_out/r2k-r2m): groups of 4 tiles, 64-channel
// workgroup tiles, halo rows without the pad slot (-20..-38 % staged bytes per MFMA: +-3 %); a start delay for the
// second workgroup of each CU (no effect); one 8-wave workgroup per CU with every buffer doubled and the prefetch
// DMA issued between the MFMAs of the previous item (same time for float32 pieces, 15-30 % slower for bf16).  The
// Also not kept: a transposed epilogue (operands swapped so that a lane holds 4 x 4 consecutive channels of ONE pixel
// and stores them straight from the accumulators, no transpose through LDS): correct, but each wave store then
// covers 16 bytes of 32 different pixel rows instead of 8 whole rows, and the step went from 4.48 to 5.01 ms.  The
// kernel sits at 1.0-1.2 PFLOP/s of executed bf16 MFMA on the deep layers (the float32-by-3xbf16 arithmetic: x 1/6),
// which is where the best known hand-scheduled GEMM of the CDNA4 guide ends on random data (1.32-1.34 PFLOP/s at a
// sustained 1.9 GHz); the 32-channel layers are bound by HBM (6 B per activation element in, 4 B out).
// 3x3 stride 2, 1x1 stride 2 (on the full-resolution input: the halo tile simply has the stride) and the 2x2 stride-1
// contractions of a stride-2 layer's input gradient (one per output parity class)
template <int R, int S>
void dispatch_other(rfi_ctx* ctx, PConvDev& d) {
    const PConvArgs& a = d.a;
    if (a.Cout >= 64 && a.W >= 32) return launch_cfg<R, S, 8, 32, 4, 1, 2, 2, 1, 1, 1>(ctx, d);
    if (a.W >= 32) launch_cfg<R, S, 8, 32, 4, 1, 2, 1, 1, 2, 1>(ctx, d);
    else if (a.W >= 16) launch_cfg<R, S, 16, 16, 4, 1, 2, 1, 1, 2, 1>(ctx, d);
    else launch_cfg<R, S, 8, 8, 2, 2, 1, 1, 1, 2, 1>(ctx, d);
}

template <int P>
void dispatch(rfi_ctx* ctx, PConvDev& d) {
    const PConvArgs& a = d.a;
    // bf16 flow, 64 output channels and more: 2 x 2 blocks per wave on ONE tile per group (a third fewer LDS bytes per MFMA
    // than 2 x 1 blocks on two tiles; pays only together with the double-buffered staging: -0.9 % step time;
    // RFI_PCONV_NT2=0 for A/B runs)
    static const int wide = getenv("RFI_PCONV_NT2") ? atoi(getenv("RFI_PCONV_NT2")) : 1;
    if constexpr (P == 1) {
        if (wide && a.Cout >= 64 * wide) {
            if (a.W >= 32) return launch_cfg<3, 1, 8, 32, 4, 1, 2, 2, P, 1, 1>(ctx, d);
            if (a.W >= 16) return launch_cfg<3, 1, 16, 16, 4, 1, 2, 2, P, 1, 1>(ctx, d);
        }
    }
    if (a.W >= 32) launch_cfg<3, 1, 8, 32, 4, 1, 2, 1, P, 2, 1>(ctx, d);
    else if (a.W >= 16) launch_cfg<3, 1, 16, 16, 4, 1, 2, 1, P, 2, 1>(ctx, d);
    else launch_cfg<3, 1, 8, 8, 2, 2, 1, 1, P, 2, 1>(ctx, d);
}

}  // namespace

void launch_pconv(rfi_ctx* ctx, PConvArgs& a) {
    RFI_REQUIRE(a.P == 1 || a.P == 3, "pconv: planes must be 1 or 3");
    const bool plain = a.R == 3 && a.S == 1 && a.pad == 1;
    const bool s2 = a.R == 3 && a.S == 2 && a.pad == 1, p2 = a.R == 1 && a.S == 2 && a.pad == 0, c2 = a.R == 2 && a.S == 1 && a.pad == 0;
    const bool t1 = a.R == 1 && a.S == 1 && a.pad == 0, t2 = a.R == 2 && a.S == 2 && a.pad == 0;      // transposed conv: forward / input gradient
    RFI_REQUIRE(plain || ((s2 || p2 || c2 || t1 || t2) && a.P == 1),
                "pconv: 3x3 stride 1 pad 1; bfloat16 flow also 3x3 stride 2 pad 1, 1x1 stride 1 / 2, 2x2 stride 1 / 2 pad 0");
    RFI_REQUIRE(!a.zblocks || (t1 && a.Cout == 128 * a.zblocks && a.osy == 2 && a.osx == 2 && a.y16),
                "pconv: tap groups are the four taps of a ConvTranspose2d(k2, s2) with Cout % 32 == 0, bfloat16 output");
    RFI_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cout > 0 && a.nseg >= 1 && a.nseg <= 2, "pconv: empty shape");
    PConvDev d;
    d.a = a;
    d.nkc = a.x[0].nchunks + (a.nseg > 1 ? a.x[1].nchunks : 0);
    d.ncb = (a.Cout + 31) / 32;
    if (a.nseg == 1) d.a.x[1] = PlaneSeg{a.x[0].p, a.x[0].pstride, 0};
    const int64_t pix = (int64_t)a.N * a.Hin * a.Win;
    for (int s = 0; s < 2; ++s) {
        const int64_t bytes = pix * d.a.x[s].pstride * 2;
        RFI_REQUIRE(bytes + 64 < ((int64_t)1 << 32), "pconv: input tensor too large for 32-bit byte offsets");
        d.x_zero[s] = (unsigned)bytes;                 // every plane tensor is allocated with a zeroed 64-byte tail
    }
    const int64_t wbytes = (int64_t)wb_elems(a.R * a.R, a.Cout, 16 * a.x[0].nchunks, a.nseg > 1 ? 16 * a.x[1].nchunks : 0, a.P) * 2;
    RFI_REQUIRE(wbytes + 64 < ((int64_t)1 << 32), "pconv: filter tensor too large");
    d.wb_zero = (unsigned)wbytes;                      // ... and so is every wB tensor
    static const int diag = getenv("RFI_PCONV_DIAG") ? atoi(getenv("RFI_PCONV_DIAG")) : 0;
    d.diag = diag;
    d.stamps = nullptr;
    RFI_REQUIRE((int64_t)a.N * a.Hout * a.Wout * a.y_pstride < (int64_t)1 << 31, "pconv: output too large for 32-bit offsets");
    const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * (double)a.Cout * (a.R * a.R) * 16.0 * d.nkc;
    std::string label;
    if (ctx->profiling)
        label = "pconv N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" + std::to_string(a.W) + " k" +
                std::to_string(d.nkc * 16) + "->" + std::to_string(a.Cout) + (a.P == 3 ? " 3xbf16" : " bf16") +
                (plain ? "" : " r" + std::to_string(a.R) + "s" + std::to_string(a.S));
    // algorithmic HBM bytes: the plane input (2 P bytes per value), the float32 output, the filters
    const double bytes = (double)a.N * a.Hin * a.Win * d.nkc * 16.0 * 2 * a.P + (a.y16 ? 2.0 : 4.0) * a.N * a.H * a.W * a.Cout +
                         (a.bwd_y16 ? 2.0 * a.N * a.H * a.W * a.Cout : 0.0) + (double)wbytes;     // (+ the bf16 Y of the BatchNorm-backward epilogue)
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, bytes, label);
    if (s2) dispatch_other<3, 2>(ctx, d);
    else if (p2) dispatch_other<1, 2>(ctx, d);
    else if (c2) dispatch_other<2, 1>(ctx, d);
    else if (t1) dispatch_other<1, 1>(ctx, d);
    else if (t2) dispatch_other<2, 2>(ctx, d);
    else if (a.P == 3) dispatch<3>(ctx, d);
    else dispatch<1>(ctx, d);
    a.stats = d.a.stats;
    a.stats_records = d.a.stats_records;
}

// Bridge for callers that hold float32 NHWC tensors (the kernel-level C ABI): split the input (with its load
// transform) and the filters into temporary plane tensors, run the plane kernel, free the temporaries.
void launch_pconv_from_f32(rfi_ctx* ctx, ConvArgs& c, int P) {
    RFI_REQUIRE(c.R == 3 && c.S == 1 && c.pad == 1 && c.zgroups == 1, "pconv bridge: 3x3 stride-1 only");
    const int64_t pix = (int64_t)c.N * c.Hin * c.Win;
    const size_t xe = plane_elems(pix, c.Cin, P), we = wb_elems(9, c.Cout, c.Cin, 0, P);
    bf16_t* xp = static_cast<bf16_t*>(ctx->alloc(xe * 2 + 64));
    bf16_t* wb = static_cast<bf16_t*>(ctx->alloc(we * 2 + 64));
    struct Free {
        rfi_ctx* c; void* a; void* b;
        ~Free() { (void)hipStreamSynchronize(c->stream); try { c->release(a); c->release(b); } catch (...) {} }
    } fr{ctx, xp, wb};
    RFI_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(xp) + xe * 2, 0, 64, ctx->stream));
    RFI_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(wb) + we * 2, 0, 64, ctx->stream));
    const int64_t ps = (int64_t)plane_chunks(c.Cin) * P * 16;
    launch_act_split(ctx, c.x, pix, c.Cin, c.xf, P, xp, ps);
    launch_weights_to_wb_one(ctx, WBDesc{c.w, wb, 9, c.Cout, c.Cin, {c.Cin, 0}, P});
    PConvArgs a;
    a.x[0] = PlaneSeg{xp, ps, plane_chunks(c.Cin)};
    a.nseg = 1; a.P = P;
    a.N = c.N; a.H = c.H; a.W = c.W; a.Hin = c.Hin; a.Win = c.Win; a.Cout = c.Cout;
    a.wB = wb; a.bias = c.bias;
    a.y = c.y.p; a.y_pstride = c.y.pstride;
    a.Hout = c.Hout; a.Wout = c.Wout; a.osy = c.osy; a.osx = c.osx; a.ooy = c.ooy; a.oox = c.oox;
    a.stats = c.stats; a.stats_max_records = c.stats_max_records;
    a.algo_flops = c.algo_flops >= 0 ? c.algo_flops : 2.0 * c.N * c.H * c.W * (double)c.Cout * 9 * c.Cin;
    launch_pconv(ctx, a);
    c.stats = a.stats;
    c.stats_records = a.stats_records;
}

}  // namespace rfi
