// NHWC implicit-GEMM "conv-like" contraction on the gfx950 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: bit-exact fmaf chain, 256 FLOP/clk/CU).
//
//   y[n,oy,ox,co] = bias[co] + sum_{tap=(r,s)} sum_ci T(X)[n, oy*S+r-pad, ox*S+s-pad, ci] * Wf[tap][co][ci]
//
// GEMM view: M = N*H*W output pixels, N = Cout, K = taps*Cin.  One workgroup (256 threads = 4
// waves) owns a TH x TW spatial tile of one image and BN output channels.  Per 16-channel K-chunk
// it stages, through registers, into LDS:
//   * the input HALO tile ((TH*S+R-S) x (TW*S+R-S) pixels x 16 ch) -- every tap then reads the same
//     LDS tile at a shifted pixel offset, so the input crosses L2->LDS once, not taps times;
//     the producing layer's BatchNorm-apply + ReLU is applied in flight (InXform), borders are
//     zero-filled AFTER that transform;
//   * the filter tile ([taps][BN][16 ch]).
// Rows are padded to 20 floats: fragment reads are ds_read_b128 (4 consecutive k per lane) and
// conflict-free for consecutive pixels (slot = 5*i mod 16 is a bijection).
// MFMA operand mapping (32x32x2): lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; a lane's
// b128 holds k = kb*8 + 4*(l>>5) + {0..3}; step t multiplies element t of both operands, so A and
// B agree on k without any shuffle.  Workgroups are persistent over several spatial tiles and run a
// software pipeline over (tile, chunk) items: the next item's global loads (unconditional, issued
// back to back) are in flight under the current item's MFMAs and are transformed + written to LDS
// afterwards (issue-early / write-late), across tile boundaries as well.
#include <algorithm>
#include <vector>

#include "planes.hpp"

namespace rfi {

void launch_conv_direct(rfi_ctx* ctx, const ConvArgs& a);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 4 floats (one lane's k-quad of a fragment) -> 4 bf16, round-to-nearest-even (2 x v_cvt_pk_bf16_f32)
__device__ __forceinline__ s16x4 pack_bf16(f32x4 x) {
    const f32x2 lo = {x.x, x.y}, hi = {x.z, x.w};
    u32x2 r;
    r.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
    r.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
    return __builtin_bit_cast(s16x4, r);
}

// ---- "3 x bf16" float32 emulation.  A float32 value is the exact sum of three bf16 pieces (8 + 8 + 8
// significand bits): h = bf16(v), m = bf16(v - h), l = bf16(v - h - m).  A product a*b is then the sum of nine
// piece products, each exact in the float32 accumulator; the three smallest (m*l, l*m, l*l <= 2^-24 |a b|) are
// dropped, i.e. an error of the size of ONE float32 rounding per product.  Six v_mfma_f32_32x32x16_bf16
// (32 cycles each, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x the matrix-pipe
// throughput at float32-level accuracy (tools/mfma_rate.hip: 142 vs 2425 TFLOP/s raw).
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
__device__ __forceinline__ Split3 split3(f32x4 x0, f32x4 x1) {      // 8 consecutive k of one lane
    unsigned h[4], m[4], l[4];
    split_pair(x0.x, x0.y, h[0], m[0], l[0]);
    split_pair(x0.z, x0.w, h[1], m[1], l[1]);
    split_pair(x1.x, x1.y, h[2], m[2], l[2]);
    split_pair(x1.z, x1.w, h[3], m[3], l[3]);
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

constexpr int KC = 16;    // channels per K-chunk
constexpr int KCP = 20;   // padded LDS row (floats)

constexpr int WROW3 = 24;  // filter row of the 3 x bf16 path: [h 32 B | m 32 B | l 32 B], pre-split in HBM (ConvArgs::w3)
constexpr int HROW3 = 28;  // halo row of the 3 x bf16 path: [h 32 B | m 32 B | l 32 B | pad 16 B] = 112 B, conflict-free b128 reads

template <int R, int S, int TH, int TW, int BN, int WM, int WN, int PREC = 0>
struct Cfg {
    static constexpr int HROW = PREC >= 2 ? HROW3 : KCP;   // floats per halo-tile row
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH * S + R - S;
    static constexpr int HW = TW * S + R - S;
    static constexpr int HP = HH * HW;
    static constexpr int NTAP = R * R;
    static constexpr int MT = BM / WM / 32;       // 32-pixel m-tiles per wave
    static constexpr int NTL = BN / WN / 32;      // 32-channel n-tiles per wave
    static constexpr int HALO_ITEMS = (HP * 4 + 255) / 256;
    static constexpr int W_ITEMS = (NTAP * BN * (PREC >= 2 ? 6 : 4) + 255) / 256;
    static constexpr int WROW = PREC >= 2 ? WROW3 : KCP;   // floats per filter-tile row
    static constexpr int WQ = PREC >= 2 ? 6 : 4;           // float4 pieces per filter row (3 x bf16: h | m | l of 16 channels)
    static constexpr int STAGE_FLOATS = HP * HROW + NTAP * BN * WROW;
    static constexpr int EPI_FLOATS = 4 * 32 * 36;            // epilogue transpose scratch (4 waves)
    static constexpr int LDS_FLOATS0 = STAGE_FLOATS > EPI_FLOATS ? STAGE_FLOATS : EPI_FLOATS;
    static constexpr int LDS_FLOATS = (LDS_FLOATS0 + 1) & ~1;          // the fp64 statistics area follows, 8-byte aligned
    static constexpr int STAT_DOUBLES = 4 * NTL * 32 * 2;              // [wave][n-tile][channel][sum, sumsq]
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(32 % TW == 0 || TW % 32 == 0, "TW must divide or be a multiple of 32");
};

// BF: bf16 compute mode -- the SAME fp32 tiles in HBM and LDS; each lane's k-quad of an operand fragment is
// rounded to bf16 in registers and ONE v_mfma_f32_32x32x8_bf16 (fp32 accumulate) replaces the four
// fp32 32x32x2 MFMAs of that quad (identical k-to-lane mapping: lane half lh holds k = 4 lh .. 4 lh + 3).
// PREC: 0 float32 MFMA, 1 bf16 (above), 2 float32 emulated by 3 x bf16 (Split3 above), 3 bf16 on the data path of 2:
// operands rounded once at staging time, ONE v_mfma_f32_32x32x16_bf16 per K = 16 block (the gfx950 form, twice the rate
// of the K = 8 instruction of mode 1) -- the h plane of the split is bf16(v), RNE, i.e. the same operand values as mode 1
template <int R, int S, int TH, int TW, int BN, int WM, int WN, int PREC>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a) {
    using C = Cfg<R, S, TH, TW, BN, WM, WN, PREC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_halo = smem;
    float* s_w = smem + C::HP * C::HROW;

    // the conv kernels carry the dependent chain of the step; when a weight-gradient kernel of the side
    // stream shares the SIMD, the arbiter should prefer these waves
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- persistent workgroup: which spatial tiles are mine.  Blocks b and b+8 share an XCD
    // (round-robin dispatch), so each XCD label gets one contiguous range of tiles and its
    // workgroups stride through it: neighbouring tiles (shared halos, same filters) hit one L2.
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    const int G = gridDim.x;
    int t_begin, t_count, j, gx;
    if (G >= 8 && (G & 7) == 0) {
        const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
        j = blockIdx.x >> 3;
        gx = G >> 3;
        t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_count = q8 + (xcd < r8 ? 1 : 0);
    } else {
        j = blockIdx.x; gx = G; t_begin = 0; t_count = ntiles;
    }
    const int my_tiles = (j < t_count) ? (t_count - j + gx - 1) / gx : 0;
    const int n0 = blockIdx.y * BN;
    // BatchNorm statistics of the output (forward convs of the model): every workgroup leaves ONE
    // fp64 (sum, sumsq) record per channel of its channel tile, merged by bn_finalize in record order
    double* const st_out = a.stats ? a.stats + ((size_t)blockIdx.x * a.Cout + n0) * 2 : nullptr;
    if (my_tiles == 0) {                             // uniform for the workgroup, before any barrier
        if (st_out && tid < BN && n0 + tid < a.Cout) st_out[tid * 2] = st_out[tid * 2 + 1] = 0.0;
        return;
    }
    double* const s_stat = reinterpret_cast<double*>(smem + C::LDS_FLOATS);
    if (st_out)
        for (int i = tid; i < C::STAT_DOUBLES; i += 256) s_stat[i] = 0.0;   // visible after the first barrier below
    const int z = blockIdx.z;
    const float* __restrict__ wbase =
        PREC >= 2 ? a.w3 + (a.zgroups > 1 ? (size_t)z * a.Cout * ((a.Cin + KC - 1) / KC) * WROW3 : 0)
                  : a.w + (a.zgroups > 1 ? (size_t)z * a.Cout * a.Cin : 0);
    const int ooy = a.zgroups > 1 ? (z >> 1) : a.ooy;
    const int oox = a.zgroups > 1 ? (z & 1) : a.oox;
    const int q4 = (tid & 3) * 4;                    // channel offset of this thread's float4 in a chunk
    const int nchunks = (a.Cin + KC - 1) / KC;
    const int nitems = my_tiles * nchunks;

    // ---- staging descriptors.  Loads are UNCONDITIONAL (invalid items read a safe address and are
    // zeroed when written to LDS) so the compiler can issue them back to back and wait once.
    // (32-bit float offsets: the launcher checks that every tensor has < 2^31 elements)
    int h_off[C::HALO_ITEMS];
    unsigned h_mask = 0;                             // bit it: item `it` of the tile being loaded is in range
#pragma unroll
    for (int it = 0; it < C::HALO_ITEMS; ++it) h_off[it] = 0;
    int w_off[C::W_ITEMS];
    const int nchunks_w = (a.Cin + KC - 1) / KC;      // 3 x bf16: records of 24 floats per (tap, cout, chunk)
#pragma unroll
    for (int it = 0; it < C::W_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int row = idx / C::WQ, q = idx % C::WQ;       // row = tap*BN + nloc
        w_off[it] = -1;
        if (idx < C::NTAP * BN * C::WQ) {
            const int tap = row / BN, nloc = row % BN;
            if (n0 + nloc < a.Cout) {
                if constexpr (PREC >= 2) w_off[it] = ((tap * a.Cout + n0 + nloc) * nchunks_w) * WROW3 + q * 4;
                else w_off[it] = (tap * a.Cout + n0 + nloc) * a.Cin + q * 4;
            }
        }
    }
    const int lds_item0 = (tid >> 2) * KCP + (tid & 3) * 4;   // LDS float offset of item 0; item it: + it*64*KCP
    struct Tile { int n, oy0, ox0; };
    auto tile_of = [&](int k) {
        const int t = t_begin + j + k * gx;
        Tile r;
        r.ox0 = (t % tiles_x) * TW;
        r.oy0 = ((t / tiles_x) % tiles_y) * TH;
        r.n = t / (tiles_x * tiles_y);
        return r;
    };
    // tile-independent halo coordinates of every item (row<<16 | col); out-of-range items get a
    // row that fails every bounds test
    int h_rc[C::HALO_ITEMS];
#pragma unroll
    for (int it = 0; it < C::HALO_ITEMS; ++it) {
        const int idx = tid + it * 256, pix = idx >> 2;
        h_rc[it] = idx < C::HP * 4 ? (((pix / C::HW) << 16) | (pix % C::HW)) : (0x4000 << 16);
    }
    auto setup_halo = [&](const Tile& t) {
        const int iy0 = t.oy0 * S - a.pad, ix0 = t.ox0 * S - a.pad, nb = t.n * a.Hin;
        h_mask = 0;
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it) {
            const int iy = iy0 + (h_rc[it] >> 16), ix = ix0 + (h_rc[it] & 0xffff);
            const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            h_off[it] = ok ? ((nb + iy) * a.Win + ix) * a.x.pstride + q4 : 0;
            h_mask |= (ok ? 1u : 0u) << it;
        }
    };

    f32x4 hreg[C::HALO_ITEMS];
    f32x4 wreg[C::W_ITEMS];
    f32x4 screg = {1.f, 1.f, 1.f, 1.f}, shreg = {0.f, 0.f, 0.f, 0.f};
    bool cv_l = false;                               // channel validity of the chunk in the registers

    auto issue_loads = [&](int c0) {
        cv_l = (c0 + q4) < a.Cin;                    // Cin % 4 == 0 is a launch precondition
        const int cc = cv_l ? c0 : 0;
        if (a.xf.scale) {
            screg = *reinterpret_cast<const f32x4*>(a.xf.scale + cc + (cv_l ? q4 : 0));
            shreg = *reinterpret_cast<const f32x4*>(a.xf.shift + cc + (cv_l ? q4 : 0));
        }
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it)
            hreg[it] = *reinterpret_cast<const f32x4*>(
                a.x.p + (unsigned)((((h_mask >> it) & 1u) && cv_l) ? h_off[it] + cc : 0));   // never past the tensor
#pragma unroll
        for (int it = 0; it < C::W_ITEMS; ++it) {
            if constexpr (PREC >= 2)     // pre-split records: chunk c0/16 of the row, always in range (zero padded)
                wreg[it] = *reinterpret_cast<const f32x4*>(wbase + (unsigned)(w_off[it] >= 0 ? w_off[it] + (c0 / KC) * WROW3 : 0));
            else
                wreg[it] = *reinterpret_cast<const f32x4*>(wbase + (unsigned)((w_off[it] >= 0 && cv_l) ? w_off[it] + cc : 0));
        }
    };
    auto store_chunk = [&]() {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it) {
            f32x4 v = hreg[it];
            if (a.xf.scale) {
                v = v * screg + shreg;
                // ReLU / LeakyReLU in one branch-free form: max(v, slope*v), slope in [0, 1) (0 = ReLU)
                if (a.xf.relu) v = __builtin_elementwise_max(v, v * a.xf.slope);
            }
            v = (((h_mask >> it) & 1u) && cv_l) ? v : zero;      // zero padding AFTER the transform
            if constexpr (PREC >= 2) {
                // split ONCE per element here (each halo element feeds up to 9 taps): three bf16 planes per row
                // (mode 3: only the h plane = the bf16-rounded operand is stored and read)
                unsigned h0, m0, l0, h1, m1, l1;
                split_pair(v.x, v.y, h0, m0, l0);
                split_pair(v.z, v.w, h1, m1, l1);
                if (tid + it * 256 < C::HP * 4) {
                    float* row = s_halo + ((tid >> 2) + it * 64) * C::HROW + (tid & 3) * 2;   // 4 channels = 8 B per plane
                    *reinterpret_cast<u32x2*>(row) = u32x2{h0, h1};
                    if constexpr (PREC == 2) {
                        *reinterpret_cast<u32x2*>(row + 8) = u32x2{m0, m1};
                        *reinterpret_cast<u32x2*>(row + 16) = u32x2{l0, l1};
                    }
                }
            } else {
                if (tid + it * 256 < C::HP * 4) *reinterpret_cast<f32x4*>(s_halo + lds_item0 + it * 64 * KCP) = v;
            }
        }
#pragma unroll
        for (int it = 0; it < C::W_ITEMS; ++it) {
            if constexpr (PREC >= 2) {      // 6 float4 pieces per 96-byte row: item idx -> row idx/6, piece idx%6
                const int idx = tid + it * 256;
                const f32x4 v = w_off[it] >= 0 ? wreg[it] : zero;
                if (idx < C::NTAP * BN * 6 && (PREC == 2 || idx % 6 < 2))         // (mode 3: the h plane only)
                    *reinterpret_cast<f32x4*>(s_w + (idx / 6) * WROW3 + (idx % 6) * 4) = v;
            } else {
                const f32x4 v = (w_off[it] >= 0 && cv_l) ? wreg[it] : zero;
                if (tid + it * 256 < C::NTAP * BN * 4) *reinterpret_cast<f32x4*>(s_w + lds_item0 + it * 64 * KCP) = v;
            }
        }
    };

    // ---- fragment addresses
    int a_base[C::MT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int p = (wm * C::MT + mt) * 32 + li;
        const int ty = p / TW, tx = p % TW;
        a_base[mt] = ((ty * S) * C::HW + tx * S) * C::HROW + lh * 4;      // lh: k-quad (float32) / 8 bf16 of a plane (3 x bf16)
    }
    int b_base[C::NTL];
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt)
        b_base[nt] = (wn * (BN / WN) + nt * 32 + li) * C::WROW + lh * 4;

    f32x16 acc[C::MT][C::NTL];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // ---- software pipeline over (tile, chunk) items: item q+1's global loads are in flight while
    // item q's MFMAs run; across tile boundaries too, so only the first item of a workgroup's life
    // exposes its load latency
    Tile ctile = tile_of(0);                         // tile being computed
    int lk = 0, lch = 0;                             // load cursor (tile index, chunk)
    setup_halo(ctile);
    issue_loads(0);
#ifdef RFI_DIAG_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0};
#define RFI_T(v) const unsigned long long v = clock64()
#define RFI_ACC(i, a_, b_) st_[i] += (b_) - (a_)
#else
#define RFI_T(v)
#define RFI_ACC(i, a_, b_)
#endif
    for (int q = 0; q < nitems; ++q) {
        RFI_T(t0);
        store_chunk();
        RFI_T(t1);
        __syncthreads();
        RFI_T(t2);
        const int cch = lch;                         // chunk now in LDS
        Tile ltile = ctile;
        if (q + 1 < nitems) {
            if (++lch == nchunks) {
                lch = 0;
                ++lk;
                ltile = tile_of(lk);
                setup_halo(ltile);
            }
        }
        issue_loads(lch * KC);                       // (the very last item is re-read once: harmless)
        RFI_T(t3);
        if constexpr (PREC >= 2) {
            // one K = 16 group per tap: lane half lh holds channels 8 lh .. 8 lh + 7 of the 16-channel chunk row
#pragma unroll
            for (int tap = 0; tap < C::NTAP; ++tap) {
                const int tr = tap / R, ts = tap % R;
                Split3 bs[C::NTL];
#pragma unroll
                for (int nt = 0; nt < C::NTL; ++nt) {
                    const float* bp = s_w + b_base[nt] + tap * BN * WROW3;        // planes h | m | l, 8 floats apart
                    bs[nt] = Split3{__builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(bp)),
                                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(bp + 8)),
                                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(bp + 16))};
                }
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    const float* ap = s_halo + a_base[mt] + (tr * C::HW + ts) * C::HROW;   // planes h | m | l, 8 floats apart
                    const Split3 as{__builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ap)),
                                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ap + 8)),
                                    __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(ap + 16))};
#pragma unroll
                    for (int nt = 0; nt < C::NTL; ++nt) {
                        if constexpr (PREC == 2) acc[mt][nt] = mfma_3xbf16(as, bs[nt], acc[mt][nt]);
                        else acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as.h, bs[nt].h, acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
        for (int tap = 0; tap < C::NTAP; ++tap) {
            const int tr = tap / R, ts = tap % R;
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                f32x4 af[C::MT], bf[C::NTL];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
                    af[mt] = *reinterpret_cast<const f32x4*>(
                        s_halo + a_base[mt] + (tr * C::HW + ts) * C::HROW + kb * 8);
#pragma unroll
                for (int nt = 0; nt < C::NTL; ++nt)
                    bf[nt] = *reinterpret_cast<const f32x4*>(s_w + b_base[nt] + tap * BN * KCP + kb * 8);
                if constexpr (PREC == 1) {
                    s16x4 ah[C::MT], bh[C::NTL];
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt) ah[mt] = pack_bf16(af[mt]);
#pragma unroll
                    for (int nt = 0; nt < C::NTL; ++nt) bh[nt] = pack_bf16(bf[nt]);
#pragma unroll
                    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < C::NTL; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < C::NTL; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][t], bf[nt][t],
                                                                                  acc[mt][nt], 0, 0, 0);
                }
            }
        }
        }
        RFI_T(t4);
        __syncthreads();
        RFI_T(t5);
        if (cch == nchunks - 1) {
            // ---- epilogue of tile `ctile`.  C/D layout of 32x32: col = lane&31 (channel), row =
            // (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel): a lane holds ONE channel of 16 pixels, i.e.
            // 16 dword stores a tile.  64 narrow stores per wave saturate the 6-bit vmcnt queue and the
            // next item's staging then waits on them, so each 32x32 tile is transposed through the
            // (now idle) staging LDS and leaves as 4 x 16-byte-per-lane stores of whole 128-B pixel rows.
            const int xs = a.osx * a.y.pstride;                         // floats per tile column step
            const bool vec_out = (a.Cout & 3) == 0 && (a.y.pstride & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(a.y.p) & 15) == 0;
            float* s_ep = smem + wave * (32 * 36);                      // 4.5 KiB per wave, rows of 36 floats
#pragma unroll
            for (int nt = 0; nt < C::NTL; ++nt) {
                const int co0 = n0 + wn * (BN / WN) + nt * 32;
                f32x4 p1 = {0.f, 0.f, 0.f, 0.f}, p2 = p1;                // this lane's share of sum y, sum y^2
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt) {
                    constexpr int ROWS = (32 + TW - 1) / TW;            // image rows per 32-pixel m-tile
                    const int oyb = ctile.oy0 + ((wm * C::MT + mt) * 32) / TW;
                    if (vec_out) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            s_ep[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[mt][nt][r];
                        const int g4 = (lane & 7) * 4;                  // 4 channels of this lane
                        const int co = co0 + g4;
                        // folded convT phases: channel co of the GEMM is channel cc of output phase zf (fold % 4 == 0,
                        // so a lane's four channels share the phase)
                        const int zf = a.fold ? co / a.fold : 0;
                        const int cc = a.fold ? co - zf * a.fold : co;
                        const int ooy_l = a.fold ? (zf >> 1) : ooy, oox_l = a.fold ? (zf & 1) : oox;
                        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                        if (a.bias && co < a.Cout) b4 = *reinterpret_cast<const f32x4*>(a.bias + cc);
                        // BatchNorm-backward sums of the layer whose activated output this tensor is the gradient of
                        f32x4 bsc = b4, bsh = b4, bmu = b4, bis = b4;
                        if (st_out && a.bwd_y && co < a.Cout) {
                            bsc = *reinterpret_cast<const f32x4*>(a.bwd_scale + co);
                            bsh = *reinterpret_cast<const f32x4*>(a.bwd_shift + co);
                            bmu = *reinterpret_cast<const f32x4*>(a.bwd_mean + co);
                            bis = *reinterpret_cast<const f32x4*>(a.bwd_invstd + co);
                        }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int pp = ps * 8 + (lane >> 3);        // pixel of the m-tile
                            const f32x4 v = *reinterpret_cast<const f32x4*>(s_ep + pp * 36 + g4) + b4;
                            const int oy = oyb + pp / TW, ox = ctile.ox0 + pp % TW;
                            if (co < a.Cout && oy < a.H && ox < a.W) {
                                const unsigned off = (unsigned)(((ctile.n * a.Hout + oy * a.osy + ooy_l) * a.Wout + ox * a.osx + oox_l) *
                                                                a.y.pstride + cc);
                                if (a.y16) {                            // bf16 output (RNE), 8 bytes per lane
                                    const __bf16 q0 = (__bf16)v[0], q1 = (__bf16)v[1], q2 = (__bf16)v[2], q3 = (__bf16)v[3];
                                    u32x2 pk;
                                    pk[0] = (unsigned)__builtin_bit_cast(unsigned short, q0) | ((unsigned)__builtin_bit_cast(unsigned short, q1) << 16);
                                    pk[1] = (unsigned)__builtin_bit_cast(unsigned short, q2) | ((unsigned)__builtin_bit_cast(unsigned short, q3) << 16);
                                    *reinterpret_cast<u32x2*>(a.y16 + off) = pk;
                                } else {
                                    *reinterpret_cast<f32x4*>(a.y.p + off) = v;
                                }
                                if (st_out) {
                                    if (a.bwd_y) {
                                        const f32x4 yv = *reinterpret_cast<const f32x4*>(a.bwd_y + off);
                                        const f32x4 z = yv * bsc + bsh, xh = (yv - bmu) * bis;
                                        f32x4 dz;
#pragma unroll
                                        for (int e = 0; e < 4; ++e) dz[e] = z[e] > 0.0f ? v[e] : v[e] * a.bwd_slope;
                                        p1 += dz;
                                        p2 += dz * xh;
                                    } else {
                                        p1 += v;
                                        p2 += v * v;
                                    }
                                }
                            }
                        }
                    } else {
                        const int co = co0 + li;
                        const bool cok = co < a.Cout;
                        const float bv = (a.bias && cok) ? a.bias[co] : 0.0f;
                        const int lane_x = ctile.ox0 + 4 * lh;
#pragma unroll
                        for (int dyi = 0; dyi < ROWS; ++dyi) {
                            const int oy = oyb + dyi;
                            const bool yok = cok && oy < a.H;
                            float* rowp = a.y.p + (unsigned)(((ctile.n * a.Hout + oy * a.osy + ooy) * a.Wout + oox) *
                                                                 a.y.pstride + co);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int c = (r & 3) + 8 * (r >> 2);   // row of reg r for lh == 0
                                if (c / TW != dyi) continue;            // compile-time filter
                                const int ox = lane_x + (c % TW);
                                if (yok && ox < a.W) rowp[(unsigned)(ox * xs)] = acc[mt][nt][r] + bv;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
                }
                if (st_out && vec_out) {
                    // lanes l, l^8, l^16, l^32 hold the same 4 channels of different pixels: fold them
                    // (<= 128 fp32 terms per channel), then continue in fp64 in this wave's LDS slots
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int o = 8; o < 64; o <<= 1) {
                            p1[e] += __shfl_xor(p1[e], o, 64);
                            p2[e] += __shfl_xor(p2[e], o, 64);
                        }
                    }
                    if (lane < 8) {
                        double* d = s_stat + ((wave * C::NTL + nt) * 32 + lane * 4) * 2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            d[e * 2] += (double)p1[e];
                            d[e * 2 + 1] += (double)p2[e];
                        }
                    }
                }
            }
            // the staging LDS is about to be overwritten by the next item: wait for this wave's scratch reads only
            // (a __syncthreads() would also wait for the output stores to retire: one HBM write latency per tile)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ctile = ltile;
        }
#ifdef RFI_DIAG_STAMPS
        {
            const unsigned long long t6 = clock64();
            RFI_ACC(0, t0, t1); RFI_ACC(1, t1, t2); RFI_ACC(2, t2, t3); RFI_ACC(3, t3, t4); RFI_ACC(4, t4, t5);
            RFI_ACC(5, t5, t6);
        }
#endif
    }
    if (st_out) {
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout) {          // channel tid of the tile: waves (wm, wn_c) cover it
            const int wn_c = tid / (BN / WN), nt_c = (tid % (BN / WN)) / 32, cl = tid % 32;
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const double* d = s_stat + (((m * WN + wn_c) * C::NTL + nt_c) * 32 + cl) * 2;
                t1 += d[0];
                t2 += d[1];
            }
            st_out[tid * 2] = t1;
            st_out[tid * 2 + 1] = t2;
        }
    }
#ifdef RFI_DIAG_STAMPS
    if (a.stamps && (tid & 63) == 0) {
        unsigned long long* o = a.stamps + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 32 + wave * 8;
        for (int i = 0; i < 6; ++i) o[i] = st_[i];
        o[6] = nitems;
    }
#endif
}

// LDS bytes -> resident workgroups per CU (160 KiB LDS, <= 8 waves/SIMD is never the limit here)
static int occupancy_for(size_t lds_bytes) {
    static const int cap = getenv("RFI_CONV_OCC") ? atoi(getenv("RFI_CONV_OCC")) : 0;     // tuning experiments
    if (cap > 0) return cap;
    int o = (int)((160 * 1024) / lds_bytes);
    return o < 1 ? 1 : (o > 3 ? 3 : o);   // VGPRs: <= 215 for the 72 KB tiles (2 waves/SIMD), <= 151 for the others (3)
}

template <int R, int S, int TH, int TW, int BN, int WM, int WN, int PREC = 0>
void launch_cfg(rfi_ctx* ctx, ConvArgs& a) {
    if constexpr (PREC == 0) {
        if (a.bf16 && bf16_k16()) return launch_cfg<R, S, TH, TW, BN, WM, WN, 3>(ctx, a);
        if (a.bf16) return launch_cfg<R, S, TH, TW, BN, WM, WN, 1>(ctx, a);
        if (a.bf16x3) return launch_cfg<R, S, TH, TW, BN, WM, WN, 2>(ctx, a);
    }
    using C = Cfg<R, S, TH, TW, BN, WM, WN, PREC>;
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    const int ychunks = (int)cdiv(a.Cout, BN);
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float) + (size_t)C::STAT_DOUBLES * sizeof(double);
    // persistent grid: about (256 CUs x occupancy) workgroups in total, a multiple of 8 along x,
    // tiles spread evenly over the workgroups of each XCD
    const int gmax = std::max(8, (256 * occupancy_for(lds)) / (ychunks * a.zgroups));
    const int tx = (int)cdiv(ntiles, 8);                       // tiles per XCD label (max)
    const int per = (int)cdiv(tx, std::max(1, gmax / 8));      // tiles per workgroup
    int gx = (int)cdiv(tx, per);
    int G = 8 * gx;
    if (ntiles < 8) G = ntiles;                                // tiny problems: one tile per workgroup
    // fused output statistics need the 16-byte store path and one channel-tile column per record
    const bool vec_out = (a.Cout & 3) == 0 && (a.y.pstride & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y.p) & 15) == 0;
    if (a.stats && vec_out && a.zgroups == 1 && G <= a.stats_max_records) a.stats_records = G;
    else a.stats = nullptr;
    dim3 grid(G, ychunks, a.zgroups);
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_igemm_kernel<R, S, TH, TW, BN, WM, WN, PREC>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    });
#ifdef RFI_DIAG_STAMPS
    {   // diagnostic build: run with per-phase cycle stamps and print the per-wave averages
        ConvArgs b = a;
        const size_t nw = (size_t)G * ychunks * a.zgroups * 32;
        RFI_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&b.stamps), nw * 8));
        RFI_CHECK_HIP(hipMemsetAsync(b.stamps, 0, nw * 8, ctx->stream));
        hipLaunchKernelGGL((conv_igemm_kernel<R, S, TH, TW, BN, WM, WN, PREC>), grid, dim3(256), lds, ctx->stream, b);
        std::vector<unsigned long long> h(nw);
        RFI_CHECK_HIP(hipMemcpyAsync(h.data(), b.stamps, nw * 8, hipMemcpyDeviceToHost, ctx->stream));
        RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        double s[7] = {0, 0, 0, 0, 0, 0, 0};
        size_t cnt = 0;
        for (size_t w = 0; w < nw / 8; ++w) {
            if (h[w * 8 + 6] == 0) continue;
            for (int i = 0; i < 7; ++i) s[i] += (double)h[w * 8 + i];
            ++cnt;
        }
        std::fprintf(stderr, "[stamps] conv<%d,%d,%d,%d,%d> N%d %dx%d %d->%d grid %dx%d items/wg %.1f | per item (cycles): "
                     "store %.0f bar1 %.0f loads %.0f mfma %.0f bar2 %.0f epi %.0f\n", R, S, TH, TW, BN, a.N, a.H, a.W,
                     a.Cin, a.Cout, G, ychunks, s[6] / cnt, s[0] / s[6], s[1] / s[6], s[2] / s[6], s[3] / s[6], s[4] / s[6],
                     s[5] / s[6]);
        RFI_CHECK_HIP(hipFree(b.stamps));
        return;
    }
#endif
    hipLaunchKernelGGL((conv_igemm_kernel<R, S, TH, TW, BN, WM, WN, PREC>), grid, dim3(256), lds, ctx->stream,
                       a);
    check_launch("conv_igemm");
}

// tile choice: by output width (TW = 32 / 16 / 8) and channel tile (BN = 32 for Cout <= 32, else 64).
// Where the grid stays >= 2 workgroups per CU, a workgroup takes twice the pixels (wave tile 64x64 or
// 128x32): twice the MFMAs between barriers and half the filter staging per output.
template <int R, int S>
void dispatch_tiles(rfi_ctx* ctx, ConvArgs& a) {
    if constexpr (R == 1) {
        // transposed conv with its four phases folded into the channels (a GEMM: K = Cin, N = 4 Cout): one staged
        // input chunk feeds a 128-channel tile, i.e. 2-4x the MFMAs per barrier of the per-phase launches
        if (a.fold && a.Cout >= 128) {
            if (a.W >= 32) return launch_cfg<R, S, 4, 32, 128, 2, 2>(ctx, a);
            if (a.W >= 16) return launch_cfg<R, S, 8, 16, 128, 2, 2>(ctx, a);
            return launch_cfg<R, S, 8, 8, 128, 1, 4>(ctx, a);
        }
    }
    const int ychunks64 = (int)cdiv(a.Cout, 64) * a.zgroups;
    auto big_ok = [&](int th, int tw, int ych) {
        return (int64_t)a.N * cdiv(a.H, th) * cdiv(a.W, tw) * ych >= 512;
    };
    if constexpr (S == 1) {          // (a stride-2 halo of the double tile would not fit the LDS)
      if (!a.bf16x3 && !(a.bf16 && bf16_k16())) {   // (the bf16 halo planes of a double tile would leave one workgroup per CU)
        if (a.W >= 32 && a.Cout <= 32 && big_ok(16, 32, a.zgroups)) return launch_cfg<R, S, 16, 32, 32, 4, 1>(ctx, a);
        if (a.W >= 32 && a.Cout > 32 && big_ok(8, 32, ychunks64)) return launch_cfg<R, S, 8, 32, 64, 4, 1>(ctx, a);
        if (a.W >= 16 && a.W < 32 && a.Cout > 32 && big_ok(16, 16, ychunks64))
            return launch_cfg<R, S, 16, 16, 64, 4, 1>(ctx, a);
      }
    }
    if (a.W >= 32) {
        if (a.Cout <= 32) launch_cfg<R, S, 8, 32, 32, 4, 1>(ctx, a);
        else launch_cfg<R, S, 4, 32, 64, 2, 2>(ctx, a);
    } else if (a.W >= 16) {
        if (a.Cout <= 32) launch_cfg<R, S, 8, 16, 32, 4, 1>(ctx, a);
        else launch_cfg<R, S, 8, 16, 64, 2, 2>(ctx, a);
    } else {
        if (a.Cout <= 32) launch_cfg<R, S, 16, 8, 32, 4, 1>(ctx, a);
        else launch_cfg<R, S, 8, 8, 64, 2, 2>(ctx, a);
    }
}

}  // namespace

size_t weights_x3_floats(int taps, int Cout, int Cin) {
    return (size_t)taps * Cout * ((Cin + KC - 1) / KC) * WROW3;
}
namespace {
// w [taps][Cout][Cin] float32 -> records [taps][Cout][chunk][h16 | m16 | l16] bf16 (zero beyond Cin)
__global__ void weights_to_x3_kernel(const float* __restrict__ w, int64_t rows, int Cin, int nchunks,
                                     float* __restrict__ out) {
    const int64_t total = rows * nchunks * 4;                    // one thread per 4 channels
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        const int64_t rc = i >> 2;
        const int chunk = (int)(rc % nchunks);
        const int64_t row = rc / nchunks;
        const int c = chunk * KC + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < Cin) v = *reinterpret_cast<const f32x4*>(w + row * Cin + c);      // Cin % 4 == 0
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(v.x, v.y, h0, m0, l0);
        split_pair(v.z, v.w, h1, m1, l1);
        float* rec = out + rc * WROW3 + q * 2;
        *reinterpret_cast<u32x2*>(rec) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(rec + 8) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(rec + 16) = u32x2{l0, l1};
    }
}
// the same for MANY filter tensors in one launch (all layers, both layouts, once per optimiser step)
__global__ void weights_to_x3_batched_kernel(const X3Desc* __restrict__ descs) {
    const X3Desc d = descs[blockIdx.y];
    const int64_t total = d.rows * d.nchunks * 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 3);
        const int64_t rc = i >> 2;
        const int chunk = (int)(rc % d.nchunks);
        const int64_t row = rc / d.nchunks;
        const int c = chunk * KC + q * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < d.Cin) v = *reinterpret_cast<const f32x4*>(d.src + row * d.Cin + c);
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(v.x, v.y, h0, m0, l0);
        split_pair(v.z, v.w, h1, m1, l1);
        float* rec = d.dst + rc * WROW3 + q * 2;
        *reinterpret_cast<u32x2*>(rec) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(rec + 8) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(rec + 16) = u32x2{l0, l1};
    }
}
}  // namespace
void launch_weights_to_x3_batched(rfi_ctx* ctx, const X3Desc* descs_dev, int n, double total_bytes) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, total_bytes);
    hipLaunchKernelGGL(weights_to_x3_batched_kernel, dim3(128, n), dim3(256), 0, ctx->stream, descs_dev);
    check_launch("weights_to_x3_batched");
}
void launch_weights_to_x3(rfi_ctx* ctx, const float* w, int taps, int Cout, int Cin, float* out) {
    const int nchunks = (Cin + KC - 1) / KC;
    const int64_t rows = (int64_t)taps * Cout, total = rows * nchunks * 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)rows * Cin * 4 + (double)rows * nchunks * 96);
    int64_t blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(weights_to_x3_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w, rows, Cin, nchunks, out);
    check_launch("weights_to_x3");
}

bool conv_mfma_eligible(const ConvArgs& a) {
    if (a.Cin % 4 != 0 || a.x.pstride % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.x.p) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15)) return false;
    if (a.xf.scale && ((reinterpret_cast<uintptr_t>(a.xf.scale) & 15) ||
                       (reinterpret_cast<uintptr_t>(a.xf.shift) & 15)))
        return false;
    // R = 2, S = 1: the 2x2 form of a 3x3 / stride-2 conv on its space-to-depth input (pad 1: forward, top/left
    // padding only; pad 0: its input gradient, which reads rows y, y + 1) -- resnet_kernels.hip
    const bool known = (a.R == 3 && a.S == 1 && a.pad == 1) || (a.R == 1 && a.S == 1 && a.pad == 0) ||
                       (a.R == 2 && a.S == 2 && a.pad == 0) || (a.R == 2 && a.S == 1 && (a.pad == 1 || a.pad == 0));
    return known;
}

void launch_conv(rfi_ctx* ctx, ConvArgs& a, int impl) {
    if (impl == IMPL_PLANES_X3 || impl == IMPL_PLANES_BF16) {
        RFI_REQUIRE(a.R == 3 && a.S == 1 && a.zgroups == 1, "conv: the plane kernels cover 3x3 stride-1 convolutions");
        a.planes = impl == IMPL_PLANES_X3 ? 3 : 1;
        impl = IMPL_MFMA;
    }
    if (impl == IMPL_MFMA_BF16) {
        a.bf16 = true;
        impl = IMPL_MFMA;
    }
    if (impl == IMPL_MFMA_BF16X3) {
        a.bf16x3 = true;
        a.wB3 = nullptr;                // (this implementation was asked for by name)
        impl = IMPL_MFMA;
    }
    if (impl == IMPL_WS_X3 && conv_stem_eligible(a)) {       // kernel-level API / tests: the stem kernel by name
        a.bf16x3 = true;
        launch_conv_stem(ctx, a);
        return;
    }
    if (impl == IMPL_WS_X3 || impl == IMPL_WS_BF16) {       // kernel-level API / tests: a temporary B-operand-order copy of the filters
        const int P = impl == IMPL_WS_X3 ? 3 : 1;
        const bool gw = !conv_ws_eligible(a) && gemm_ws_eligible(a);
        RFI_REQUIRE(gw || conv_ws_eligible(a), "conv: shape not eligible for the wave-specialised kernels");
        RFI_REQUIRE(!gw || P == 3, "conv: the wave-specialised GEMM kernel exists in the 3 x bf16 arithmetic only");
        a.bf16x3 = P == 3;
        a.bf16 = P == 1;
        // conv_ws: [9][Cout][Cin]; gemm_ws: the transposed conv forward as ONE tap of 4 Cout channels, its input gradient
        // as four taps, a 1x1 conv as one tap
        const int taps = gw ? (a.R == 2 ? 4 : 1) : 9, cout = gw && a.zgroups == 4 ? 4 * a.Cout : a.Cout;
        const size_t we = wb_elems(taps, cout, a.Cin, 0, P);
        bf16_t* wb = static_cast<bf16_t*>(ctx->alloc(we * 2 + 64));
        struct Free {
            rfi_ctx* c; void* p;
            ~Free() { (void)hipStreamSynchronize(c->stream); try { c->release(p); } catch (...) {} }
        } fr{ctx, wb};
        launch_weights_to_wb_one(ctx, WBDesc{a.w, wb, taps, cout, a.Cin, {a.Cin, 0}, P});
        if (gw) launch_gemm_ws(ctx, a, wb);
        else launch_conv_ws(ctx, a, wb, P);
        return;
    }
    static const bool no_ws = getenv("RFI_NO_WS") != nullptr;                    // A/B runs: round 2's kernels
    static const bool no_gw = getenv("RFI_NO_GW") != nullptr;
    static const bool no_stem = getenv("RFI_NO_STEM") != nullptr;
    if (impl == IMPL_AUTO && a.bf16x3 && !no_stem && !no_ws && conv_stem_eligible(a)) {
        launch_conv_stem(ctx, a);
        return;
    }
    if (impl == IMPL_AUTO && a.bf16x3 && a.wB3 && !no_ws && conv_ws_eligible(a)) {
        launch_conv_ws(ctx, a, a.wB3, 3);
        return;
    }
    if (impl == IMPL_AUTO && a.bf16 && !bf16_k16() && a.wB1 && !no_ws && conv_ws_eligible(a)) {
        launch_conv_ws(ctx, a, a.wB1, 1);
        return;
    }
    if (impl == IMPL_AUTO && a.bf16x3 && a.wB3 && !no_ws && !no_gw && gemm_ws_eligible(a)) {
        launch_gemm_ws(ctx, a, a.wB3);
        return;
    }
    // (a layer of the plain U-Net that carries a wave-specialised filter copy keeps no pre-split records up to date --
    // rfi_model::ws_set clears ConvArgs::w3 for it -- so a shape those kernels declined runs below on a temporary split copy)
    RFI_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "conv: empty shape");
    RFI_REQUIRE(a.x.pstride >= a.Cin && a.y.pstride >= a.Cout, "conv: pixel stride smaller than channels");
    RFI_REQUIRE(a.zgroups == 1 || (a.zgroups == 4 && a.R == 1), "conv: zgroups only for convT forward");
    RFI_REQUIRE((int64_t)a.N * a.Hin * a.Win * a.x.pstride < (int64_t)1 << 31 &&
                    (int64_t)a.N * a.Hout * a.Wout * a.y.pstride < (int64_t)1 << 31 &&
                    (int64_t)a.R * a.R * a.Cin * a.Cout * a.zgroups < (int64_t)1 << 31,
                "conv: tensor too large for 32-bit element offsets");
    if (a.planes) {                 // IMPL_PLANES_*: the plane kernels (conv_planes.hip) through temporary plane copies
        launch_pconv_from_f32(ctx, a, a.planes);
        return;
    }
    const bool ok = conv_mfma_eligible(a);
    if (impl == IMPL_MFMA) RFI_REQUIRE(ok, "conv: shape/alignment not eligible for the MFMA kernel");
    if (impl == IMPL_DIRECT || !ok) {
        RFI_REQUIRE(!a.y16, "conv: bfloat16 output needs the MFMA kernel");
        launch_conv_direct(ctx, a);
        return;
    }
    // 3 x bf16: the filters are read pre-split (ConvArgs::w3); callers that only have float32 weights (the
    // kernel-level API) get a temporary split copy
    float* tmp_w3 = nullptr;
    if ((a.bf16x3 || (a.bf16 && bf16_k16())) && !a.w3) {
        const int taps = a.R * a.R * a.zgroups;
        tmp_w3 = static_cast<float*>(ctx->alloc(weights_x3_floats(taps, a.Cout, a.Cin) * sizeof(float)));
        launch_weights_to_x3(ctx, a.w, taps, a.Cout, a.Cin, tmp_w3);
        a.w3 = tmp_w3;
    }
    struct FreeTmp {
        rfi_ctx* c; float* p; ConvArgs& a;
        ~FreeTmp() {
            if (p) {
                (void)hipStreamSynchronize(c->stream);
                try { c->release(p); } catch (...) {}
                a.w3 = nullptr;
            }
        }
    } free_tmp{ctx, tmp_w3, a};
    const double flops = a.algo_flops >= 0 ? a.algo_flops
                                           : 2.0 * a.N * a.H * a.W * (double)a.Cout * a.R * a.R * a.Cin * a.zgroups;
    std::string label;
    if (ctx->profiling)
        label = "conv R" + std::to_string(a.R) + "S" + std::to_string(a.S) + " N" + std::to_string(a.N) + " " +
                std::to_string(a.H) + "x" + std::to_string(a.W) + " " + std::to_string(a.Cin) + "->" +
                std::to_string(a.Cout) + (a.xf.scale ? " xf" : "") + (a.zgroups > 1 ? " z4" : "") +
                (a.bf16 ? " bf16" : "") + (a.bf16x3 ? " 3xbf16" : "");
    // algorithmic HBM bytes: every input, output and filter element once (float32 tensors)
    const double bytes = 4.0 * ((double)a.N * a.Hin * a.Win * a.Cin + (double)a.R * a.R * a.zgroups * a.Cin * a.Cout) +
                         (a.y16 ? 2.0 : 4.0) * a.N * a.H * a.W * a.zgroups * a.Cout;
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, bytes, label);
    static const bool no_fold = getenv("RFI_NO_CONVT_FOLD") != nullptr;          // A/B runs
    const bool y_ok = a.y16 ? (reinterpret_cast<uintptr_t>(a.y16) & 7) == 0 : (reinterpret_cast<uintptr_t>(a.y.p) & 15) == 0;
    RFI_REQUIRE(!a.y16 || (a.R == 1 && a.zgroups == 4 && (a.Cout & 3) == 0 && (a.y.pstride & 3) == 0 && y_ok && !a.stats),
                "conv: bfloat16 output exists for the folded transposed convolution only (Cout % 4 == 0, aligned output)");
    if (a.R == 1 && a.zgroups == 4 && (!no_fold || a.y16) && (a.Cout & 3) == 0 && (a.y.pstride & 3) == 0 && y_ok && !a.stats) {
        ConvArgs f = a;                 // [4][Cout][Cin] filters (and their 3 x bf16 records) ARE [4 Cout][Cin]
        f.fold = a.Cout;
        f.Cout = 4 * a.Cout;
        f.zgroups = 1;
        dispatch_tiles<1, 1>(ctx, f);
        return;
    }
    if (a.R == 3) dispatch_tiles<3, 1>(ctx, a);
    else if (a.R == 1) dispatch_tiles<1, 1>(ctx, a);
    else if (a.S == 1) dispatch_tiles<2, 1>(ctx, a);
    else dispatch_tiles<2, 2>(ctx, a);
}

// bf16 mode on float32 tensors: round 1's K = 8 instruction on float32 LDS tiles (default), or -- RFI_BF16_K16=1 -- the
// K = 16 instruction on the split path's data flow (mode 3).  Measured (round 2): mode 3 is SLOWER, 28.1 vs 24.0 ms per
// UNetResNet18 step at 1 x 1024^2 and 8.9 vs 7.5 ms per mask-head step: what it gains in matrix rate it loses to the
// split path's staging (conversion at staging time, 112-byte rows, no double tiles within the LDS budget).  The bf16
// mode that does use the K = 16 instruction well is the plane data flow (conv_planes.hip).
bool bf16_k16() {
    static const bool k16 = getenv("RFI_BF16_K16") != nullptr;
    return k16;
}


}  // namespace rfi
