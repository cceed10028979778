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
// B agree on k without any shuffle.  The next chunk's global loads are issued before the MFMAs of
// the current chunk and written to LDS after them (issue-early / write-late).
#include "kernels.hpp"

namespace rfi {

void launch_conv_direct(rfi_ctx* ctx, const ConvArgs& a);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;    // channels per K-chunk
constexpr int KCP = 20;   // padded LDS row (floats)

template <int R, int S, int TH, int TW, int BN, int WM, int WN>
struct Cfg {
    static constexpr int BM = TH * TW;
    static constexpr int HH = TH * S + R - S;
    static constexpr int HW = TW * S + R - S;
    static constexpr int HP = HH * HW;
    static constexpr int NTAP = R * R;
    static constexpr int MT = BM / WM / 32;       // 32-pixel m-tiles per wave
    static constexpr int NTL = BN / WN / 32;      // 32-channel n-tiles per wave
    static constexpr int HALO_ITEMS = (HP * 4 + 255) / 256;
    static constexpr int W_ITEMS = (NTAP * BN * 4 + 255) / 256;
    static constexpr int LDS_FLOATS = HP * KCP + NTAP * BN * KCP;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "tile/wave mismatch");
    static_assert(32 % TW == 0 || TW % 32 == 0, "TW must divide or be a multiple of 32");
};

template <int R, int S, int TH, int TW, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
    using C = Cfg<R, S, TH, TW, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_halo = smem;
    float* s_w = smem + C::HP * KCP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- which tile: XCD-aware bijective remap of blockIdx.x (blocks b and b+8 share an XCD)
    const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
    const int ntiles = a.N * tiles_y * tiles_x;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    }
    const int tx_i = bid % tiles_x;
    const int ty_i = (bid / tiles_x) % tiles_y;
    const int n = bid / (tiles_x * tiles_y);
    const int oy0 = ty_i * TH, ox0 = tx_i * TW;
    const int n0 = blockIdx.y * BN;
    const int z = blockIdx.z;
    const float* __restrict__ wbase = a.w + (a.zgroups > 1 ? (size_t)z * a.Cout * a.Cin : 0);
    const int ooy = a.zgroups > 1 ? (z >> 1) : a.ooy;
    const int oox = a.zgroups > 1 ? (z & 1) : a.oox;

    // ---- per-thread staging descriptors (independent of the chunk)
    long h_off[C::HALO_ITEMS];    // global float offset of the item's pixel (+4q), or -1
    int h_lds[C::HALO_ITEMS];
#pragma unroll
    for (int it = 0; it < C::HALO_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int pix = idx >> 2, q = idx & 3;
        h_lds[it] = pix * KCP + q * 4;
        h_off[it] = -1;
        if (idx < C::HP * 4) {
            const int hy = pix / C::HW, hx = pix % C::HW;
            const int iy = oy0 * S - a.pad + hy, ix = ox0 * S - a.pad + hx;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                h_off[it] = (((long)n * a.Hin + iy) * a.Win + ix) * a.x.pstride + q * 4;
        } else {
            h_lds[it] = -1;
        }
    }
    long w_off[C::W_ITEMS];
    int w_lds[C::W_ITEMS];
#pragma unroll
    for (int it = 0; it < C::W_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int row = idx >> 2, q = idx & 3;       // row = tap*BN + nloc
        w_lds[it] = row * KCP + q * 4;
        w_off[it] = -1;
        if (idx < C::NTAP * BN * 4) {
            const int tap = row / BN, nloc = row % BN;
            if (n0 + nloc < a.Cout) w_off[it] = ((long)tap * a.Cout + n0 + nloc) * a.Cin + q * 4;
        } else {
            w_lds[it] = -1;
        }
    }
    const int q4 = (tid & 3) * 4;    // channel offset of this thread's float4 within a chunk

    f32x4 hreg[C::HALO_ITEMS];
    f32x4 wreg[C::W_ITEMS];

    auto load_chunk = [&](int c0) {
        const bool cvalid = (c0 + q4) < a.Cin;     // Cin % 4 == 0 is a launch precondition
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.xf.scale && cvalid) {
            sc = *reinterpret_cast<const f32x4*>(a.xf.scale + c0 + q4);
            sh = *reinterpret_cast<const f32x4*>(a.xf.shift + c0 + q4);
        }
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (h_off[it] >= 0 && cvalid) {
                v = *reinterpret_cast<const f32x4*>(a.x.p + h_off[it] + c0);
                if (a.xf.scale) {
                    v = v * sc + sh;
                    if (a.xf.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
            }
            hreg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < C::W_ITEMS; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (w_off[it] >= 0 && cvalid) v = *reinterpret_cast<const f32x4*>(wbase + w_off[it] + c0);
            wreg[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < C::HALO_ITEMS; ++it)
            if (h_lds[it] >= 0) *reinterpret_cast<f32x4*>(s_halo + h_lds[it]) = hreg[it];
#pragma unroll
        for (int it = 0; it < C::W_ITEMS; ++it)
            if (w_lds[it] >= 0) *reinterpret_cast<f32x4*>(s_w + w_lds[it]) = wreg[it];
    };

    // ---- fragment addresses
    int a_base[C::MT];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
        const int p = (wm * C::MT + mt) * 32 + li;
        const int ty = p / TW, tx = p % TW;
        a_base[mt] = ((ty * S) * C::HW + tx * S) * KCP + lh * 4;
    }
    int b_base[C::NTL];
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt)
        b_base[nt] = (wn * (BN / WN) + nt * 32 + li) * KCP + lh * 4;

    f32x16 acc[C::MT][C::NTL];
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int nchunks = (a.Cin + KC - 1) / KC;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = (ch + 1) < nchunks;
        if (more) load_chunk((ch + 1) * KC);
#pragma unroll
        for (int tap = 0; tap < C::NTAP; ++tap) {
            const int tr = tap / R, ts = tap % R;
#pragma unroll
            for (int kb = 0; kb < KC / 8; ++kb) {
                f32x4 af[C::MT], bf[C::NTL];
#pragma unroll
                for (int mt = 0; mt < C::MT; ++mt)
                    af[mt] = *reinterpret_cast<const f32x4*>(
                        s_halo + a_base[mt] + (tr * C::HW + ts) * KCP + kb * 8);
#pragma unroll
                for (int nt = 0; nt < C::NTL; ++nt)
                    bf[nt] = *reinterpret_cast<const f32x4*>(s_w + b_base[nt] + tap * BN * KCP + kb * 8);
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
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of 32x32: col = lane&31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) {
        const int co = n0 + wn * (BN / WN) + nt * 32 + li;
        const bool cok = co < a.Cout;
        const float bv = (a.bias && cok) ? a.bias[co] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int p = (wm * C::MT + mt) * 32 + row;
                const int oy = oy0 + p / TW, ox = ox0 + p % TW;
                if (cok && oy < a.H && ox < a.W) {
                    const long opix = ((long)n * a.Hout + (oy * a.osy + ooy)) * a.Wout + (ox * a.osx + oox);
                    a.y.p[opix * a.y.pstride + co] = acc[mt][nt][r] + bv;
                }
            }
        }
    }
}

template <int R, int S, int TH, int TW, int BN, int WM, int WN>
void launch_cfg(rfi_ctx* ctx, const ConvArgs& a) {
    using C = Cfg<R, S, TH, TW, BN, WM, WN>;
    const int tiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    dim3 grid(tiles, (unsigned)cdiv(a.Cout, BN), a.zgroups);
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        RFI_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_igemm_kernel<R, S, TH, TW, BN, WM, WN>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<R, S, TH, TW, BN, WM, WN>), grid, dim3(256), lds, ctx->stream,
                       a);
    check_launch("conv_igemm");
}

template <int R, int S>
void dispatch_tiles(rfi_ctx* ctx, const ConvArgs& a) {
    // tile shape by output width, channel tile by Cout
    if (a.W >= 32) {
        if (a.Cout <= 32) launch_cfg<R, S, 8, 32, 32, 4, 1>(ctx, a);
        else              launch_cfg<R, S, 4, 32, 64, 2, 2>(ctx, a);
    } else if (a.W >= 16) {
        if (a.Cout <= 32) launch_cfg<R, S, 8, 16, 32, 4, 1>(ctx, a);
        else              launch_cfg<R, S, 8, 16, 64, 2, 2>(ctx, a);
    } else {
        if (a.Cout <= 32) launch_cfg<R, S, 16, 8, 32, 4, 1>(ctx, a);
        else              launch_cfg<R, S, 8, 8, 64, 2, 2>(ctx, a);
    }
}

}  // namespace

bool conv_mfma_eligible(const ConvArgs& a) {
    if (a.Cin % 4 != 0 || a.x.pstride % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(a.x.p) & 15) || (reinterpret_cast<uintptr_t>(a.w) & 15)) return false;
    if (a.xf.scale && ((reinterpret_cast<uintptr_t>(a.xf.scale) & 15) ||
                       (reinterpret_cast<uintptr_t>(a.xf.shift) & 15)))
        return false;
    const bool known = (a.R == 3 && a.S == 1 && a.pad == 1) || (a.R == 1 && a.S == 1 && a.pad == 0) ||
                       (a.R == 2 && a.S == 2 && a.pad == 0);
    return known;
}

void launch_conv(rfi_ctx* ctx, ConvArgs& a, int impl) {
    RFI_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "conv: empty shape");
    RFI_REQUIRE(a.x.pstride >= a.Cin && a.y.pstride >= a.Cout, "conv: pixel stride smaller than channels");
    RFI_REQUIRE(a.zgroups == 1 || (a.zgroups == 4 && a.R == 1), "conv: zgroups only for convT forward");
    const bool ok = conv_mfma_eligible(a);
    if (impl == IMPL_MFMA) RFI_REQUIRE(ok, "conv: shape/alignment not eligible for the MFMA kernel");
    if (impl == IMPL_DIRECT || !ok) {
        launch_conv_direct(ctx, a);
        return;
    }
    const double flops = a.algo_flops >= 0 ? a.algo_flops
                                           : 2.0 * a.N * a.H * a.W * (double)a.Cout * a.R * a.R * a.Cin * a.zgroups;
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, 0);
    if (a.R == 3) dispatch_tiles<3, 1>(ctx, a);
    else if (a.R == 1) dispatch_tiles<1, 1>(ctx, a);
    else dispatch_tiles<2, 2>(ctx, a);
}

}  // namespace rfi
