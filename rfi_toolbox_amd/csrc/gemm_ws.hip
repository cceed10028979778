// Wave-specialised "one tap per K chunk" contractions on float32 NHWC tensors in the float32-by-3xbf16 arithmetic: the
// transposed convolution (k2, s2) forward with its four output phases folded into the channel dimension, its input
// gradient (a 2x2 / stride-2 convolution: every tap reads its own pixels, nothing is shared between taps), and plain
// 1x1 convolutions.
//
//   forward   y[n, 2h+a, 2w+b, co] = bias[co] + sum_ci T(X)[n,h,w,ci] * W[(a,b)][co][ci]        GEMM: M = N h w, K = Cin, N = 4 Cout
//   gradient  dx[n,h,w,ci] = sum_{a,b,co} dUp[n, 2h+a, 2w+b, co] * Wd[(a,b)][ci][co]            GEMM: M = N h w, K = 4 Cout, N = Cin
//
// Same roles as conv_ws.hip (waves 4-7 load, transform, split and stage; waves 0-3 multiply), but without a halo there is
// no reuse of a split element across taps: the only reuse is across OUTPUT CHANNELS, so a workgroup takes a tile of 128
// pixels against NB = 8 blocks of 32 channels (consumer wave = 32 pixels x 256 channels, 48 MFMAs per 16-channel item)
// and the producers split 128 x 16 values per item (2 float4 per thread).  Everything is double-buffered ([A0 | A1 | B0 |
// B1] = 2 x 14 + 2 x 24 KiB), one barrier per item.
//
// [r4] PWV: the number of producer waves.  These contractions are bound by the producers' split (a producer wave issues one
// vector instruction per ~14 cycles beside an MFMA wave: latency, not issue bandwidth), so where the consumers' registers
// leave room (NB <= 4: 64 accumulator registers) every SIMD hosts TWO producer waves (PWV = 8, 768 threads, <= 168
// registers): one float4 per producer thread and item instead of two.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ws_common.hpp"

namespace rfi {
namespace {

using namespace ws;

struct GwDev {
    const float* x;                   // input tensor (float32 NHWC)
    unsigned x_bytes;                 // its size (< 2^31: an offset of 2^31 reads as zero)
    int x_ps;
    const float* scale;               // InXform of the input (null: identity)
    const float* shift;
    float slope;
    int M, H, W;                      // output-grid pixels N * H * W; grid height / width
    int gather;                       // 0: input pixel = output-grid pixel; 1: the 2x2 / stride-2 gather (input grid 2H x 2W, tap (a, b) reads (2h + a, 2w + b))
    int nkc_tap, ntap;                // 16-channel chunks per tap; taps (1 or 4)
    const bf16_t* wB;                 // [tap][kc][cb][3 planes][64 lanes][8] bf16 (planes.hpp)
    int ncb, Ngemm;                   // 32-channel blocks / channels of the GEMM's N
    const float* bias;
    float* y;
    int y_ps;
    int fold;                         // > 0 (forward): GEMM channel c is channel c % fold of output phase c / fold, written to pixel
                                      // (2h + phase / 2, 2w + phase % 2) of a 2H x 2W grid
};

constexpr int ROWB = 112;             // bytes per pixel in LDS: [h 32 B | m 32 B | l 32 B | pad 16 B]
constexpr int TM = 128;               // pixels per tile
constexpr int A_BYTES = TM * ROWB;

template <int NB, int PWV = 4>
struct GwCfg {
    static constexpr int NPIECE = 3 * NB;
    static constexpr int B_BYTES = NPIECE * 1024;
    static constexpr int B_ITEMS = (NPIECE + PWV - 1) / PWV;
    static constexpr int A_ITEMS = 8 / PWV;                          // float4 per producer thread and item (128 pixels x 4 groups)
    static_assert(PWV == 4 || PWV == 8, "4 or 8 producer waves");
    static constexpr int B_OFF = 2 * A_BYTES;
    static constexpr int LDS_BYTES = B_OFF + 2 * B_BYTES;
};

// XF: 0 none, 1 relu(x * scale + shift), 2 x * scale + shift followed by max(v, v * slope)
template <int NB, int XF, int PWV = 4>
__global__ __launch_bounds__(256 + 64 * PWV) void gemm_ws_kernel(GwDev d) {
    using C = GwCfg<NB, PWV>;
    constexpr int AI = C::A_ITEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int ntiles = (d.M + TM - 1) / TM;
    const int GX = gridDim.x;
    int t_begin, t_count, j, gx;
    if (GX >= 8 && (GX & 7) == 0) {                  // XCD-contiguous tile ranges (as conv_ws.hip)
        const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
        j = blockIdx.x >> 3;
        gx = GX >> 3;
        t_begin = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        t_count = q8 + (xcd < r8 ? 1 : 0);
    } else {
        j = blockIdx.x; gx = GX; t_begin = 0; t_count = ntiles;
    }
    const int my_tiles = (j < t_count) ? (t_count - j + gx - 1) / gx : 0;
    if (my_tiles == 0) return;                       // uniform for the workgroup, before any barrier
    const int cb0 = blockIdx.y * NB, n0 = cb0 * 32;
    const int nkc = d.nkc_tap * d.ntap;
    const int nitems = my_tiles * nkc;
    auto tile_m0 = [&](int k) { return (t_begin + j + k * gx) * TM; };
    // input-tensor element offset of output-grid pixel m (tap (0, 0) in the gather form)
    auto in_pixel = [&](int m) -> unsigned {
        if (!d.gather) return (unsigned)m * (unsigned)d.x_ps;
        const int w = m % d.W, t = m / d.W, h = t % d.H, n = t / d.H;
        return (unsigned)(((n * 2 * d.H + 2 * h) * 2 * d.W + 2 * w)) * (unsigned)d.x_ps;
    };

    if (wave >= 4) {
        // =============================================================== producers
        const int ptid = tid - 256, pw = wave - 4;
        const unsigned q4b = (unsigned)(ptid & 3) * 16u;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.x), 0, (int)d.x_bytes, 0x00020000);
        unsigned voff[AI], vmsk[AI];
        auto setup_tile = [&](int m0) {
#pragma unroll
            for (int it = 0; it < AI; ++it) {
                const int m = m0 + (ptid >> 2) + it * 64;
                const bool ok = m < d.M;
                voff[it] = ok ? in_pixel(m) * 4u + q4b : 0x80000000u;
                vmsk[it] = ok ? 0xffffffffu : 0u;
            }
        };
        u32x4 raw[AI];
        f32x4 screg = {1.f, 1.f, 1.f, 1.f}, shreg = {0.f, 0.f, 0.f, 0.f};
        unsigned pl[AI][6];
        // chunk jc of a tile: tap jc / nkc_tap, channels 16 (jc % nkc_tap) ...; the tap's pixel offset and the channel
        // offset are wave-uniform: they ride in the scalar offset of the buffer load
        auto issue_loads = [&](int jc) {
            const int tap = jc / d.nkc_tap, c0 = (jc - tap * d.nkc_tap) * 16;
            const int soff = (((tap >> 1) * 2 * d.W + (tap & 1)) * d.x_ps + c0) * 4;
            if constexpr (XF != 0) {
                screg = *reinterpret_cast<const f32x4*>(d.scale + c0 + (ptid & 3) * 4);
                shreg = *reinterpret_cast<const f32x4*>(d.shift + c0 + (ptid & 3) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int it = 0; it < AI; ++it) raw[it] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff[it], soff, 0);
        };
        auto split_all = [&]() {
#pragma unroll
            for (int it = 0; it < AI; ++it) {
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
                split_pair(v.x, v.y, pl[it][0], pl[it][2], pl[it][4]);
                split_pair(v.z, v.w, pl[it][1], pl[it][3], pl[it][5]);
            }
        };
        auto write_all = [&](int buf) {
            unsigned char* const sA = smem + buf * A_BYTES + (ptid >> 2) * ROWB + (ptid & 3) * 8;
#pragma unroll
            for (int it = 0; it < AI; ++it) {
                unsigned char* row = sA + it * 64 * ROWB;
                *reinterpret_cast<u32x2*>(row) = u32x2{pl[it][0], pl[it][1]};
                *reinterpret_cast<u32x2*>(row + 32) = u32x2{pl[it][2], pl[it][3]};
                *reinterpret_cast<u32x2*>(row + 64) = u32x2{pl[it][4], pl[it][5]};
            }
        };
        const unsigned char* const wb = reinterpret_cast<const unsigned char*>(d.wB);
        auto issue_B = [&](int jc, int buf) {        // (tap, kc) of wB = (jc / nkc_tap, jc % nkc_tap) = chunk jc of the K order
            unsigned char* const sB = smem + C::B_OFF + buf * C::B_BYTES;
#pragma unroll
            for (int i = 0; i < C::B_ITEMS; ++i) {
                const int p = pw + PWV * i;
                if (p < C::NPIECE) {
                    const int nb = p / 3, plane = p - nb * 3;
                    const int cb = cb0 + nb < d.ncb ? cb0 + nb : d.ncb - 1;       // (a block beyond the last: products never stored)
                    const unsigned off = (unsigned)(((jc * d.ncb + cb) * 3 + plane) * 1024 + lane * 16);
                    __builtin_amdgcn_global_load_lds((gbl_void*)(wb + off), (lds_void*)(sB + p * 1024), 16, 0, 0);
                }
            }
        };
        int lk = 0, lch = 0;
        setup_tile(tile_m0(0));
        issue_loads(0);
        for (int q = -1; q < nitems; ++q) {
            if (q + 1 < nitems) {
                split_all();                             // (waits for the loads of item q + 1)
                write_all((q + 1) & 1);                  // the consumers left this buffer an item ago
                __builtin_amdgcn_sched_barrier(0);
                issue_B(lch, (q + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (q + 2 < nitems) {
                    if (++lch == nkc) {
                        lch = 0;
                        ++lk;
                        setup_tile(tile_m0(lk));
                    }
                }
                issue_loads(lch);                        // (the very last item is re-read once: harmless)
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt<AI>();                        // all but the item's own loads: the DMA pieces have landed
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            wg_barrier();
        }
    } else {
        // =============================================================== consumers
        const int cw = wave, li = lane & 31, lh = lane >> 5;
        const int a_base = (cw * 32 + li) * ROWB + lh * 16;
        const int b_base = lane * 16;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
        auto run_item = [&](auto first_c, const unsigned char* sA, const unsigned char* sB) {
            constexpr bool FIRST = decltype(first_c)::value;
            bf16x8 af[3], bf[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bf16x8*>(sA + a_base + p * 32);
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[0][p] = *reinterpret_cast<const bf16x8*>(sB + p * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if (nb + 1 < NB) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[(nb + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(sB + ((nb + 1) * 3 + p) * 1024);
                }
                f32x16 c = acc[nb];
                if (FIRST)
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                acc[nb] = mma3(af, bf[nb & 1], c);
#pragma unroll
                for (int i = 0; i < 6; ++i) {            // the next block's three reads between this block's six MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (nb + 1 < NB && (i & 1)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // fast epilogue addressing: a 32-pixel block = whole rows of one image (W a power of two below 32, H W % 32 == 0) or
        // a piece of one row (W % 32 == 0); dense outputs: consecutive pixels are y_ps apart whatever the row structure
        const bool pow2 = (d.W & (d.W - 1)) == 0;
        const bool fast = !d.fold || (d.W % 32 == 0) || (pow2 && d.W < 32 && (d.H * d.W) % 32 == 0);
        unsigned loff[16];                               // this lane's part of a store offset: its pixel of the block + its channel
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int po = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (d.fold) {
                const int row = d.W >= 32 ? 0 : po / d.W, col = d.W >= 32 ? po : po % d.W;
                loff[r] = (unsigned)((row * 2 * 2 * d.W + 2 * col) * d.y_ps + li);
            } else {
                loff[r] = (unsigned)(po * d.y_ps + li);
            }
        }
        int ck = 0, cch = 0;
        wg_barrier();                                    // item 0 is staged
        for (int q = 0; q < nitems; ++q) {
            const unsigned char* const sA = smem + (q & 1) * A_BYTES;
            const unsigned char* const sB = smem + C::B_OFF + (q & 1) * C::B_BYTES + b_base;
            if (cch == 0) run_item(std::true_type{}, sA, sB);
            else run_item(std::false_type{}, sA, sB);
            if (++cch == nkc) {
                // ---- epilogue of the tile.  C/D layout of 32x32: col = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2)
                // + 4 * (lane >> 5) (pixel): a wave store writes 128 contiguous bytes of two pixels
                const int m0 = tile_m0(ck);
                const int mb = m0 + cw * 32;             // first pixel of this wave's 32-pixel block (wave-uniform)
                if (fast && m0 + TM <= d.M && n0 + NB * 32 <= d.Ngemm) {
                    // the block lies inside one image, in whole rows or inside one: its 16 per-lane offsets were computed
                    // once (loff); what changes per tile and per channel block is wave-uniform
                    unsigned base;
                    if (d.fold) {
                        const int w0 = mb % d.W, t = mb / d.W, h0 = t % d.H, n = t / d.H;
                        base = (unsigned)(((n * 2 * d.H + 2 * h0) * 2 * d.W + 2 * w0)) * (unsigned)d.y_ps;
                    } else {
                        base = (unsigned)mb * (unsigned)d.y_ps;
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int cg0 = n0 + nb * 32;
                        int cout0 = cg0;
                        unsigned poff = 0;
                        if (d.fold) {                    // fold % 32 == 0: the block lies inside one phase
                            const int z = cg0 / d.fold;
                            cout0 = cg0 - z * d.fold;
                            poff = (unsigned)(((z >> 1) * 2 * d.W + (z & 1)) * d.y_ps);
                        }
                        const float bv = d.bias ? d.bias[cout0 + li] : 0.0f;
                        const unsigned uoff = base + poff + (unsigned)cout0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) d.y[uoff + loff[r]] = acc[nb][r] + bv;
                    }
                } else {
                    unsigned obase[16];                  // output element offset of this lane's 16 pixels (phase (0, 0), channel 0)
                    unsigned omask = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const bool ok = m < d.M;
                        omask |= (ok ? 1u : 0u) << r;
                        if (d.fold) {
                            const int w = m % d.W, t = m / d.W, h = t % d.H, n = t / d.H;
                            obase[r] = (unsigned)(((n * 2 * d.H + 2 * h) * 2 * d.W + 2 * w)) * (unsigned)d.y_ps;
                        } else {
                            obase[r] = (unsigned)m * (unsigned)d.y_ps;
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int cg0 = n0 + nb * 32;    // first GEMM channel of the block (wave-uniform)
                        if (cg0 < d.Ngemm) {
                            int cout0 = cg0;
                            unsigned poff = 0;
                            if (d.fold) {
                                const int z = cg0 / d.fold;
                                cout0 = cg0 - z * d.fold;
                                poff = (unsigned)(((z >> 1) * 2 * d.W + (z & 1)) * d.y_ps);
                            }
                            const bool cok = cg0 + li < d.Ngemm;
                            const float bv = (d.bias && cok) ? d.bias[cout0 + li] : 0.0f;
                            const unsigned coff = poff + (unsigned)(cout0 + li);
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if (cok && ((omask >> r) & 1u)) d.y[obase[r] + coff] = acc[nb][r] + bv;
                        }
                    }
                }
                cch = 0;
                ++ck;
            }
            wg_barrier();
        }
    }
}

template <int NB, int XF, int PWV = 4>
void launch_gw(rfi_ctx* ctx, GwDev& d) {
    using C = GwCfg<NB, PWV>;
    const int ntiles = (int)cdiv(d.M, TM);
    const int ycols = (int)cdiv(d.ncb, NB);
    static const int wgs = getenv("RFI_GW_WGS") ? atoi(getenv("RFI_GW_WGS")) : 256;
    const int gmax = std::max(8, wgs / ycols);
    const int tx = (int)cdiv(ntiles, 8);
    const int per = (int)cdiv(tx, std::max(1, gmax / 8));
    int GX = 8 * (int)cdiv(tx, per);
    if (ntiles < 8) GX = ntiles;
    static PerDeviceOnce attr_once;
    attr_once.run(ctx->device, [&] {
        RFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ws_kernel<NB, XF, PWV>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    });
    hipLaunchKernelGGL((gemm_ws_kernel<NB, XF, PWV>), dim3(GX, ycols), dim3(256 + 64 * PWV), C::LDS_BYTES, ctx->stream, d);
    check_launch("gemm_ws");
}

}  // namespace

// the transposed conv forward (R = 1, four phase groups), its input gradient (R = 2, S = 2, pad 0) and plain 1x1 convs
bool gemm_ws_eligible(const ConvArgs& a) {
    const bool fwd_t = a.R == 1 && a.S == 1 && a.pad == 0 && a.zgroups == 4 && a.osy == 2 && a.osx == 2 && a.Hout == 2 * a.H &&
                       a.Wout == 2 * a.W && a.Hin == a.H && a.Win == a.W && a.Cout % 32 == 0;
    const bool dgrad_t = a.R == 2 && a.S == 2 && a.pad == 0 && a.zgroups == 1 && a.Hin == 2 * a.H && a.Win == 2 * a.W &&
                         a.Hout == a.H && a.Wout == a.W && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0;
    const bool one = a.R == 1 && a.S == 1 && a.pad == 0 && a.zgroups == 1 && a.Hin == a.H && a.Win == a.W && a.Hout == a.H &&
                     a.Wout == a.W && a.osy == 1 && a.osx == 1 && a.ooy == 0 && a.oox == 0;
    if (!(fwd_t || dgrad_t || one) || a.fold || a.y16 || a.bwd_y || a.stats) return false;
    if (a.Cin % 16 != 0 || a.x.pstride % 4 != 0 || (reinterpret_cast<uintptr_t>(a.x.p) & 15)) return false;
    if (a.xf.scale && ((reinterpret_cast<uintptr_t>(a.xf.scale) & 15) || (reinterpret_cast<uintptr_t>(a.xf.shift) & 15))) return false;
    if ((int64_t)a.N * a.Hin * a.Win * a.x.pstride * 4 >= (int64_t)1 << 31 || (int64_t)a.N * a.Hout * a.Wout * a.y.pstride >= (int64_t)1 << 31) return false;
    return true;
}

// wB3: forward -- the [4 Cout][Cin] filters as ONE tap; gradient -- [4 taps][Cout (= layer's Cin)][Cin (= layer's Cout)]
// (launch_weights_to_wb with P = 3, one K segment)
void launch_gemm_ws(rfi_ctx* ctx, ConvArgs& a, const bf16_t* wB3) {
    RFI_REQUIRE(gemm_ws_eligible(a), "gemm_ws: shape not eligible");
    GwDev d;
    d.x = a.x.p; d.x_ps = a.x.pstride;
    d.x_bytes = (unsigned)((int64_t)a.N * a.Hin * a.Win * a.x.pstride * 4);
    d.scale = a.xf.scale; d.shift = a.xf.shift;
    d.slope = a.xf.relu == 0 ? 1.0f : a.xf.slope;
    d.M = a.N * a.H * a.W; d.H = a.H; d.W = a.W;
    d.gather = a.R == 2 ? 1 : 0;
    d.ntap = a.R == 2 ? 4 : 1;
    d.nkc_tap = a.Cin / 16;
    d.wB = wB3;
    d.Ngemm = a.Cout * a.zgroups;
    d.ncb = (d.Ngemm + 31) / 32;
    d.fold = a.zgroups == 4 ? a.Cout : 0;
    d.bias = a.bias;
    d.y = a.y.p; d.y_ps = a.y.pstride;
    const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * d.M * (double)d.Ngemm * a.Cin * d.ntap;
    std::string label;
    if (ctx->profiling)
        label = std::string("gemm_ws ") + (a.zgroups == 4 ? "convT" : a.R == 2 ? "convT-dgrad" : "1x1") + " N" + std::to_string(a.N) + " " +
                std::to_string(a.H) + "x" + std::to_string(a.W) + " " + std::to_string(a.Cin) + "->" + std::to_string(a.Cout) +
                (a.xf.scale ? " xf" : "") + " 3xbf16";
    const double bytes = 4.0 * ((double)a.N * a.Hin * a.Win * a.Cin + (double)d.ntap * a.Cin * d.Ngemm) + 4.0 * d.M * (double)d.Ngemm;
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, bytes, label);
    const int xf = !a.xf.scale ? 0 : (a.xf.relu == 1 || (a.xf.relu == 2 && a.xf.slope == 0.0f)) ? 1 : 2;
    // 8 blocks per workgroup where the grid still covers the chip, else 4 (2 for a 64-channel output)
    const bool wide = d.ncb > 4 && cdiv(d.M, TM) * cdiv(d.ncb, 8) >= 192;
    static const bool pwv8 = !(getenv("RFI_GW_PWV") && atoi(getenv("RFI_GW_PWV")) == 4);      // (A/B: RFI_GW_PWV=4)
    // (measured and not kept: 64-channel column groups with eight producer waves where 128-channel ones leave half the CUs
    // without a workgroup -- the 8 x 8 maps: the kernel alone 59 -> 53 us, the step +0.7 %: the idle CUs are where the side
    // stream's weight gradient runs)
    if (wide) {
        if (xf == 0) launch_gw<8, 0>(ctx, d);
        else if (xf == 1) launch_gw<8, 1>(ctx, d);
        else launch_gw<8, 2>(ctx, d);
    } else if (d.ncb <= 2) {
        if (xf == 0) launch_gw<2, 0>(ctx, d);
        else if (xf == 1) launch_gw<2, 1>(ctx, d);
        else launch_gw<2, 2>(ctx, d);
    } else if (pwv8) {
        if (xf == 0) launch_gw<4, 0, 8>(ctx, d);
        else if (xf == 1) launch_gw<4, 1, 8>(ctx, d);
        else launch_gw<4, 2, 8>(ctx, d);
    } else {
        if (xf == 0) launch_gw<4, 0>(ctx, d);
        else if (xf == 1) launch_gw<4, 1>(ctx, d);
        else launch_gw<4, 2>(ctx, d);
    }
}

}  // namespace rfi
