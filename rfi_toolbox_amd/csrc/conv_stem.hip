// The first 3x3 conv of a network (stem): Cin = 3 padded to 4, Cout = 32 or 64, stride 1, pad 1, float32 tensors, the
// float32-by-3xbf16 arithmetic.  Reference op: nn.Conv2d(in_channels, features, 3, padding=1), the first layer of
// DoubleConv (/root/reference/rfi_toolbox/models/unet.py:10-12).
//
// The general kernels run a K = 16-channel chunk per tap, so the stem pays nine taps x 16 channels of which 3 (4) carry
// data: 54 MFMAs per 32 x 32 block product at 20 TFLOP/s (77 us of the U-Net step, round 3).  Here the K dimension is
// (tap, channel) = 9 x 4 = 36, padded to 48 = THREE k-steps of 16: lane half kh of k-step s holds taps 4 s + 2 kh and
// 4 s + 2 kh + 1, four channels each -- two 8-byte LDS reads from a halo image of [pixel][4 channels] bf16 per plane, at
// the two taps' pixel offsets: the im2col happens in the address of the fragment read, nothing is materialised.  18 MFMAs
// per block; the filters (9 x Cout x 4 floats) are split once per wave into registers.  What is left is the output
// stream: 64 x 128 x 128 x 32 floats = 134 MB, i.e. the kernel is bound by HBM writes (~30 us at 4.5 TB/s).
//
// Plain (not wave-specialised) workgroups of 4 waves, one 8 x 32-pixel tile at a time, 8.2 KB of LDS: several workgroups
// per CU overlap each other's staging, MFMAs and stores.
#include <algorithm>

#include "ws_common.hpp"

namespace rfi {
namespace {

using namespace ws;

struct StemDev {
    const float* x;                   // [N][H][W][4] float32 (channel 3 zero)
    int N, H, W;
    const float* w;                   // [9][Cout][4] float32
    const float* bias;
    float* y;                         // [N][H][W][y_ps]
    int y_ps, Cout;
    double* stats;                    // [gridDim.x][Cout][2] fp64 (sum y, sum y^2) records, or null
};

constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2, HP = HH * HW;      // tile and its halo
constexpr int PLANE = HP * 8;                                               // bytes of one plane image: [pixel][4 bf16]

template <int NT>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemDev d) {
    __shared__ __attribute__((aligned(16))) unsigned char s_img[3 * PLANE];
    __shared__ double s_stat[4 * NT * 64 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;

    // ---- the filters as B fragments, split once: k = 16 s + 8 kh + j  <->  tap 4 s + 2 kh + (j >> 2), channel j & 3
    bf16x8 bfr[NT][3][3];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            f32x4 v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tap = 4 * s + 2 * kh + h, co = nt * 32 + li;
                v[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tap < 9 && co < d.Cout) v[h] = *reinterpret_cast<const f32x4*>(d.w + ((size_t)tap * d.Cout + co) * 4);
            }
            unsigned pl[3][4];
            split_pair(v[0].x, v[0].y, pl[0][0], pl[1][0], pl[2][0]);
            split_pair(v[0].z, v[0].w, pl[0][1], pl[1][1], pl[2][1]);
            split_pair(v[1].x, v[1].y, pl[0][2], pl[1][2], pl[2][2]);
            split_pair(v[1].z, v[1].w, pl[0][3], pl[1][3], pl[2][3]);
#pragma unroll
            for (int p = 0; p < 3; ++p) bfr[nt][s][p] = __builtin_bit_cast(bf16x8, u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]});
        }
    // halo-pixel offsets (bytes inside a plane image) of this lane's two taps per k-step, relative to the block's first pixel
    int toff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tap = std::min(4 * s + 2 * kh + h, 8);                   // (taps 9 .. 11: zero filters, any valid address)
            toff[s][h] = ((tap / 3) * HW + tap % 3 + li) * 8;
        }
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = (d.bias && nt * 32 + li < d.Cout) ? d.bias[nt * 32 + li] : 0.0f;
    double d1[NT], d2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) d1[nt] = d2[nt] = 0.0;

    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int ntiles = d.N * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ox0 = (t % tiles_x) * TW, oy0 = ((t / tiles_x) % tiles_y) * TH, n = t / (tiles_x * tiles_y);
        __syncthreads();                                 // the previous tile's fragment reads are done
        // ---- stage the halo tile: float4 per pixel -> three bf16 planes of 8 bytes
        for (int idx = tid; idx < HP; idx += 256) {
            const int iy = oy0 - 1 + idx / HW, ix = ox0 - 1 + idx % HW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
                v = *reinterpret_cast<const f32x4*>(d.x + ((size_t)(n * d.H + iy) * d.W + ix) * 4);
            unsigned h0, m0, l0, h1, m1, l1;
            split_pair(v.x, v.y, h0, m0, l0);
            split_pair(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2*>(s_img + idx * 8) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(s_img + PLANE + idx * 8) = u32x2{m0, m1};
            *reinterpret_cast<u32x2*>(s_img + 2 * PLANE + idx * 8) = u32x2{l0, l1};
        }
        __syncthreads();
        // ---- wave w: rows 2 w and 2 w + 1 of the tile, one 32-pixel block each
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = 2 * wave + rb;
            const unsigned char* base = s_img + row * HW * 8;
            f32x16 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bf16x8 af[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const u32x2 a0 = *reinterpret_cast<const u32x2*>(base + p * PLANE + toff[s][0]);
                    const u32x2 a1 = *reinterpret_cast<const u32x2*>(base + p * PLANE + toff[s][1]);
                    af[p] = __builtin_bit_cast(bf16x8, u32x4{a0.x, a0.y, a1.x, a1.y});
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mma3(af, bfr[nt][s], acc[nt]);
            }
            // epilogue: C row = pixel (reg & 3) + 8 (reg >> 2) + 4 kh of the block, column = channel li
            const int oy = oy0 + row;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = nt * 32 + li;
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float v = acc[nt][r] + bias[nt];
                    if (oy < d.H && ox < d.W && co < d.Cout) {
                        d.y[((size_t)(n * d.H + oy) * d.W + ox) * d.y_ps + co] = v;
                        s1 += v;
                        s2 += v * v;
                    }
                }
                d1[nt] += (double)s1;
                d2[nt] += (double)s2;
            }
        }
    }
    if (d.stats) {                                       // one record per workgroup: channel sums over its lanes and waves
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s_stat[((wave * NT + nt) * 64 + lane) * 2] = d1[nt];
            s_stat[((wave * NT + nt) * 64 + lane) * 2 + 1] = d2[nt];
        }
        __syncthreads();
        if (tid < 32 * NT && tid < d.Cout) {
            const int nt = tid >> 5, cl = tid & 31;
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    t1 += s_stat[((w * NT + nt) * 64 + h * 32 + cl) * 2];
                    t2 += s_stat[((w * NT + nt) * 64 + h * 32 + cl) * 2 + 1];
                }
            d.stats[((size_t)blockIdx.x * d.Cout + tid) * 2] = t1;
            d.stats[((size_t)blockIdx.x * d.Cout + tid) * 2 + 1] = t2;
        }
    }
}

}  // namespace

bool conv_stem_eligible(const ConvArgs& a) {
    if (!(a.R == 3 && a.S == 1 && a.pad == 1 && a.zgroups == 1 && a.Cin == 4 && a.x.pstride == 4 && (a.Cout == 32 || a.Cout == 64))) return false;
    if (a.Hin != a.H || a.Win != a.W || a.Hout != a.H || a.Wout != a.W || a.osy != 1 || a.osx != 1 || a.ooy || a.oox) return false;
    if (a.xf.scale || a.fold || a.y16 || a.bwd_y) return false;
    if ((reinterpret_cast<uintptr_t>(a.x.p) | reinterpret_cast<uintptr_t>(a.w)) & 15) return false;
    return (int64_t)a.N * a.H * a.W * std::max<int64_t>(a.y.pstride, 4) * 4 < ((int64_t)1 << 40);
}

void launch_conv_stem(rfi_ctx* ctx, ConvArgs& a) {
    RFI_REQUIRE(conv_stem_eligible(a) && a.bf16x3, "conv_stem: shape or arithmetic not eligible");
    const int ntiles = a.N * (int)cdiv(a.H, TH) * (int)cdiv(a.W, TW);
    int GX = std::min(ntiles, 1024);                     // four workgroups per CU
    if (a.stats && GX > a.stats_max_records) GX = a.stats_max_records > 0 ? a.stats_max_records : GX;
    if (a.stats && a.stats_max_records > 0) a.stats_records = GX;
    else { a.stats = nullptr; a.stats_records = 0; }
    StemDev d{a.x.p, a.N, a.H, a.W, a.w, a.bias, a.y.p, (int)a.y.pstride, a.Cout, a.stats};
    const double flops = a.algo_flops >= 0 ? a.algo_flops : 2.0 * a.N * a.H * a.W * 9.0 * 3 * a.Cout;
    const double bytes = 4.0 * ((double)a.N * a.H * a.W * (4 + a.Cout) + 9.0 * 4 * a.Cout);
    std::string label;
    if (ctx->profiling)
        label = "conv_stem N" + std::to_string(a.N) + " " + std::to_string(a.H) + "x" + std::to_string(a.W) + " 4->" + std::to_string(a.Cout) +
                " 3xbf16";
    ProfScope ps(ctx, FAM_CONV_MFMA, flops, bytes, label);
    if (a.Cout == 32) hipLaunchKernelGGL(conv_stem_kernel<1>, dim3(GX), dim3(256), 0, ctx->stream, d);
    else hipLaunchKernelGGL(conv_stem_kernel<2>, dim3(GX), dim3(256), 0, ctx->stream, d);
    check_launch("conv_stem");
}

}  // namespace rfi
