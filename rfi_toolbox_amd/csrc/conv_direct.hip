// VALU direct forms of the conv-like contraction and of the weight gradient.  They serve the
// shapes the MFMA implicit-GEMM kernels do not take (the 3-channel stem, channel counts that are
// not a multiple of 4) and accept exactly the same argument structs.
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kBlock = 256;

// thread = (output pixel, cout); cout fastest so that stores and weight reads coalesce
__global__ void conv_direct_kernel(ConvArgs a) {
    const int z = blockIdx.y;
    const float* __restrict__ w = a.w + (size_t)z * a.Cout * a.Cin * (a.zgroups > 1 ? 1 : 0);
    const int ooy = a.zgroups > 1 ? (z >> 1) : a.ooy;
    const int oox = a.zgroups > 1 ? (z & 1) : a.oox;
    const int64_t total = (int64_t)a.N * a.H * a.W * a.Cout;
    const int taps = a.R * a.R;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % a.Cout);
        int64_t t = i / a.Cout;
        const int ox = (int)(t % a.W);
        t /= a.W;
        const int oy = (int)(t % a.H);
        const int n = (int)(t / a.H);
        float acc = a.bias ? a.bias[co] : 0.0f;
        for (int tap = 0; tap < taps; ++tap) {
            const int iy = oy * a.S + tap / a.R - a.pad;
            const int ix = ox * a.S + tap % a.R - a.pad;
            if (iy < 0 || iy >= a.Hin || ix < 0 || ix >= a.Win) continue;
            const float* __restrict__ xp = a.x.p + (((int64_t)n * a.Hin + iy) * a.Win + ix) * a.x.pstride;
            const float* __restrict__ wp = w + ((size_t)tap * a.Cout + co) * a.Cin;
            if (a.xf.scale) {
                for (int ci = 0; ci < a.Cin; ++ci) {
                    float v = xp[ci] * a.xf.scale[ci] + a.xf.shift[ci];
                    if (a.xf.relu) v = v > 0.0f ? v : v * a.xf.slope;
                    acc = fmaf(v, wp[ci], acc);
                }
            } else {
                for (int ci = 0; ci < a.Cin; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
            }
        }
        const int64_t opix = ((int64_t)n * a.Hout + (oy * a.osy + ooy)) * a.Wout + (ox * a.osx + oox);
        a.y.p[opix * a.y.pstride + co] = acc;
    }
}

// thread = one dW element, ordered (tap, cx, cy) with cy fastest; grid.y = pixel split.
// slab[split][tap][cx][cy] partial sums, reduced afterwards.
__global__ void wgrad_direct_kernel(WgradArgs a, int64_t pix_per_split, int64_t nout) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    const int cy = (int)(o % a.Cy);
    int64_t t = o / a.Cy;
    const int cx = (int)(t % a.Cx);
    const int tap = (int)(t / a.Cx);
    const int r = tap / a.R, s = tap % a.R;
    const int64_t M = (int64_t)a.N * a.H * a.W;
    const int64_t m0 = (int64_t)blockIdx.y * pix_per_split;
    int64_t m1 = m0 + pix_per_split;
    if (m1 > M) m1 = M;
    float sxs = 1.0f, sxh = 0.0f, sys = 1.0f, syh = 0.0f;
    if (a.xf_x.scale) { sxs = a.xf_x.scale[cx]; sxh = a.xf_x.shift[cx]; }
    if (a.xf_y.scale) { sys = a.xf_y.scale[cy]; syh = a.xf_y.shift[cy]; }
    float acc = 0.0f;
    int x = (int)(m0 % a.W);
    int y = (int)((m0 / a.W) % a.H);
    int n = (int)(m0 / ((int64_t)a.W * a.H));
    for (int64_t m = m0; m < m1; ++m) {
        const int iy = y * a.S + r - a.pad, ix = x * a.S + s - a.pad;
        if (iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx) {
            float xv = a.xop.p[(((int64_t)n * a.Hx + iy) * a.Wx + ix) * a.xop.pstride + cx];
            if (a.xf_x.scale) {
                xv = xv * sxs + sxh;
                if (a.xf_x.relu) xv = xv > 0.0f ? xv : xv * a.xf_x.slope;
            }
            float yv = a.yop.p[m * a.yop.pstride + cy];
            if (a.xf_y.scale) {
                yv = yv * sys + syh;
                if (a.xf_y.relu) yv = yv > 0.0f ? yv : yv * a.xf_y.slope;
            }
            acc = fmaf(xv, yv, acc);
        }
        if (++x == a.W) {
            x = 0;
            if (++y == a.H) { y = 0; ++n; }
        }
    }
    a.slab[(int64_t)blockIdx.y * nout + o] = acc;
}

// dw[tap*tap_stride + cy*sy + cx*sx] = sum_split slab[split][tap][cx][cy]
__global__ void wgrad_direct_finish_kernel(const float* __restrict__ slab, int nsplit, int64_t nout,
                                           int Cx, int Cy, int64_t tap_stride, int sy, int sx,
                                           float* __restrict__ dw) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nout;
         o += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < nsplit; ++k) s += slab[(int64_t)k * nout + o];
        const int cy = (int)(o % Cy);
        int64_t t = o / Cy;
        const int cx = (int)(t % Cx);
        const int tap = (int)(t / Cx);
        dw[(int64_t)tap * tap_stride + (int64_t)cy * sy + (int64_t)cx * sx] = s;
    }
}

}  // namespace

void launch_conv_direct(rfi_ctx* ctx, const ConvArgs& a) {
    const int64_t total = (int64_t)a.N * a.H * a.W * a.Cout;
    int64_t blocks = cdiv(total, kBlock);
    if (blocks > 65536) blocks = 65536;
    const double flops = 2.0 * total * a.R * a.R * a.Cin * a.zgroups;
    ProfScope ps(ctx, FAM_CONV_DIRECT, flops, 0);
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)blocks, a.zgroups), dim3(kBlock), 0,
                       ctx->stream, a);
    check_launch("conv_direct");
}

static int wgrad_direct_splits(const WgradArgs& a) {
    const int64_t M = (int64_t)a.N * a.H * a.W;
    const int64_t nout = (int64_t)a.R * a.R * a.Cx * a.Cy;
    // aim at ~4096 blocks in total, at least 256 pixels per split
    int64_t oblocks = cdiv(nout, kBlock);
    int64_t want = cdiv(4096, oblocks);
    int64_t maxs = cdiv(M, 256);
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    return (int)want;
}

size_t wgrad_direct_slab_floats(const WgradArgs& a) {
    return (size_t)wgrad_direct_splits(a) * a.R * a.R * a.Cx * a.Cy;
}

void launch_wgrad_direct(rfi_ctx* ctx, const WgradArgs& a) {
    const int64_t M = (int64_t)a.N * a.H * a.W;
    const int64_t nout = (int64_t)a.R * a.R * a.Cx * a.Cy;
    const int nsplit = wgrad_direct_splits(a);
    RFI_REQUIRE(a.slab && a.slab_floats >= (size_t)nsplit * nout, "wgrad_direct: slab workspace too small");
    const int64_t pps = cdiv(M, nsplit);
    {
        ProfScope ps(ctx, FAM_CONV_DIRECT, 2.0 * M * nout, 0);
        hipLaunchKernelGGL(wgrad_direct_kernel, dim3((unsigned)cdiv(nout, kBlock), nsplit), dim3(kBlock),
                           0, ctx->stream, a, pps, nout);
        check_launch("wgrad_direct");
    }
    {
        ProfScope ps(ctx, FAM_REDUCE);
        int64_t blocks = cdiv(nout, kBlock);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(wgrad_direct_finish_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                           ctx->stream, a.slab, nsplit, nout, a.Cx, a.Cy, a.tap_stride, a.sy, a.sx, a.dw);
        check_launch("wgrad_direct_finish");
    }
}

}  // namespace rfi
