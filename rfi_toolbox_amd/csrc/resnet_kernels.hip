// Elementwise pieces of the ResNet-style encoder (model_resnet.cpp; SURVEY 8a row A10 -- not in the reference,
// builder-defined, oracle/resnet_unet_ref.py).  All HBM-bound, NHWC float32, 16-byte accesses.
//
// Stride-2 3x3 convolutions run on the stride-1 kernels through space-to-depth: with
//   X'[n, y, x, (a, b, c)] = X[n, 2y + a, 2x + b, c]          (a, b in {0, 1}; 4C channels, half the resolution)
// a 3x3 / stride 2 / pad 1 conv of X is a 2x2 / stride 1 conv of X' with taps (dy, dx) in {-1, 0} (top/left pad 1):
// input row 2y + r - 1 is (dy, a) = (-1, 1), (0, 0), (0, 1) for r = 0, 1, 2, so 9 of the 16 (tap, a, b) filter
// blocks are the original taps and 7 are zero.  The 1x1 / stride 2 projection is a 1x1 conv on the (a, b) = (0, 0)
// channel slice of X' (a view, no copy).
#include "kernels.hpp"

namespace rfi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

int grid_of(int64_t total) {
    int64_t b = cdiv(total, 256);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// X [N][H][W][C] -> X' [N][H/2][W/2][4C]   (one float4 of channels per thread)
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ out) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const int64_t o = ((((int64_t)n * (H / 2) + (yy >> 1)) * (W / 2) + (xx >> 1)) * 4 + ((yy & 1) * 2 + (xx & 1))) * C + c;
        *reinterpret_cast<f32x4*>(out + o) = v;
    }
}
// inverse, with the shortcut gradient added on the (0, 0) slice:  dX[n, 2y+a, 2x+b, c] = dX'[n,y,x,(a,b,c)] (+ dS[n,y,x,c])
// and an optional full-resolution addend (the skip-connection gradient)
__global__ __launch_bounds__(256) void d2s_add_kernel(const float* __restrict__ dxp, const float* __restrict__ ds,
                                                     const float* __restrict__ extra, int extra_ps, int N, int H, int W, int C,
                                                     float* __restrict__ out) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        const int64_t pix = ((int64_t)n * (H / 2) + (yy >> 1)) * (W / 2) + (xx >> 1);
        f32x4 v = *reinterpret_cast<const f32x4*>(dxp + (pix * 4 + ((yy & 1) * 2 + (xx & 1))) * C + c);
        if (ds && !(yy & 1) && !(xx & 1)) v += *reinterpret_cast<const f32x4*>(ds + pix * C + c);
        if (extra) v += *reinterpret_cast<const f32x4*>(extra + (((int64_t)n * H + yy) * W + xx) * extra_ps + c);
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

// BasicBlock tail:  a = relu(Y * scale + shift + shortcut),  shortcut = S (identity) or S * s_scale + s_shift
// (projection: the raw 1x1 conv output with its own BatchNorm).  out may be a channel slice of a wider buffer.
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ s,
                                                         const float* __restrict__ s_scale, const float* __restrict__ s_shift,
                                                         int64_t M, int C, float* __restrict__ out, int out_ps,
                                                         float* __restrict__ out2, int out2_ps) {
    const int C4 = C / 4;
    const int64_t total = M * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const int64_t m = i / C4;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + i * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c), sh = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 sv = {0.0f, 0.0f, 0.0f, 0.0f};                     // s == null: plain BatchNorm + ReLU (the stem)
        if (s) sv = *reinterpret_cast<const f32x4*>(s + i * 4);
        if (s_scale) sv = sv * *reinterpret_cast<const f32x4*>(s_scale + c) + *reinterpret_cast<const f32x4*>(s_shift + c);
        f32x4 a = (yv * sc + sh) + sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = a[e] > 0.0f ? a[e] : 0.0f;
        *reinterpret_cast<f32x4*>(out + m * out_ps + c) = a;
        if (out2) *reinterpret_cast<f32x4*>(out2 + m * out2_ps + c) = a;
    }
}
// dz = (dA (+ dA2)) * (a > 0) (+ base): the gradient entering the block tail (both BatchNorm branches see dz as is);
// with `base` = the main branch's input gradient this is the whole input gradient of an identity-shortcut block
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ da, int da_ps, const float* __restrict__ da2,
                                                       int da2_ps, const float* __restrict__ a, int a_ps,
                                                       const float* __restrict__ base, int base_ps, int64_t M, int C,
                                                       float* __restrict__ dz) {
    const int C4 = C / 4;
    const int64_t total = M * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const int64_t m = i / C4;
        f32x4 g = *reinterpret_cast<const f32x4*>(da + m * da_ps + c);
        if (da2) g += *reinterpret_cast<const f32x4*>(da2 + m * da2_ps + c);
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + m * a_ps + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = av[e] > 0.0f ? g[e] : 0.0f;
        if (base) g += *reinterpret_cast<const f32x4*>(base + m * base_ps + c);
        *reinterpret_cast<f32x4*>(dz + i * 4) = g;
    }
}
__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy1_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ x, const float* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        v += *reinterpret_cast<const f32x4*>(y + i * 4);
        *reinterpret_cast<f32x4*>(x + i * 4) = v;
    }
}

// filters [9][Cout][Cin] (3x3, stride 2, pad 1) <-> [4][Cout][4 Cin] (2x2 on the space-to-depth input).
// tap (r, s) -> 2x2 tap t = (r == 0 ? 0 : 1) * 2 + (s == 0 ? 0 : 1), slice (a, b) = (r == 1 ? 0 : 1, s == 1 ? 0 : 1)
__global__ __launch_bounds__(256) void w_s2d_kernel(const float* __restrict__ w3, int Cout, int Cin, float* __restrict__ w2, int to_s2d) {
    const int64_t total = (int64_t)4 * Cout * 4 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        int64_t t = i / Cin;
        const int ab = (int)(t % 4); t /= 4;
        const int co = (int)(t % Cout);
        const int tap2 = (int)(t / Cout);
        const int dy = tap2 >> 1, dx = tap2 & 1, a = ab >> 1, b = ab & 1;
        // (dy, a): (0, 1) -> r = 0; (1, 0) -> r = 1; (1, 1) -> r = 2; (0, 0) -> no tap
        const int r = dy == 0 ? (a == 1 ? 0 : -1) : (a == 0 ? 1 : 2);
        const int s = dx == 0 ? (b == 1 ? 0 : -1) : (b == 0 ? 1 : 2);
        if (to_s2d) {
            w2[i] = (r >= 0 && s >= 0) ? w3[((int64_t)(r * 3 + s) * Cout + co) * Cin + c] : 0.0f;
        } else if (r >= 0 && s >= 0) {
            const_cast<float*>(w3)[((int64_t)(r * 3 + s) * Cout + co) * Cin + c] = w2[i];      // gradient back to the 3x3 layout
        }
    }
}

// MaxPool2d(3, stride 2, padding 1) of a = act(y * scale + shift) (scale == null: a = y), four channels per thread; the
// position of the FIRST maximum of each window (row-major, 0..8; as torch's CPU kernel: strict >) is kept for the
// backward pass, which gathers: an input pixel belongs to at most four windows
__global__ __launch_bounds__(256) void maxpool3_fwd_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out, unsigned* __restrict__ arg4) {
    const int C4 = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        int64_t t = i / C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) { sc = *reinterpret_cast<const f32x4*>(scale + c); sh = *reinterpret_cast<const f32x4*>(shift + c); }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned arg = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int iy = 2 * oy - 1 + k / 3, ix = 2 * ox - 1 + k % 3;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(y + (((int64_t)n * H + iy) * W + ix) * C + c);
            if (scale) {
                v = v * sc + sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > best[e]) { best[e] = v[e]; arg = (arg & ~(0xffu << (8 * e))) | ((unsigned)k << (8 * e)); }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = best;
        arg4[i] = arg;
    }
}
__global__ __launch_bounds__(256) void maxpool3_bwd_kernel(const float* __restrict__ dout, const unsigned* __restrict__ arg4, int N, int H,
                                                          int W, int C, float* __restrict__ da) {
    const int C4 = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int x = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        // windows (oy, ox) with 2 oy - 1 <= y <= 2 oy + 1
        for (int oy = yy / 2; oy <= (yy + 1) / 2; ++oy)
            for (int ox = x / 2; ox <= (x + 1) / 2; ++ox) {
                if (oy >= OH || ox >= OW) continue;
                const int k = (yy - (2 * oy - 1)) * 3 + (x - (2 * ox - 1));
                const int64_t o = (((int64_t)n * OH + oy) * OW + ox) * C4 + c4;
                const unsigned arg = arg4[o];
                const f32x4 d = *reinterpret_cast<const f32x4*>(dout + o * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((arg >> (8 * e)) & 0xffu) == (unsigned)k) g[e] += d[e];
            }
        *reinterpret_cast<f32x4*>(da + i * 4) = g;
    }
}
// x[:, ::2, ::2, :] (FPN's extra level) and its adjoint added into dx
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, int N, int H, int W, int C, float* __restrict__ out) {
    const int C4 = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(x + ((((int64_t)n * H + 2 * oy) * W + 2 * ox) * C4 + c4) * 4);
    }
}
__global__ __launch_bounds__(256) void subsample2_bwd_add_kernel(const float* __restrict__ dout, int N, int H, int W, int C,
                                                                float* __restrict__ dx) {
    const int C4 = C / 4, OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        float* q = dx + ((((int64_t)n * H + 2 * oy) * W + 2 * ox) * C4 + c4) * 4;
        *reinterpret_cast<f32x4*>(q) = *reinterpret_cast<const f32x4*>(q) + *reinterpret_cast<const f32x4*>(dout + i * 4);
    }
}

// K-packing of a small-channel stem conv: out[n, oy, ox, (r * R + s) * C + c] = x[n, oy * S + r - pad, ox * S + s - pad, c]
// (zero outside the image and for k >= R R C): the R x R conv becomes a 1x1 conv (a GEMM with K = Kp) on the matrix
// cores instead of a direct VALU kernel.  One thread per 4 consecutive k of one output pixel.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int N, int H, int W, int C, int R, int S, int pad,
                                                    int OH, int OW, int Kp, float* __restrict__ out) {
    const int K4 = Kp / 4, K = R * R * C;
    const int64_t total = (int64_t)N * OH * OW * K4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k0 = (int)(i % K4) * 4;
        int64_t t = i / K4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            if (k < K) {
                const int c = k % C, tap = k / C;
                const int iy = oy * S + tap / R - pad, ix = ox * S + tap % R - pad;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v[e] = x[(((int64_t)n * H + iy) * W + ix) * C + c];
            }
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}
// filters [R R][Cout][C] <-> [Cout][Kp] (k = tap * C + c; zero padding), the second direction for the gradient
__global__ __launch_bounds__(256) void w_pack_kernel(float* __restrict__ w, int taps, int Cout, int C, int Kp, float* __restrict__ wp,
                                                    int to_packed) {
    const int64_t total = (int64_t)Cout * Kp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp), co = (int)(i / Kp);
        const bool in = k < taps * C;
        const int64_t j = in ? ((int64_t)(k / C) * Cout + co) * C + (k % C) : 0;
        if (to_packed) wp[i] = in ? w[j] : 0.0f;
        else if (in) w[j] = wp[i];
    }
}

}  // namespace

void launch_im2col(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, int R, int S, int pad, int OH, int OW, int Kp, float* out) {
    RFI_REQUIRE(Kp % 4 == 0 && Kp >= R * R * C, "im2col: packed K must be a multiple of 4 and cover R R C");
    const int64_t total = (int64_t)N * OH * OW * Kp / 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 16 + (double)N * H * W * C * 4);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_of(total)), dim3(256), 0, ctx->stream, x, N, H, W, C, R, S, pad, OH, OW, Kp, out);
    check_launch("im2col");
}
void launch_w_pack(rfi_ctx* ctx, float* w, int taps, int Cout, int C, int Kp, float* wp, bool to_packed) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)Cout * Kp * 8);
    hipLaunchKernelGGL(w_pack_kernel, dim3(grid_of((int64_t)Cout * Kp)), dim3(256), 0, ctx->stream, w, taps, Cout, C, Kp, wp,
                       to_packed ? 1 : 0);
    check_launch("w_pack");
}

void launch_maxpool3_fwd(rfi_ctx* ctx, const float* y, int N, int H, int W, int C, const float* scale, const float* shift, float* out,
                         unsigned* arg4) {
    RFI_REQUIRE(C % 4 == 0, "maxpool3: C % 4 == 0");
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * 4 + (double)total * 20);
    hipLaunchKernelGGL(maxpool3_fwd_kernel, dim3(grid_of(total)), dim3(256), 0, ctx->stream, y, N, H, W, C, scale, shift, out, arg4);
    check_launch("maxpool3_fwd");
}
void launch_maxpool3_bwd(rfi_ctx* ctx, const float* dout, const unsigned* arg4, int N, int H, int W, int C, float* da) {
    RFI_REQUIRE(C % 4 == 0, "maxpool3: C % 4 == 0");
    const int64_t total = (int64_t)N * H * W * C / 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 16 * 2);
    hipLaunchKernelGGL(maxpool3_bwd_kernel, dim3(grid_of(total)), dim3(256), 0, ctx->stream, dout, arg4, N, H, W, C, da);
    check_launch("maxpool3_bwd");
}
void launch_subsample2(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, float* out) {
    RFI_REQUIRE(C % 4 == 0, "subsample2: C % 4 == 0");
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 32);
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_of(total)), dim3(256), 0, ctx->stream, x, N, H, W, C, out);
    check_launch("subsample2");
}
void launch_subsample2_bwd_add(rfi_ctx* ctx, const float* dout, int N, int H, int W, int C, float* dx) {
    RFI_REQUIRE(C % 4 == 0, "subsample2: C % 4 == 0");
    const int64_t total = (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) * C / 4;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 48);
    hipLaunchKernelGGL(subsample2_bwd_add_kernel, dim3(grid_of(total)), dim3(256), 0, ctx->stream, dout, N, H, W, C, dx);
    check_launch("subsample2_bwd_add");
}

void launch_s2d(rfi_ctx* ctx, const float* x, int N, int H, int W, int C, float* out) {
    RFI_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "space-to-depth: C % 4 == 0, even H and W");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * 8);
    hipLaunchKernelGGL(s2d_kernel, dim3(grid_of((int64_t)N * H * W * C / 4)), dim3(256), 0, ctx->stream, x, N, H, W, C, out);
    check_launch("s2d");
}
void launch_d2s_add(rfi_ctx* ctx, const float* dxp, const float* ds, View extra, int N, int H, int W, int C, float* out) {
    RFI_REQUIRE(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "depth-to-space: C % 4 == 0, even H and W");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * 12);
    hipLaunchKernelGGL(d2s_add_kernel, dim3(grid_of((int64_t)N * H * W * C / 4)), dim3(256), 0, ctx->stream, dxp, ds, extra.p,
                       extra.pstride, N, H, W, C, out);
    check_launch("d2s_add");
}
void launch_bn_add_relu(rfi_ctx* ctx, const float* y, const float* scale, const float* shift, const float* s,
                        const float* s_scale, const float* s_shift, int64_t M, int C, MutView out, MutView out2) {
    RFI_REQUIRE(C % 4 == 0, "bn_add_relu: C % 4 == 0");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * C * (out2.p ? 16 : 12));
    hipLaunchKernelGGL(bn_add_relu_kernel, dim3(grid_of(M * C / 4)), dim3(256), 0, ctx->stream, y, scale, shift, s, s_scale,
                       s_shift, M, C, out.p, out.pstride, out2.p, out2.pstride);
    check_launch("bn_add_relu");
}
void launch_relu_mask(rfi_ctx* ctx, View da, View da2, View a, View base, int64_t M, int C, float* dz) {
    RFI_REQUIRE(C % 4 == 0, "relu_mask: C % 4 == 0");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * C * (12 + (da2.p ? 4 : 0) + (base.p ? 4 : 0)));
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid_of(M * C / 4)), dim3(256), 0, ctx->stream, da.p, da.pstride, da2.p, da2.pstride,
                       a.p, a.pstride, base.p, base.pstride, M, C, dz);
    check_launch("relu_mask");
}
// A device-to-device copy as an ordinary kernel on the context's stream: hipMemcpyAsync(DeviceToDevice) goes through the
// runtime's blit path, which left the queue idle for ~55 us in front of every copy (rocprofv3 trace of the detector's step:
// 50 copies, 2.8 of 28 ms); a kernel follows its predecessor like any other launch
void launch_copy_d2d(rfi_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, 2.0 * bytes);
    const bool v16 = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15) == 0;
    if (v16) {
        const int64_t n = (int64_t)(bytes / 16);
        hipLaunchKernelGGL(copy16_kernel, dim3(grid_of(n)), dim3(256), 0, ctx->stream, static_cast<uint4*>(dst), static_cast<const uint4*>(src), n);
    } else {
        const int64_t n = (int64_t)bytes;
        hipLaunchKernelGGL(copy1_kernel, dim3(grid_of(n)), dim3(256), 0, ctx->stream, static_cast<unsigned char*>(dst),
                           static_cast<const unsigned char*>(src), n);
    }
    check_launch("copy_d2d");
}
void launch_add_inplace(rfi_ctx* ctx, float* x, const float* y, int64_t n) {
    RFI_REQUIRE(n % 4 == 0, "add_inplace: n % 4 == 0");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 12);
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_of(n / 4)), dim3(256), 0, ctx->stream, x, y, n / 4);
    check_launch("add_inplace");
}
void launch_w_s2d(rfi_ctx* ctx, float* w3, int Cout, int Cin, float* w2, bool to_s2d) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)25 * Cout * Cin * 4);
    hipLaunchKernelGGL(w_s2d_kernel, dim3(grid_of((int64_t)16 * Cout * Cin)), dim3(256), 0, ctx->stream, w3, Cout, Cin, w2,
                       to_s2d ? 1 : 0);
    check_launch("w_s2d");
}

}  // namespace rfi
