// Producers of the plane format (planes.hpp): activation split (with the producing layer's BatchNorm-apply +
// activation folded in), filter re-layout into MFMA B-operand order, and the inverse (planes -> float32).
// All HBM-bound elementwise kernels: 16-byte loads and stores, grid-stride.
#include "planes.hpp"

namespace rfi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pair(float a, float b) {       // RNE, v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// v = h + m + l exactly (three RNE bf16 pieces); the same sequence the round-1 kernels ran at staging time
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pair(a, b);
    const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pair(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m << 16), sb = rb - __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pair(sa, sb);
}
// 8 floats -> P pieces of 8 bf16 (16 bytes each)
template <int P>
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&out)[3]) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (P == 3) split_pair(v[2 * i], v[2 * i + 1], h[i], m[i], l[i]);
        else h[i] = cvt_pair(v[2 * i], v[2 * i + 1]);
    }
    out[0] = u32x4{h[0], h[1], h[2], h[3]};
    if constexpr (P == 3) {
        out[1] = u32x4{m[0], m[1], m[2], m[3]};
        out[2] = u32x4{l[0], l[1], l[2], l[3]};
    }
}

template <int P, bool VEC>
__global__ __launch_bounds__(256) void act_split_kernel(const float* __restrict__ x, const bf16_t* __restrict__ x16, int xp, int64_t M, int C,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int relu, float slope, bf16_t* __restrict__ out, int64_t op) {
    const int groups = plane_chunks(C) * 2;                    // 8-channel groups per pixel (padded)
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        const int64_t m = i / groups;
        const int c0 = g * 8;
        float v[8];
        const float* src = x + m * xp + c0;
        if (x16) {                                  // the raw conv output stored as bfloat16 (bf16 data flow): one 16-byte load
            const u32x4 t = *reinterpret_cast<const u32x4*>(x16 + m * xp + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __builtin_bit_cast(float, t[e] << 16);
                v[2 * e + 1] = __builtin_bit_cast(float, t[e] & 0xffff0000u);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? v[e] : 0.0f;
        } else if (VEC && c0 + 8 <= C) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? src[e] : 0.0f;
        }
        if (scale) {
            float sc[8], sh[8];
            if (VEC && c0 + 8 <= C) {          // four 16-byte loads instead of sixteen dword gathers
                const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0), b = *reinterpret_cast<const f32x4*>(scale + c0 + 4);
                const f32x4 c = *reinterpret_cast<const f32x4*>(shift + c0), d = *reinterpret_cast<const f32x4*>(shift + c0 + 4);
                sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
                sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w; sh[4] = d.x; sh[5] = d.y; sh[6] = d.z; sh[7] = d.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sc[e] = c0 + e < C ? scale[c0 + e] : 0.0f;
                    sh[e] = c0 + e < C ? shift[c0 + e] : 0.0f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c0 + e < C) {
                    float t = v[e] * sc[e] + sh[e];                          // product and sum rounded separately (-ffp-contract=off)
                    if (relu) t = fmaxf(t, t * slope);                       // slope 0: ReLU
                    v[e] = t;
                }
            }
        }
        u32x4 pc[3];
        split8<P>(v, pc);
        bf16_t* o = out + m * op + (int64_t)(g >> 1) * (P * 16) + (g & 1) * 8;
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(o + p * 16) = pc[p];
    }
}

// a = act(y * scale + shift) -> skip planes (full resolution) and 2x2 max-pooled planes, 8 channels of one pooled
// pixel per thread (Encoder.forward, models/unet.py:21-28: `pool(conv(x)), conv(x)`)
template <int P>
__global__ __launch_bounds__(256) void bn_relu_pool_planes_kernel(const float* __restrict__ y, const bf16_t* __restrict__ y16,
                                                                 int64_t yps, int N, int H, int W, int C,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float slope, bf16_t* __restrict__ skip, int64_t sp,
                                                                 bf16_t* __restrict__ pooled, int64_t pp) {
    const int Hp = H >> 1, Wp = W >> 1, groups = plane_chunks(C) * 2;
    const int64_t total = (int64_t)N * Hp * Wp * groups;
    const bool vec = (C & 3) == 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % groups);
        int64_t t = i / groups;
        const int px = (int)(t % Wp); t /= Wp;
        const int py = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const int c0 = g * 8;
        float sc[8], sh[8], best[8];
        if (vec && c0 + 8 <= C) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(scale + c0), b = *reinterpret_cast<const f32x4*>(scale + c0 + 4);
            const f32x4 c = *reinterpret_cast<const f32x4*>(shift + c0), d = *reinterpret_cast<const f32x4*>(shift + c0 + 4);
            sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
            sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w; sh[4] = d.x; sh[5] = d.y; sh[6] = d.z; sh[7] = d.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc[e] = c0 + e < C ? scale[c0 + e] : 0.0f;
                sh[e] = c0 + e < C ? shift[c0 + e] : 0.0f;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t pix = ((int64_t)n * H + (2 * py + (k >> 1))) * W + (2 * px + (k & 1));
            const float* src = y + pix * C + c0;
            float v[8];
            if (y16) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(y16 + pix * yps + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = __builtin_bit_cast(float, t[e] << 16);
                    v[2 * e + 1] = __builtin_bit_cast(float, t[e] & 0xffff0000u);
                }
            } else if (vec && c0 + 8 <= C) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (c0 + e < C) ? src[e] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = v[e] * sc[e] + sh[e];
                a = a > 0.0f ? a : (slope != 0.0f ? a * slope : 0.0f);         // torch's (Leaky)ReLU
                v[e] = a;
                best[e] = (k == 0 || a > best[e]) ? a : best[e];
            }
            u32x4 pc[3];
            split8<P>(v, pc);
            bf16_t* o = skip + pix * sp + (int64_t)(g >> 1) * (P * 16) + (g & 1) * 8;
#pragma unroll
            for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(o + p * 16) = pc[p];
        }
        u32x4 pc[3];
        split8<P>(best, pc);
        bf16_t* o = pooled + (((int64_t)n * Hp + py) * Wp + px) * pp + (int64_t)(g >> 1) * (P * 16) + (g & 1) * 8;
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(o + p * 16) = pc[p];
    }
}


// ---- ResNet-style encoder on the bf16 data flow: bfloat16 [pixel][>= C] tensors with 16-byte aligned rows, C % 8 == 0
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[2 * e] = __builtin_bit_cast(float, t[e] << 16);
        v[2 * e + 1] = __builtin_bit_cast(float, t[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<u32x4*>(p) = u32x4{cvt_pair(v[0], v[1]), cvt_pair(v[2], v[3]), cvt_pair(v[4], v[5]), cvt_pair(v[6], v[7])};
}
// out = relu(y * sc + sh + shortcut), shortcut = s * ssc + ssh (a projection's raw output) or s itself (ssc == null)
__global__ __launch_bounds__(256) void bn_add_relu16_kernel(const bf16_t* __restrict__ y, int64_t yp, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, const bf16_t* __restrict__ s, int64_t sp,
                                                           const float* __restrict__ ssc, const float* __restrict__ ssh, int64_t M, int C,
                                                           bf16_t* __restrict__ out, int64_t op) {
    const int groups = C / 8;
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % groups) * 8;
        const int64_t m = i / groups;
        float v[8], u[8], a[8], b[8];
        load8(y + m * yp + c0, v);
        load8(s + m * sp + c0, u);
        load8(sc + c0, a);
        load8(sh + c0, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * a[e] + b[e];
        if (ssc) {
            load8(ssc + c0, a);
            load8(ssh + c0, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = u[e] * a[e] + b[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + u[e], 0.0f);
        store8(out + m * op + c0, v);
    }
}
// dz = (g0 + g1 + g2) * (a > 0)   (g1, g2, a optional; g2 a float32 view)
__global__ __launch_bounds__(256) void relu_mask_sum16_kernel(const bf16_t* __restrict__ g0, int64_t p0, const bf16_t* __restrict__ g1, int64_t p1,
                                                             const void* __restrict__ g2, int g2_16, int64_t p2, const bf16_t* __restrict__ a,
                                                             int64_t pa, int64_t M, int C, bf16_t* __restrict__ dz, int64_t pz) {
    const int groups = C / 8;
    const int64_t total = M * groups;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % groups) * 8;
        const int64_t m = i / groups;
        float v[8], u[8];
        load8(g0 + m * p0 + c0, v);
        if (g1) {
            load8(g1 + m * p1 + c0, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += u[e];
        }
        if (g2) {
            if (g2_16) load8(static_cast<const bf16_t*>(g2) + m * p2 + c0, u);
            else load8(static_cast<const float*>(g2) + m * p2 + c0, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += u[e];
        }
        if (a) {
            load8(a + m * pa + c0, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = u[e] > 0.0f ? v[e] : 0.0f;
        }
        store8(dz + m * pz + c0, v);
    }
}

template <int P>
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const bf16_t* __restrict__ in, int64_t ip, int64_t M, int C,
                                                            float* __restrict__ out, int op) {
    const int64_t total = M * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t m = i / C;
        const bf16_t* q = in + m * ip + (int64_t)(c >> 4) * (P * 16) + (c & 15);
        float s = 0.0f;
#pragma unroll
        for (int p = P - 1; p >= 0; --p) s += __builtin_bit_cast(float, (unsigned)q[p * 16] << 16);   // small pieces first
        out[m * op + c] = s;
    }
}


__global__ __launch_bounds__(256) void w_s2_classes_kernel(const float* __restrict__ w3, const float* __restrict__ wp, int Cout, int Cin,
                                                          float* __restrict__ dst) {
    const int64_t total = (int64_t)20 * Cin * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c0 = (int64_t)8 * Cin * Cout, per = (int64_t)4 * Cin * Cout;
        const int c = i < c0 ? 0 : 1 + (int)((i - c0) / per);
        const int64_t j = c == 0 ? i : (i - c0) % per;
        const int K = c == 0 ? 2 * Cout : Cout;
        const int k = (int)(j % K);
        const int ci = (int)((j / K) % Cin);
        const int t = (int)(j / ((int64_t)K * Cin));
        const int py = c >> 1, px = c & 1, ty = t >> 1, tx = t & 1;
        const int r = py ? (ty ? 0 : 2) : (ty ? -1 : 1), s = px ? (tx ? 0 : 2) : (tx ? -1 : 1);
        float v = 0.0f;
        if (k < Cout) { if (w3 && r >= 0 && s >= 0) v = w3[((int64_t)(r * 3 + s) * Cout + k) * Cin + ci]; }
        else if (t == 0 && wp) v = wp[(int64_t)(k - Cout) * Cin + ci];
        dst[i] = v;
    }
}

template <int P>
__device__ __forceinline__ void wb_body(const WBDesc& d) {
    const int kc0 = (d.seg_c[0] + 15) / 16, kc1 = d.seg_c[1] ? (d.seg_c[1] + 15) / 16 : 0;
    const int nkc = kc0 + kc1, ncb = (d.Cout + 31) / 32;
    const int64_t total = (int64_t)d.taps * nkc * ncb * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int cb = (int)(r % ncb); r /= ncb;
        const int kc = (int)(r % nkc);
        const int tap = (int)(r / nkc);
        const int j = lane & 31, lh = lane >> 5;
        const int co = cb * 32 + j;
        const int seg = kc >= kc0 ? 1 : 0;
        const int cl = (seg ? kc - kc0 : kc) * 16 + lh * 8;         // channel within the segment
        const int cbase = seg ? d.seg_c[0] : 0;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (co < d.Cout && cl + e < d.seg_c[seg]) ? d.src[((int64_t)tap * d.Cout + co) * d.Cin + cbase + cl + e] : 0.0f;
        u32x4 pc[3];
        split8<P>(v, pc);
        bf16_t* o = d.dst + ((((int64_t)tap * nkc + kc) * ncb + cb) * P) * 512 + lane * 8;
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<u32x4*>(o + p * 512) = pc[p];
    }
}
__global__ __launch_bounds__(256) void weights_to_wb_kernel(const WBDesc* __restrict__ descs) {
    const WBDesc d = descs[blockIdx.y];
    if (d.P == 3) wb_body<3>(d);
    else wb_body<1>(d);
}

}  // namespace

void launch_act_split(rfi_ctx* ctx, View x, int64_t M, int C, InXform xf, int P, bf16_t* out, int64_t out_pstride,
                      const bf16_t* x16, int64_t x16_pstride) {
    RFI_REQUIRE(P == 1 || P == 3, "act_split: planes must be 1 or 3");
    RFI_REQUIRE(M > 0 && C > 0, "act_split: empty tensor");
    RFI_REQUIRE(out_pstride % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "act_split: unaligned output");
    const bool vec = (x.pstride % 4 == 0) && (reinterpret_cast<uintptr_t>(x.p) & 15) == 0;
    RFI_REQUIRE(!x16 || (x16_pstride % 8 == 0 && (reinterpret_cast<uintptr_t>(x16) & 15) == 0 && x16_pstride >= 16 * plane_chunks(C)),
                "act_split: the bfloat16 input must be a chunk-padded, 16-byte aligned tensor");
    const int64_t total = M * plane_chunks(C) * 2;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * C * (x16 ? 2 : 4) + (double)total * 16 * P);
    const int xp = x16 ? (int)x16_pstride : x.pstride;
    int64_t blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    const int relu = xf.scale ? xf.relu : 0;
#define RFI_AS(P_, V_)                                                                                        \
    hipLaunchKernelGGL((act_split_kernel<P_, V_>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x.p,  \
                       x16, xp, M, C, xf.scale, xf.shift, relu, xf.slope, out, out_pstride)
    if (P == 3) { if (vec) RFI_AS(3, true); else RFI_AS(3, false); }
    else { if (vec) RFI_AS(1, true); else RFI_AS(1, false); }
#undef RFI_AS
    check_launch("act_split");
}

void launch_bn_relu_pool_planes(rfi_ctx* ctx, const float* y, int N, int H, int W, int C, const float* scale,
                                const float* shift, float slope, int P, bf16_t* skip, int64_t skip_pstride,
                                bf16_t* pooled, int64_t pooled_pstride, const bf16_t* y16, int64_t y16_pstride) {
    RFI_REQUIRE((H & 1) == 0 && (W & 1) == 0, "bn_relu_pool_planes: even H and W (the U-Net levels are)");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * plane_chunks(C) * 2;
    RFI_REQUIRE(!y16 || (y16_pstride % 8 == 0 && (reinterpret_cast<uintptr_t>(y16) & 15) == 0 && y16_pstride >= 16 * plane_chunks(C)),
                "bn_relu_pool_planes: the bfloat16 input must be a chunk-padded, 16-byte aligned tensor");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * (y16 ? 2 : 4) + (double)total * 16 * P * 5);
    int64_t blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    if (P == 3)
        hipLaunchKernelGGL(bn_relu_pool_planes_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, y16, y16_pstride,
                           N, H, W, C, scale, shift, slope, skip, skip_pstride, pooled, pooled_pstride);
    else
        hipLaunchKernelGGL(bn_relu_pool_planes_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, y16, y16_pstride,
                           N, H, W, C, scale, shift, slope, skip, skip_pstride, pooled, pooled_pstride);
    check_launch("bn_relu_pool_planes");
}


static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void launch_bn_add_relu16(rfi_ctx* ctx, const bf16_t* y, int64_t y_ps, const float* scale, const float* shift, const bf16_t* s, int64_t s_ps,
                          const float* s_scale, const float* s_shift, int64_t M, int C, bf16_t* out, int64_t out_ps) {
    RFI_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && y_ps % 8 == 0 && s_ps % 8 == 0 && out_ps % 8 == 0 && al16(y) && al16(s) && al16(out) &&
                    al16(scale) && al16(shift) && (!s_scale || (al16(s_scale) && al16(s_shift))),
                "bn_add_relu16: channels in whole groups of 8, 16-byte aligned rows");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * C * 6);
    int64_t blocks = cdiv(M * (C / 8), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_add_relu16_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, y_ps, scale, shift, s, s_ps, s_scale, s_shift,
                       M, C, out, out_ps);
    check_launch("bn_add_relu16");
}

void launch_relu_mask_sum16(rfi_ctx* ctx, const bf16_t* g0, int64_t p0, const bf16_t* g1, int64_t p1, YRef g2, const bf16_t* a, int64_t pa,
                            int64_t M, int C, bf16_t* dz, int64_t pz) {
    RFI_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && p0 % 8 == 0 && (!g1 || p1 % 8 == 0) && (!g2.p || g2.stride(C) % (g2.bf16 ? 8 : 4) == 0) && (!a || pa % 8 == 0) &&
                    pz % 8 == 0 && al16(g0) && al16(g1) && al16(g2.p) && al16(a) && al16(dz),
                "relu_mask_sum16: channels in whole groups of 8, 16-byte aligned rows");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * C * (4 + (g1 ? 2 : 0) + (g2.p ? (g2.bf16 ? 2 : 4) : 0) + (a ? 2 : 0)));
    int64_t blocks = cdiv(M * (C / 8), 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_mask_sum16_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, g0, p0, g1, p1, g2.p, g2.bf16, g2.stride(C), a, pa,
                       M, C, dz, pz);
    check_launch("relu_mask_sum16");
}

void launch_planes_to_f32(rfi_ctx* ctx, const bf16_t* in, int64_t in_pstride, int64_t M, int C, int P, float* out,
                          int out_pstride) {
    const int64_t total = M * C;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * (4 + 2 * P));
    int64_t blocks = cdiv(total, 256);
    if (blocks > 8192) blocks = 8192;
    if (P == 3) hipLaunchKernelGGL(planes_to_f32_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in, in_pstride, M, C, out, out_pstride);
    else hipLaunchKernelGGL(planes_to_f32_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in, in_pstride, M, C, out, out_pstride);
    check_launch("planes_to_f32");
}


void launch_w_s2_classes(rfi_ctx* ctx, const float* w3, const float* wp, int Cout, int Cin, float* dst) {
    const int64_t total = (int64_t)20 * Cin * Cout;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 4 + 40.0 * Cin * Cout);
    int64_t blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(w_s2_classes_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, w3, wp, Cout, Cin, dst);
    check_launch("w_s2_classes");
}

void launch_weights_to_wb(rfi_ctx* ctx, const WBDesc* descs_dev, int n, double total_bytes) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, total_bytes);
    hipLaunchKernelGGL(weights_to_wb_kernel, dim3(512, n), dim3(256), 0, ctx->stream, descs_dev);    // (grid-stride loops: small layers leave most of the row idle)
    check_launch("weights_to_wb");
}
void launch_weights_to_wb_one(rfi_ctx* ctx, const WBDesc& d) {
    WBDesc* dev = static_cast<WBDesc*>(ctx->alloc(sizeof(WBDesc)));
    RFI_CHECK_HIP(hipMemcpyAsync(dev, &d, sizeof(WBDesc), hipMemcpyHostToDevice, ctx->stream));
    launch_weights_to_wb(ctx, dev, 1, 0);
    RFI_CHECK_HIP(hipStreamSynchronize(ctx->stream));         // &d is the caller's stack; dev is freed here
    ctx->release(dev);
}

}  // namespace rfi
