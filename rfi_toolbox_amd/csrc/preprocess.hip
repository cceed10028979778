// On-device form of the per-patch hot loop of Preprocessor.create_dataset
// (rfi_toolbox/preprocessing/preprocessor.py:366-384): complex/real patch -> 3 channels
// [gradient magnitude of log-amplitude (per-patch min-max), log-amplitude, phase] -> float32 ->
// ImageNet normalisation, written NHWC.  Math in fp64 for 64-bit inputs exactly as NumPy does it,
// rounded to fp32 where the reference casts (.astype(np.float32), :376) and normalised in fp32 (:783).
// Also the confusion counts of evaluation/metrics.py and the sigmoid threshold of evaluate_model.py.
// HBM-bound: 16 B in + 12 B out per pixel for complex128.
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kBlock = 256;

template <typename T>
__device__ __forceinline__ T load_amp(const void* p, int dtype, int64_t i, T* phase) {
    // returns |z| (or |x| for real input) and the phase in *phase
    if (dtype == RFI_C128) {
        const double re = reinterpret_cast<const double*>(p)[2 * i];
        const double im = reinterpret_cast<const double*>(p)[2 * i + 1];
        *phase = (T)atan2(im, re);
        return (T)hypot(re, im);
    } else if (dtype == RFI_C64) {
        const float re = reinterpret_cast<const float*>(p)[2 * i];
        const float im = reinterpret_cast<const float*>(p)[2 * i + 1];
        *phase = (T)atan2f(im, re);
        return (T)hypotf(re, im);
    } else if (dtype == RFI_F64) {
        *phase = (T)0;
        return (T)fabs(reinterpret_cast<const double*>(p)[i]);
    } else {
        *phase = (T)0;
        return (T)fabsf(reinterpret_cast<const float*>(p)[i]);
    }
}

__device__ __forceinline__ double logamp_d(double amp) { return log10(amp + 1e-10); }
__device__ __forceinline__ float logamp_f(float amp) { return log10f(amp + 1e-10f); }

// ordered-int encoding so float min/max can use integer atomics (handles negatives)
__device__ __forceinline__ unsigned enc(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// pass 1: per patch min/max of channel 0 (gradient) and of log-amp (real input's channel 1)
// mm[patch*4 + {0,1,2,3}] = enc(min grad), enc(max grad), enc(min la), enc(max la)
template <bool WIDE>
__global__ void prep_minmax_kernel(const void* __restrict__ src, int dtype, int ph, int pw,
                                   unsigned* __restrict__ mm) {
    const int patch = blockIdx.y;
    const int64_t base = (int64_t)patch * ph * pw;
    float gmin = INFINITY, gmax = -INFINITY, lmin = INFINITY, lmax = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ph * pw; i += gridDim.x * blockDim.x) {
        const int r = i / pw, c = i % pw;
        float g, laf;
        if (WIDE) {
            double phs;
            const double la = logamp_d(load_amp<double>(src, dtype, base + i, &phs));
            const double d0 = r > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - pw, &phs)) : 0.0;
            const double d1 = c > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - 1, &phs)) : 0.0;
            g = (float)sqrt(d0 * d0 + d1 * d1);
            // min-max is taken in the source precision by the reference; order is preserved by
            // the monotone rounding to float, the exact double extrema are recomputed in pass 2
            laf = (float)la;
        } else {
            float phs;
            const float la = logamp_f(load_amp<float>(src, dtype, base + i, &phs));
            const float d0 = r > 0 ? la - logamp_f(load_amp<float>(src, dtype, base + i - pw, &phs)) : 0.0f;
            const float d1 = c > 0 ? la - logamp_f(load_amp<float>(src, dtype, base + i - 1, &phs)) : 0.0f;
            g = sqrtf(d0 * d0 + d1 * d1);
            laf = la;
        }
        if (!isnan(g)) { gmin = fminf(gmin, g); gmax = fmaxf(gmax, g); }
        if (!isnan(laf)) { lmin = fminf(lmin, laf); lmax = fmaxf(lmax, laf); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gmin = fminf(gmin, __shfl_down(gmin, o, 64));
        gmax = fmaxf(gmax, __shfl_down(gmax, o, 64));
        lmin = fminf(lmin, __shfl_down(lmin, o, 64));
        lmax = fmaxf(lmax, __shfl_down(lmax, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[patch * 4 + 0], enc(gmin));
        atomicMax(&mm[patch * 4 + 1], enc(gmax));
        atomicMin(&mm[patch * 4 + 2], enc(lmin));
        atomicMax(&mm[patch * 4 + 3], enc(lmax));
    }
}

__global__ void prep_init_mm_kernel(unsigned* mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mm[i * 4 + 0] = 0xffffffffu; mm[i * 4 + 1] = 0u;
        mm[i * 4 + 2] = 0xffffffffu; mm[i * 4 + 3] = 0u;
    }
}

template <bool WIDE>
__global__ void prep_channels_kernel(const void* __restrict__ src, int dtype, int ph, int pw,
                                     const unsigned* __restrict__ mm, float* __restrict__ out) {
    const int patch = blockIdx.y;
    const int64_t base = (int64_t)patch * ph * pw;
    const bool is_complex = dtype == RFI_C128 || dtype == RFI_C64;
    // the float-rounded extrema bracket the exact ones within 1 ulp(float); the reference's
    // (g-min)/(max-min) is evaluated in source precision -> recover exact extrema for WIDE below
    const float gminf = dec(mm[patch * 4 + 0]), gmaxf = dec(mm[patch * 4 + 1]);
    const float lminf = dec(mm[patch * 4 + 2]), lmaxf = dec(mm[patch * 4 + 3]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ph * pw; i += gridDim.x * blockDim.x) {
        const int r = i / pw, c = i % pw;
        float ch0, ch1, ch2;
        if (WIDE) {
            double phs, tmp;
            const double la = logamp_d(load_amp<double>(src, dtype, base + i, &phs));
            const double d0 = r > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - pw, &tmp)) : 0.0;
            const double d1 = c > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - 1, &tmp)) : 0.0;
            const double g = sqrt(d0 * d0 + d1 * d1);
            const double gmin = (double)gminf, gmax = (double)gmaxf;
            ch0 = (gmax > gmin) ? (float)((g - gmin) / (gmax - gmin)) : 0.0f;
            if (is_complex) {
                double v = (la - (-3.0)) / (4.0 - (-3.0));
                v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                ch1 = (float)v;
                ch2 = (float)((phs + 3.141592653589793) / (2.0 * 3.141592653589793));
            } else {
                const double lmin = (double)lminf, lmax = (double)lmaxf;
                ch1 = (lmax > lmin) ? (float)((la - lmin) / (lmax - lmin)) : 0.0f;
                ch2 = 0.0f;
            }
        } else {
            float phs, tmp;
            const float la = logamp_f(load_amp<float>(src, dtype, base + i, &phs));
            const float d0 = r > 0 ? la - logamp_f(load_amp<float>(src, dtype, base + i - pw, &tmp)) : 0.0f;
            const float d1 = c > 0 ? la - logamp_f(load_amp<float>(src, dtype, base + i - 1, &tmp)) : 0.0f;
            const float g = sqrtf(d0 * d0 + d1 * d1);
            ch0 = (gmaxf > gminf) ? (g - gminf) / (gmaxf - gminf) : 0.0f;
            if (is_complex) {
                float v = (la + 3.0f) / 7.0f;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                ch1 = v;
                ch2 = (phs + 3.14159265358979f) / (2.0f * 3.14159265358979f);
            } else {
                ch1 = (lmaxf > lminf) ? (la - lminf) / (lmaxf - lminf) : 0.0f;
                ch2 = 0.0f;
            }
        }
        float* o = out + (base + i) * 3;
        o[0] = (ch0 - 0.485f) / 0.229f;
        o[1] = (ch1 - 0.456f) / 0.224f;
        o[2] = (ch2 - 0.406f) / 0.225f;
    }
}

// exact fp64 extrema for 64-bit inputs (second reduction in double through 64-bit atomics)
__device__ __forceinline__ unsigned long long encd(double f) {
    unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double decd(unsigned long long u) {
    return __longlong_as_double((long long)((u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u));
}

__global__ void prep_minmax64_kernel(const void* __restrict__ src, int dtype, int ph, int pw,
                                     unsigned long long* __restrict__ mm) {
    const int patch = blockIdx.y;
    const int64_t base = (int64_t)patch * ph * pw;
    double gmin = INFINITY, gmax = -INFINITY, lmin = INFINITY, lmax = -INFINITY;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ph * pw; i += gridDim.x * blockDim.x) {
        const int r = i / pw, c = i % pw;
        double phs;
        const double la = logamp_d(load_amp<double>(src, dtype, base + i, &phs));
        const double d0 = r > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - pw, &phs)) : 0.0;
        const double d1 = c > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - 1, &phs)) : 0.0;
        const double g = sqrt(d0 * d0 + d1 * d1);
        if (!isnan(g)) { gmin = fmin(gmin, g); gmax = fmax(gmax, g); }
        if (!isnan(la)) { lmin = fmin(lmin, la); lmax = fmax(lmax, la); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gmin = fmin(gmin, __shfl_down(gmin, o, 64));
        gmax = fmax(gmax, __shfl_down(gmax, o, 64));
        lmin = fmin(lmin, __shfl_down(lmin, o, 64));
        lmax = fmax(lmax, __shfl_down(lmax, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[patch * 4 + 0], encd(gmin));
        atomicMax(&mm[patch * 4 + 1], encd(gmax));
        atomicMin(&mm[patch * 4 + 2], encd(lmin));
        atomicMax(&mm[patch * 4 + 3], encd(lmax));
    }
}
__global__ void prep_init_mm64_kernel(unsigned long long* mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mm[i * 4 + 0] = ~0ull; mm[i * 4 + 1] = 0ull;
        mm[i * 4 + 2] = ~0ull; mm[i * 4 + 3] = 0ull;
    }
}
__global__ void prep_channels64_kernel(const void* __restrict__ src, int dtype, int ph, int pw,
                                       const unsigned long long* __restrict__ mm, float* __restrict__ out) {
    const int patch = blockIdx.y;
    const int64_t base = (int64_t)patch * ph * pw;
    const bool is_complex = dtype == RFI_C128;
    const double gmin = decd(mm[patch * 4 + 0]), gmax = decd(mm[patch * 4 + 1]);
    const double lmin = decd(mm[patch * 4 + 2]), lmax = decd(mm[patch * 4 + 3]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ph * pw; i += gridDim.x * blockDim.x) {
        const int r = i / pw, c = i % pw;
        double phs, tmp;
        const double la = logamp_d(load_amp<double>(src, dtype, base + i, &phs));
        const double d0 = r > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - pw, &tmp)) : 0.0;
        const double d1 = c > 0 ? la - logamp_d(load_amp<double>(src, dtype, base + i - 1, &tmp)) : 0.0;
        const double g = sqrt(d0 * d0 + d1 * d1);
        float ch0 = (gmax > gmin) ? (float)((g - gmin) / (gmax - gmin)) : 0.0f;
        float ch1, ch2;
        if (is_complex) {
            double v = (la - (-3.0)) / (4.0 - (-3.0));
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            ch1 = (float)v;
            ch2 = (float)((phs + 3.141592653589793) / (2.0 * 3.141592653589793));
        } else {
            ch1 = (lmax > lmin) ? (float)((la - lmin) / (lmax - lmin)) : 0.0f;
            ch2 = 0.0f;
        }
        float* o = out + (base + i) * 3;
        o[0] = (ch0 - 0.485f) / 0.229f;
        o[1] = (ch1 - 0.456f) / 0.224f;
        o[2] = (ch2 - 0.406f) / 0.225f;
    }
}

// ------------------------------------------------------------------ metrics
__device__ __forceinline__ bool nz(const void* p, int dtype, int64_t i) {
    return dtype == RFI_U8 ? reinterpret_cast<const uint8_t*>(p)[i] != 0
                           : reinterpret_cast<const float*>(p)[i] != 0.0f;
}
__global__ void confusion_kernel(const void* __restrict__ pred, int pdt, const void* __restrict__ truth,
                                 int tdt, int64_t count, unsigned long long* __restrict__ out3) {
    unsigned tp = 0, fp = 0, fn = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const bool p = nz(pred, pdt, i), t = nz(truth, tdt, i);
        tp += (p && t);
        fp += (p && !t);
        fn += (!p && t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_down(tp, o, 64);
        fp += __shfl_down(fp, o, 64);
        fn += __shfl_down(fn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out3[0], (unsigned long long)tp);
        atomicAdd(&out3[1], (unsigned long long)fp);
        atomicAdd(&out3[2], (unsigned long long)fn);
    }
}
__global__ void threshold_kernel(const float* __restrict__ logits, int64_t count, float thr,
                                 uint8_t* __restrict__ mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = (1.0f / (1.0f + expf(-logits[i]))) > thr ? 1 : 0;
}

}  // namespace

// minmax_ws: n*4 64-bit words
void launch_preprocess(rfi_ctx* ctx, const void* patches, int dtype, int n, int ph, int pw,
                       float* minmax_ws, float* out_nhwc) {
    RFI_REQUIRE(dtype >= RFI_C128 && dtype <= RFI_F32, "preprocess: unknown dtype");
    const int per = ph * pw;
    int bx = (int)cdiv(per, kBlock);
    if (bx > 64) bx = 64;
    const double in_b = dtype == RFI_C128 ? 16 : (dtype == RFI_F32 ? 4 : 8);
    const bool wide = dtype == RFI_C128 || dtype == RFI_F64;
    if (wide) {
        unsigned long long* mm = reinterpret_cast<unsigned long long*>(minmax_ws);
        {
            ProfScope ps(ctx, FAM_PREPROCESS);
            hipLaunchKernelGGL(prep_init_mm64_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, ctx->stream, mm, n);
            check_launch("prep_init_mm64");
        }
        {
            ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * in_b);
            hipLaunchKernelGGL(prep_minmax64_kernel, dim3(bx, n), dim3(kBlock), 0, ctx->stream, patches,
                               dtype, ph, pw, mm);
            check_launch("prep_minmax64");
        }
        {
            ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * (in_b + 12));
            hipLaunchKernelGGL(prep_channels64_kernel, dim3(bx, n), dim3(kBlock), 0, ctx->stream, patches,
                               dtype, ph, pw, mm, out_nhwc);
            check_launch("prep_channels64");
        }
    } else {
        unsigned* mm = reinterpret_cast<unsigned*>(minmax_ws);
        {
            ProfScope ps(ctx, FAM_PREPROCESS);
            hipLaunchKernelGGL(prep_init_mm_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, ctx->stream, mm, n);
            check_launch("prep_init_mm");
        }
        {
            ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * in_b);
            hipLaunchKernelGGL(prep_minmax_kernel<false>, dim3(bx, n), dim3(kBlock), 0, ctx->stream, patches,
                               dtype, ph, pw, mm);
            check_launch("prep_minmax");
        }
        {
            ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * (in_b + 12));
            hipLaunchKernelGGL(prep_channels_kernel<false>, dim3(bx, n), dim3(kBlock), 0, ctx->stream,
                               patches, dtype, ph, pw, mm, out_nhwc);
            check_launch("prep_channels");
        }
    }
}

void launch_confusion(rfi_ctx* ctx, const void* pred, int pred_dtype, const void* truth,
                      int truth_dtype, int64_t count, unsigned long long* counts3) {
    ProfScope ps(ctx, FAM_METRICS, 0, (double)count * ((pred_dtype ? 4 : 1) + (truth_dtype ? 4 : 1)));
    RFI_CHECK_HIP(hipMemsetAsync(counts3, 0, 3 * sizeof(unsigned long long), ctx->stream));
    int64_t blocks = cdiv(count, kBlock * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (count > 0) {
        hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, pred,
                           pred_dtype, truth, truth_dtype, count, counts3);
        check_launch("confusion");
    }
}

void launch_threshold(rfi_ctx* ctx, const float* logits, int64_t count, float threshold,
                      uint8_t* mask) {
    ProfScope ps(ctx, FAM_METRICS, 0, (double)count * 5);
    int64_t blocks = cdiv(count, kBlock * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, logits,
                       count, threshold, mask);
    check_launch("threshold");
}

}  // namespace rfi
