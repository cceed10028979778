// On-device form of the per-patch hot loop of Preprocessor.create_dataset
// (rfi_toolbox/preprocessing/preprocessor.py:366-384): complex/real patch -> 3 channels
// [gradient magnitude of log-amplitude (per-patch min-max), log-amplitude, phase] -> float32 ->
// ImageNet normalisation, written NHWC.  Math in fp64 for 64-bit inputs exactly as NumPy does it,
// rounded to fp32 where the reference casts (.astype(np.float32), :376) and normalised in fp32 (:783).
// Also the confusion counts of evaluation/metrics.py and the sigmoid threshold of evaluate_model.py.
// Algorithmic traffic 16 B in + 12 B out (+1 B label on the host side) per pixel for complex128;
// the per-patch min-max forces a second, 24 B/pixel pass over the output.
#include <algorithm>

#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kBlock = 256;

// Where pixel (y, x) of output patch `e` lives in the waterfall planes (C x T each), or -1 for the
// zero padding beyond a view's edge (preprocessor.py:478-560).  Views (:413-446): 0 plane,
// 1 plane[::-1, :], 2 plane.T, 3 plane.T[::-1, :].
struct GatherMap {
    const rfi_patch_src* table;     // device copy, one entry per output patch
    int C, T;
};
__device__ __forceinline__ int64_t gather_index(const GatherMap& g, const rfi_patch_src& e, int y, int x) {
    const int vy = e.row0 + y, vx = e.col0 + x;
    const bool tr = e.view >= 2;
    const int Hv = tr ? g.T : g.C, Wv = tr ? g.C : g.T;
    if (vy >= Hv || vx >= Wv) return -1;
    const int fy = (e.view & 1) ? Hv - 1 - vy : vy;            // rows flipped
    const int sy = tr ? vx : fy, sx = tr ? fy : vx;            // transpose: view[a][b] = plane[b][a]
    return ((int64_t)e.plane * g.C + sy) * g.T + sx;
}

template <typename T>
__device__ __forceinline__ T load_amp(const void* p, int dtype, int64_t i, T* phase) {
    // returns |z| (or |x| for real input) and the phase in *phase; i < 0 is a padded zero
    if (i < 0) {
        *phase = (T)0;
        return (T)0;
    }
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

// ordered-int encoding so double min/max can use 64-bit integer atomics (handles negatives)
__device__ __forceinline__ unsigned long long encd(double f) {
    unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double decd(unsigned long long u) {
    return __longlong_as_double((long long)((u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u));
}

// ------------------------------------------------------------------ two passes, one log/phase per pixel
// Pass 1: a block owns NPB consecutive pixels of one patch (row-major) plus the pw pixels before
// them (the row above); log-amplitude is computed ONCE per pixel into LDS, channels 1 (log-amp) and
// 2 (phase) leave in final form, the raw gradient magnitude goes to channel 0's slot and its
// per-patch extrema to mm (ordered-int atomics).  Real input: raw log-amp goes to channel 1's slot
// and its extrema to mm as well.  Pass 2 rescales the slots with the extrema.  T = double for
// 64-bit inputs (NumPy's precision), float for 32-bit inputs.

template <typename T> struct LogAmp;
template <> struct LogAmp<double> { static __device__ double f(double a) { return log10(a + 1e-10); } };
template <> struct LogAmp<float> { static __device__ float f(float a) { return log10f(a + 1e-10f); } };

template <typename T, bool GATHER>
__global__ void prep_pass1_kernel(const void* __restrict__ src, int dtype, int ph, int pw, int NPB, GatherMap gm,
                                  unsigned long long* __restrict__ mm, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* s_la = reinterpret_cast<T*>(smem_raw);                      // [pw + NPB]
    const int patch = blockIdx.y, npix = ph * pw;
    const int p0 = blockIdx.x * NPB;
    const int64_t base = (int64_t)patch * npix;
    const bool is_complex = dtype == RFI_C128 || dtype == RFI_C64;
    rfi_patch_src ent{};
    if constexpr (GATHER) ent = gm.table[patch];
    // log-amp of [p0 - pw, p0 + NPB) ∩ [0, npix); channels 1/2 of the block's own pixels
    for (int k = threadIdx.x; k < pw + NPB; k += blockDim.x) {
        const int p = p0 - pw + k;
        if (p < 0 || p >= npix) continue;
        T phs;
        int64_t si = base + p;
        if constexpr (GATHER) si = gather_index(gm, ent, p / pw, p % pw);
        const T la = LogAmp<T>::f(load_amp<T>(src, dtype, si, &phs));
        s_la[k] = la;
        if (p >= p0) {
            float* o = out + (base + p) * 3;
            if (is_complex) {
                T v = (la - (T)(-3.0)) / (T)7.0;
                v = v < (T)0 ? (T)0 : (v > (T)1 ? (T)1 : v);
                o[1] = ((float)v - 0.456f) / 0.224f;
                o[2] = ((float)((phs + (T)3.141592653589793) / (T)(2.0 * 3.141592653589793)) - 0.406f) / 0.225f;
            } else {
                o[1] = (float)la;                                   // rescaled by pass 2
                o[2] = (0.0f - 0.406f) / 0.225f;
            }
        }
    }
    __syncthreads();
    T gmin = INFINITY, gmax = -INFINITY, lmin = INFINITY, lmax = -INFINITY;
    for (int k = threadIdx.x; k < NPB; k += blockDim.x) {
        const int p = p0 + k;
        if (p >= npix) break;
        const int r = p / pw, c = p % pw;
        const T la = s_la[pw + k];
        const T d0 = r > 0 ? la - s_la[k] : (T)0;                  // pixel p - pw
        const T d1 = c > 0 ? la - s_la[pw + k - 1] : (T)0;
        const T g = sqrt(d0 * d0 + d1 * d1);
        out[(base + p) * 3] = (float)g;                             // rescaled by pass 2
        if (!isnan(g)) { gmin = g < gmin ? g : gmin; gmax = g > gmax ? g : gmax; }
        if (!isnan(la)) { lmin = la < lmin ? la : lmin; lmax = la > lmax ? la : lmax; }
    }
    double e[4] = {(double)gmin, (double)gmax, (double)lmin, (double)lmax};
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        e[0] = fmin(e[0], __shfl_down(e[0], o, 64));
        e[1] = fmax(e[1], __shfl_down(e[1], o, 64));
        e[2] = fmin(e[2], __shfl_down(e[2], o, 64));
        e[3] = fmax(e[3], __shfl_down(e[3], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[patch * 4 + 0], encd(e[0]));
        atomicMax(&mm[patch * 4 + 1], encd(e[1]));
        atomicMin(&mm[patch * 4 + 2], encd(e[2]));
        atomicMax(&mm[patch * 4 + 3], encd(e[3]));
    }
}

// pass 2: channel 0 <- (g - gmin)/(gmax - gmin) (0 when the patch is flat), real input also channel 1,
// then the ImageNet normalisation of those slots.  The extrema are the exact T-precision values; g
// was stored rounded to fp32, which moves the quotient by < 1 ulp(fp32).
__global__ void prep_pass2_kernel(int real_input, int npix, const unsigned long long* __restrict__ mm,
                                  float* __restrict__ out) {
    const int patch = blockIdx.y;
    const double gmin = decd(mm[patch * 4 + 0]), gmax = decd(mm[patch * 4 + 1]);
    const double lmin = decd(mm[patch * 4 + 2]), lmax = decd(mm[patch * 4 + 3]);
    const double gs = gmax > gmin ? 1.0 / (gmax - gmin) : 0.0, ls = lmax > lmin ? 1.0 / (lmax - lmin) : 0.0;
    float* o = out + (int64_t)patch * npix * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const float ch0 = gmax > gmin ? (float)(((double)o[i * 3] - gmin) / (gmax - gmin)) : 0.0f;
        o[i * 3] = (ch0 - 0.485f) / 0.229f;
        if (real_input) {
            const float ch1 = lmax > lmin ? (float)(((double)o[i * 3 + 1] - lmin) / (lmax - lmin)) : 0.0f;
            o[i * 3 + 1] = (ch1 - 0.456f) / 0.224f;
        }
    }
    (void)gs; (void)ls;
}

__global__ void prep_init_mm64_kernel(unsigned long long* mm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        mm[i * 4 + 0] = ~0ull; mm[i * 4 + 1] = 0ull;
        mm[i * 4 + 2] = ~0ull; mm[i * 4 + 3] = 0ull;
    }
}

// labels of the gathered patches: the same index map applied to the uint8 flag planes (padding = 0)
__global__ void gather_labels_kernel(const uint8_t* __restrict__ flags, GatherMap gm, int ps,
                                     uint8_t* __restrict__ out) {
    const int patch = blockIdx.y, npix = ps * ps;
    const rfi_patch_src ent = gm.table[patch];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int64_t si = gather_index(gm, ent, p / ps, p % ps);
        out[(int64_t)patch * npix + p] = si >= 0 && flags[si] != 0 ? 1 : 0;
    }
}
// any_out[patch] = 1 when the patch holds a flagged pixel (blank-patch removal, preprocessor.py:746-756);
// any_out must be zeroed before the launch
__global__ void patch_any_flag_kernel(const uint8_t* __restrict__ flags, GatherMap gm, int ps,
                                      unsigned* __restrict__ any_out) {
    const int patch = blockIdx.y, npix = ps * ps;
    const rfi_patch_src ent = gm.table[patch];
    bool any = false;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const int64_t si = gather_index(gm, ent, p / ps, p % ps);
        any |= si >= 0 && flags[si] != 0;
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(&any_out[patch], 1u);
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
                       float* minmax_ws, float* out_nhwc, const rfi_patch_src* table_dev, int C, int T) {
    RFI_REQUIRE(dtype >= RFI_C128 && dtype <= RFI_F32, "preprocess: unknown dtype");
    const int per = ph * pw;
    const double in_b = dtype == RFI_C128 ? 16 : (dtype == RFI_F32 ? 4 : 8);
    const bool wide = dtype == RFI_C128 || dtype == RFI_F64;
    const bool real_input = dtype == RFI_F64 || dtype == RFI_F32;
    unsigned long long* mm = reinterpret_cast<unsigned long long*>(minmax_ws);
    {
        ProfScope ps(ctx, FAM_PREPROCESS);
        hipLaunchKernelGGL(prep_init_mm64_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, ctx->stream, mm, n);
        check_launch("prep_init_mm64");
    }
    {
        ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * (in_b + 12));
        // pixels per block: at least 4 rows so the halo row costs <= 25 % extra log/phase work
        const int esz = wide ? 8 : 4;
        int npb = std::max(2048, 4 * pw);
        npb = std::min(npb, 64 * 1024 / esz - pw);
        RFI_REQUIRE(npb >= pw, "preprocess: patch rows this long are not supported");
        const dim3 grid((unsigned)cdiv(per, npb), n);
        const size_t lds = (size_t)(pw + npb) * esz;
        const GatherMap gm{table_dev, C, T};
        if (table_dev) {
            if (wide)
                hipLaunchKernelGGL((prep_pass1_kernel<double, true>), grid, dim3(kBlock), lds, ctx->stream, patches,
                                   dtype, ph, pw, npb, gm, mm, out_nhwc);
            else
                hipLaunchKernelGGL((prep_pass1_kernel<float, true>), grid, dim3(kBlock), lds, ctx->stream, patches,
                                   dtype, ph, pw, npb, gm, mm, out_nhwc);
        } else {
            if (wide)
                hipLaunchKernelGGL((prep_pass1_kernel<double, false>), grid, dim3(kBlock), lds, ctx->stream, patches,
                                   dtype, ph, pw, npb, gm, mm, out_nhwc);
            else
                hipLaunchKernelGGL((prep_pass1_kernel<float, false>), grid, dim3(kBlock), lds, ctx->stream, patches,
                                   dtype, ph, pw, npb, gm, mm, out_nhwc);
        }
        check_launch("prep_pass1");
    }
    {
        ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * 24);
        int bx = (int)cdiv(per, kBlock * 4);
        const int cap = std::max(1, 4096 / n);              // ~4096 blocks in total
        if (bx > cap) bx = cap;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(prep_pass2_kernel, dim3(bx, n), dim3(kBlock), 0, ctx->stream, real_input ? 1 : 0, per, mm,
                           out_nhwc);
        check_launch("prep_pass2");
    }
}

void launch_gather_labels(rfi_ctx* ctx, const uint8_t* flags, const rfi_patch_src* table_dev, int C, int T, int n,
                          int ps, uint8_t* out) {
    ProfScope ps_(ctx, FAM_PREPROCESS, 0, (double)n * ps * ps * 2);
    int bx = (int)cdiv((int64_t)ps * ps, kBlock * 4);
    hipLaunchKernelGGL(gather_labels_kernel, dim3(bx, n), dim3(kBlock), 0, ctx->stream, flags, GatherMap{table_dev, C, T},
                       ps, out);
    check_launch("gather_labels");
}
void launch_patch_any_flag(rfi_ctx* ctx, const uint8_t* flags, const rfi_patch_src* table_dev, int C, int T, int n,
                           int ps, unsigned* any_out) {
    ProfScope ps_(ctx, FAM_PREPROCESS, 0, (double)n * ps * ps);
    RFI_CHECK_HIP(hipMemsetAsync(any_out, 0, (size_t)n * sizeof(unsigned), ctx->stream));
    int bx = (int)cdiv((int64_t)ps * ps, kBlock * 4);
    hipLaunchKernelGGL(patch_any_flag_kernel, dim3(bx, n), dim3(kBlock), 0, ctx->stream, flags,
                       GatherMap{table_dev, C, T}, ps, any_out);
    check_launch("patch_any_flag");
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
