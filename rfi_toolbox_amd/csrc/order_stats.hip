// Per-patch order statistics and the real-input branch of Preprocessor.create_dataset on the GPU
// (rfi_toolbox/preprocessing/preprocessor.py: median normalise :646-670, SQRT / LOG10 stretch with
// the MAD of the finite values replacing infinities :672-706, MAD flags :708-745).
//
// np.median / np.nanmedian of a patch = mean of the order statistics (n-1)/2 and n/2 of its n valid
// values.  One workgroup per patch finds an order statistic EXACTLY by radix selection on the
// order-preserving 64-bit image of the doubles: 8 passes of 8 bits, a 256-bin LDS histogram per pass
// over the elements that still match the prefix.  The patch (128 KiB at 128 x 128) stays in L2.
// Everything here is byte-moving / integer work bound by L2 bandwidth; no MFMA.
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ unsigned long long ord64(double f) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unord64(unsigned long long k) {
    return __longlong_as_double((long long)((k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k));
}

// value of element i of the patch under the selected view, and whether it takes part
struct OsView {
    const double* v;          // patch base
    int absdev;               // 1: |v - centre|
    double centre;
    int finite_only;          // 0: skip NaN (np.nanmedian / _mad); 1: skip NaN and +-inf (stretch's `finite`)
    int f32;                  // the data are float32 values held in doubles: every arithmetic result is rounded to float32
                              // (what NumPy's float32 loops give for float32 input: + - * / sqrt are correctly rounded in
                              // both, and rounding the exact double result once more to float32 is the float32 result)
};
__device__ __forceinline__ double r32(double x) { return (double)(float)x; }
__device__ __forceinline__ bool os_get(const OsView& w, int i, double& x) {
    x = w.v[i];
    if (w.finite_only ? !isfinite(x) : isnan(x)) return false;
    if (w.absdev) {
        x = fabs(x - w.centre);
        if (w.f32) x = r32(x);
    }
    return true;
}

// rank-r element (0-based, r < count of participating elements) of the view; all threads return it
__device__ double os_select(const OsView& w, int per, unsigned r, unsigned* hist, unsigned long long* s_pref,
                            unsigned* s_rank) {
    unsigned long long prefix = 0, mask = 0;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        hist[threadIdx.x] = 0;                                 // kBlock == 256 bins
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += kBlock) {
            double x;
            if (!os_get(w, i, x)) continue;
            const unsigned long long k = ord64(x);
            if ((k & mask) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned cum = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (cum + hist[b] > r) break;
                cum += hist[b];
            }
            *s_pref = prefix | ((unsigned long long)b << shift);
            *s_rank = r - cum;
        }
        __syncthreads();
        prefix = *s_pref;
        r = *s_rank;
        mask |= 0xffull << shift;
        __syncthreads();
    }
    return unord64(prefix);
}

// out[patch] = median of the view (NaN when nothing participates); cnt_out[patch] = participants
__global__ void patch_median_kernel(const double* __restrict__ v, int per, int absdev,
                                    const double* __restrict__ centre, int finite_only, int f32,
                                    double* __restrict__ out, int* __restrict__ cnt_out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long s_pref;
    __shared__ unsigned s_rank, s_cnt;
    const int patch = blockIdx.x;
    OsView w{v + (size_t)patch * per, absdev, centre ? centre[patch] : 0.0, finite_only, f32};
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned c = 0;
    for (int i = threadIdx.x; i < per; i += kBlock) {
        double x;
        c += os_get(w, i, x) ? 1u : 0u;
    }
    atomicAdd(&s_cnt, c);
    __syncthreads();
    const unsigned n = s_cnt;
    if (cnt_out && threadIdx.x == 0) cnt_out[patch] = (int)n;
    if (n == 0) {
        if (threadIdx.x == 0) out[patch] = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    const double a = os_select(w, per, (n - 1) / 2, hist, &s_pref, &s_rank);
    const double b = (n & 1u) ? a : os_select(w, per, n / 2, hist, &s_pref, &s_rank);
    if (threadIdx.x == 0) out[patch] = (n & 1u) ? a : (f32 ? r32(a + b) : (a + b)) / 2.0;     // np.mean of the two middle values
}

// v /= (med > 0 ? med : 1)   (_normalize, :646-670)
__global__ void scale_by_median_kernel(double* __restrict__ v, int per, const double* __restrict__ med, int f32) {
    const int patch = blockIdx.y;
    const double m = med[patch];
    const double s = m > 0 ? m : 1.0;
    double* p = v + (size_t)patch * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) p[i] = f32 ? r32(p[i] / s) : p[i] / s;
}
// v = sqrt(|v|) or log10(|v|)   (_apply_stretch, :672-706)
__global__ void stretch_kernel(double* __restrict__ v, int64_t total, int kind, int f32) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = fabs(v[i]);
        const double r = kind == 1 ? sqrt(a) : log10(a);
        v[i] = f32 ? r32(r) : r;         // (float32 log10: correctly rounded here, within 1 ulp of it in NumPy)
    }
}
// infinities <- MAD of the patch's finite values (0 when it has none)
__global__ void replace_inf_kernel(double* __restrict__ v, int per, const double* __restrict__ mad,
                                   const int* __restrict__ nfinite) {
    const int patch = blockIdx.y;
    const double fill = nfinite[patch] > 0 ? mad[patch] : 0.0;
    double* p = v + (size_t)patch * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x)
        if (isinf(p[i])) p[i] = fill;
}
// flags = (p > med + mad*sigma) | (p < med - mad*sigma)   (_generate_mad_flags, :708-745)
__global__ void mad_flags_kernel(const double* __restrict__ v, int per, const double* __restrict__ med,
                                 const double* __restrict__ mad, double sigma, uint8_t* __restrict__ flags, int f32) {
    const int patch = blockIdx.y;
    double hi = med[patch] + mad[patch] * sigma, lo = med[patch] - mad[patch] * sigma;
    if (f32) {                                   // float32 scalars: sigma is cast to float32 first (NumPy's weak Python scalar)
        const double t = r32(mad[patch] * r32(sigma));
        hi = r32(med[patch] + t);
        lo = r32(med[patch] - t);
    }
    const double* p = v + (size_t)patch * per;
    uint8_t* f = flags + (size_t)patch * per;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x)
        f[i] = (p[i] > hi || p[i] < lo) ? 1 : 0;
}
// |z| of complex128 / complex64, or a widening copy of float32
__global__ void to_abs_f64_kernel(const void* __restrict__ src, int dtype, int64_t total, double* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        double r;
        if (dtype == RFI_C128) r = hypot(reinterpret_cast<const double*>(src)[2 * i], reinterpret_cast<const double*>(src)[2 * i + 1]);
        else if (dtype == RFI_C64) r = (double)hypotf(reinterpret_cast<const float*>(src)[2 * i], reinterpret_cast<const float*>(src)[2 * i + 1]);
        else if (dtype == RFI_F64) r = reinterpret_cast<const double*>(src)[i];
        else r = (double)reinterpret_cast<const float*>(src)[i];
        dst[i] = r;
    }
}

__global__ void narrow_f32_kernel(const double* __restrict__ src, int64_t total, float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

int grid1(int64_t total) {
    int64_t b = cdiv(total, kBlock * 4);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

void launch_patch_median(rfi_ctx* ctx, const double* v, int n, int per, bool absdev, const double* centre,
                         bool finite_only, double* out, int* cnt_out, bool f32) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * 8 * 17);
    hipLaunchKernelGGL(patch_median_kernel, dim3(n), dim3(kBlock), 0, ctx->stream, v, per, absdev ? 1 : 0, centre,
                       finite_only ? 1 : 0, f32 ? 1 : 0, out, cnt_out);
    check_launch("patch_median");
}
void launch_scale_by_median(rfi_ctx* ctx, double* v, int n, int per, const double* med, bool f32) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * 16);
    hipLaunchKernelGGL(scale_by_median_kernel, dim3(grid1(per), n), dim3(kBlock), 0, ctx->stream, v, per, med, f32 ? 1 : 0);
    check_launch("scale_by_median");
}
void launch_stretch(rfi_ctx* ctx, double* v, int64_t total, int kind, bool f32) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)total * 16);
    hipLaunchKernelGGL(stretch_kernel, dim3(grid1(total)), dim3(kBlock), 0, ctx->stream, v, total, kind, f32 ? 1 : 0);
    check_launch("stretch");
}
void launch_replace_inf(rfi_ctx* ctx, double* v, int n, int per, const double* mad, const int* nfinite) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * 8);
    hipLaunchKernelGGL(replace_inf_kernel, dim3(grid1(per), n), dim3(kBlock), 0, ctx->stream, v, per, mad, nfinite);
    check_launch("replace_inf");
}
void launch_mad_flags(rfi_ctx* ctx, const double* v, int n, int per, const double* med, const double* mad,
                      double sigma, uint8_t* flags, bool f32) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)n * per * 9);
    hipLaunchKernelGGL(mad_flags_kernel, dim3(grid1(per), n), dim3(kBlock), 0, ctx->stream, v, per, med, mad, sigma,
                       flags, f32 ? 1 : 0);
    check_launch("mad_flags");
}
void launch_to_abs_f64(rfi_ctx* ctx, const void* src, int dtype, int64_t total, double* dst) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)total * 24);
    hipLaunchKernelGGL(to_abs_f64_kernel, dim3(grid1(total)), dim3(kBlock), 0, ctx->stream, src, dtype, total, dst);
    check_launch("to_abs_f64");
}

void launch_narrow_f32(rfi_ctx* ctx, const double* src, int64_t total, float* dst) {
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)total * 12);
    hipLaunchKernelGGL(narrow_f32_kernel, dim3(grid1(total)), dim3(kBlock), 0, ctx->stream, src, total, dst);
    check_launch("narrow_f32");
}

}  // namespace rfi
