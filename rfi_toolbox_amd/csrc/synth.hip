// Synthetic complex waterfalls with exact RFI masks, generated in HBM (the step in
// front of the hot path, rfi_toolbox/data_generation/synthetic_generator.py:520-815).
//
// Same physical model as the reference's _generate_single_sample: noise N(n, 0.1 n) mJy (:553),
// t^order bandpass roll-off on the outer 10 % of channels (:658-673), rectangular / swept RFI events
// of constant amplitude summed into the signal with their union as the exact mask (:675-815),
// polarisation 1 = corr * signal + (1 - corr) * N(0, 0.1 n) + baseline, polarisations >= 2 noise
// only with an empty mask (:626-644), every polarisation multiplied by exp(i U(0, 2 pi)) (:647-648).
// NOT the reference's random stream (that is NumPy's global Mersenne twister, sequential by
// construction): each pixel draws from Philox4x32-10 keyed by the seed and counted by its own
// coordinates, so the result is independent of the launch geometry and reproducible anywhere.
// Parity is therefore distribution-level (tests compare moments with reference fixtures) and exact
// for everything deterministic given the event table: bandpass, signal sum, mask.
// HBM-bound: 16 B (complex128) or 8 B (complex64) + 1 B mask written per pixel, nothing read but the
// event table (scalar loads, shared by all lanes).
#include "kernels.hpp"

namespace rfi {
namespace {

constexpr int kBlock = 256;

struct U4 { unsigned x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        const U4 n{(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// (0, 1] and [0, 1) uniforms from 32 random bits
__device__ __forceinline__ double u01_open0(unsigned u) { return ((double)u + 1.0) * (1.0 / 4294967296.0); }
__device__ __forceinline__ double u01(unsigned u) { return (double)u * (1.0 / 4294967296.0); }
// two standard normals from two words (Box-Muller)
__device__ __forceinline__ void normal2(unsigned a, unsigned b, double& n0, double& n1) {
    const double r = sqrt(-2.0 * log(u01_open0(a)));
    double s, c;
    sincos(6.283185307179586 * u01(b), &s, &c);
    n0 = r * c;
    n1 = r * s;
}

struct SynthDev {
    unsigned long long seed;
    int n_samples, n_pol, C, T;
    double noise;
    int bandpass, order;
    double corr;
    const rfi_event* events;        // device
    const int* offsets;             // device, n_samples + 1
    int out_dtype;                  // RFI_C128 / RFI_C64
    void* planes;
    uint8_t* flags;
};

__device__ __forceinline__ double bandpass_gain(int r, int C, int order) {
    const int edge = (int)((double)C * 0.1);
    int i = -1;
    if (r < edge) i = r;
    else if (r >= C - edge) i = C - 1 - r;
    if (i < 0) return 1.0;
    return pow((double)i / (double)edge, (double)order);
}

__global__ void synth_kernel(SynthDev d) {
    const int64_t per_plane = (int64_t)d.C * d.T;
    const int64_t total = (int64_t)d.n_samples * d.n_pol * per_plane;
    const unsigned k0 = (unsigned)d.seed, k1 = (unsigned)(d.seed >> 32);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t plane = idx / per_plane;
        const int64_t pix = idx - plane * per_plane;
        const int s = (int)(plane / d.n_pol), p = (int)(plane % d.n_pol);
        const int r = (int)(pix / d.T), t = (int)(pix % d.T);
        const unsigned long long spix = (unsigned long long)s * (unsigned long long)per_plane + (unsigned long long)pix;
        // stream 0: per (sample, pixel), shared by polarisations 0 and 1 -> the common baseline noise
        const U4 a = philox4x32_10(U4{(unsigned)spix, (unsigned)(spix >> 32), 0u, 0u}, k0, k1);
        // stream 1 + p: per (sample, pol, pixel) -> phase, and the polarisation's own noise
        const U4 b = philox4x32_10(U4{(unsigned)spix, (unsigned)(spix >> 32), 1u + (unsigned)p, 0u}, k0, k1);
        double nb0, nb1, np0, np1;
        normal2(a.x, a.y, nb0, nb1);
        normal2(b.x, b.y, np0, np1);
        (void)nb1; (void)np1;
        const double gain = d.bandpass ? bandpass_gain(r, d.C, d.order) : 1.0;
        const double baseline = (d.noise + 0.1 * d.noise * nb0) * gain;
        double sig = 0.0;
        bool flagged = false;
        if (p < 2) {
            const int e0 = d.offsets[s], e1 = d.offsets[s + 1];
            for (int e = e0; e < e1; ++e) {
                const rfi_event ev = d.events[e];
                bool hit;
                if (ev.kind == 0) {
                    hit = r >= ev.r0 && r < ev.r1 && t >= ev.c0 && t < ev.c1;
                } else {   // sweep: centre channel follows f0 + (f1 - f0) (t / T)^order (:789-815)
                    const double x = (double)t / (double)d.T;
                    const int c = (int)((double)ev.r0 + (double)(ev.r1 - ev.r0) * (ev.c1 == 2 ? x * x : x));
                    int lo = c - ev.c0 / 2, hi = c + ev.c0 / 2;
                    lo = lo < 0 ? 0 : lo;
                    hi = hi > d.C ? d.C : hi;
                    hit = r >= lo && r < hi;
                }
                if (hit) {
                    sig += ev.amp;
                    flagged = true;
                }
            }
        }
        double real;
        if (p == 0) real = baseline + sig;
        else if (p == 1) real = d.corr * sig + (1.0 - d.corr) * (0.1 * d.noise * np0) + baseline;
        else real = d.noise + 0.1 * d.noise * np0;
        double sn, cs;
        sincos(6.283185307179586 * u01(b.z), &sn, &cs);
        if (d.out_dtype == RFI_C128) {
            reinterpret_cast<double2*>(d.planes)[idx] = make_double2(real * cs, real * sn);
        } else {
            reinterpret_cast<float2*>(d.planes)[idx] = make_float2((float)(real * cs), (float)(real * sn));
        }
        d.flags[idx] = (p < 2 && flagged) ? 1 : 0;
    }
}

}  // namespace

void launch_synth(rfi_ctx* ctx, unsigned long long seed, int n_samples, int n_pol, int C, int T, double noise,
                  int bandpass, int order, double corr, const rfi_event* events_dev, const int* offsets_dev,
                  int out_dtype, void* planes, uint8_t* flags) {
    const int64_t total = (int64_t)n_samples * n_pol * C * T;
    ProfScope ps(ctx, FAM_PREPROCESS, 0, (double)total * ((out_dtype == RFI_C128 ? 16 : 8) + 1));
    int64_t blocks = cdiv(total, kBlock * 2);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    SynthDev d{seed, n_samples, n_pol, C, T, noise, bandpass, order, corr, events_dev, offsets_dev, out_dtype, planes,
               flags};
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d);
    check_launch("synth");
}

}  // namespace rfi
