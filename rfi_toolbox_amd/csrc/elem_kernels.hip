// HBM-bound kernels of the U-Net step: batch-norm statistics / backward, pool, head, loss,
// optimiser, layout changes.  gfx950 (wave64).  Reductions over pixels accumulate in fp64 per
// thread (what torch's CPU BatchNorm does, acc_type<float> == double) and are two-stage
// (per-block partials + a finishing launch) so every result is bitwise run-to-run reproducible.
// BatchNorm-apply is x*scale + shift with the product and the sum rounded SEPARATELY (the build
// uses -ffp-contract=off): torch's CPU kernel applies x*alpha+beta with a vector multiply and a
// vector add, and the ReLU mask of elements at the threshold follows that rounding.
#include <hip/hip_ext.h>

#include "kernels.hpp"

namespace rfi {

namespace {

constexpr int kBlock = 256;
constexpr int kFinCh = 2;                    // finishing kernels: channels per block ...
constexpr int kFinLanes = kBlock / kFinCh;   // ... x record lanes

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// finishing kernels: thread = (channel lane cl = tid % kFinCh, record lane tid / kFinCh).  Sums the
// record lanes of each channel lane: xor-shuffles over the lane bits above kFinCh inside a wave,
// then the 4 waves through LDS.  Result valid in threads tid < kFinCh.  red: 4 * kFinCh doubles.
__device__ __forceinline__ double finish_sum(double v, double* red) {
#pragma unroll
    for (int o = 32; o >= kFinCh; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // red may still be read from a previous call
    if (lane < kFinCh) red[wave * kFinCh + lane] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x < kFinCh)
        for (int w = 0; w < kBlock / 64; ++w) t += red[w * kFinCh + threadIdx.x];
    return t;
}

// channel-lane geometry shared by the per-channel reductions over a [M][C] tensor:
// a block = CL channel lanes x RL row lanes (CL*RL == 256), grid = (row blocks, channel blocks)
struct ChanGeom {
    int V;                 // channels per lane: 4 (float4 loads) when C % 4 == 0, else 1
    int CL, RL, cblocks;
    int64_t rows_per_block;
    int rblocks;
};

static ChanGeom chan_geom(int64_t M, int C, int max_rblocks, bool allow_vec = true, int force_v = 0) {
    ChanGeom g;
    g.V = force_v ? force_v : (allow_vec && C % 4 == 0) ? 4 : 1;
    const int lanes_needed = C / g.V;
    int cl = (g.V >= 4) ? 1 : 4;
    while (cl < lanes_needed && cl < 64) cl <<= 1;
    g.CL = cl;
    g.RL = kBlock / cl;
    g.cblocks = (int)cdiv(lanes_needed, cl);
    int64_t rpb = cdiv(M, max_rblocks);
    int64_t min_rows = (int64_t)g.RL * 8;
    if (rpb < min_rows) rpb = min_rows;
    rpb = cdiv(rpb, g.RL) * g.RL;
    g.rows_per_block = rpb;
    g.rblocks = (int)cdiv(M, rpb);
    return g;
}

// V consecutive channels per lane (V = 4: one 16-byte load per row and lane)
template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&v)[V]) {
    if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = p[0];
    }
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&v)[V]) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else p[0] = v[0];
}
// V channels of the raw conv output Y at element index idx: float32 or (y16) bfloat16 (kernels.hpp, YRef)
template <int V>
__device__ __forceinline__ void ldy(const void* y, int y16, int64_t idx, float (&v)[V]) {
    if (y16) {
        const unsigned short* q = static_cast<const unsigned short*>(y) + idx;
        if constexpr (V == 4) {
            const uint2 t = *reinterpret_cast<const uint2*>(q);
            v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
            v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
        } else {
            v[0] = __builtin_bit_cast(float, (unsigned)q[0] << 16);
        }
    } else {
        ldv<V>(static_cast<const float*>(y) + idx, v);
    }
}
__device__ __forceinline__ float ldy1(const void* y, int y16, int64_t idx) {
    float v[1];
    ldy<1>(y, y16, idx, v);
    return v[0];
}
// four values -> bfloat16 (RNE; NaN stays NaN), stored as 8 bytes; v is left holding the rounded values
__device__ __forceinline__ void round_store_bf16x4(unsigned short* p, float (&v)[4]) {
    unsigned short q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        q[e] = __builtin_bit_cast(unsigned short, h);
        v[e] = (float)h;
    }
    *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)q[0] | ((unsigned)q[1] << 16), (unsigned)q[2] | ((unsigned)q[3] << 16));
}
// block-level sum over the RL row lanes of NQ per-lane quantities of V channels each;
// lane rl == 0 ends up with the totals in acc.  red: NQ * V * 256 doubles of LDS.
template <int V, int NQ>
__device__ __forceinline__ void row_lane_reduce(double (&acc)[NQ][V], double* red, int CL, int RL, int cl,
                                                int rl) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int v = 0; v < V; ++v) red[(q * V + v) * kBlock + threadIdx.x] = acc[q][v];
    __syncthreads();
    // row lanes 0..3 each fold a quarter of the RL entries (four independent chains of LDS reads instead of one serial one:
    // with 8 channels per lane and 64 row lanes the single chain was longer than the block's whole main loop), then lane 0
    // adds the four in fixed order
    const int NF = RL >= 4 ? 4 : 1;
    if (rl < NF) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int v = 0; v < V; ++v) {
                double t0 = rl == 0 ? acc[q][v] : 0.0, t1 = 0.0;
                const double* rp = red + (q * V + v) * kBlock + cl;
                int k = rl == 0 ? NF : rl;
                for (; k + NF < RL; k += 2 * NF) {
                    t0 += rp[k * CL];
                    t1 += rp[(k + NF) * CL];
                }
                if (k < RL) t0 += rp[k * CL];
                acc[q][v] = t0 + t1;
            }
    }
    if (NF > 1) {
        __syncthreads();
        if (rl > 0 && rl < NF) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < V; ++v) red[(q * V + v) * kBlock + threadIdx.x] = acc[q][v];
        }
        __syncthreads();
        if (rl == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    double t = acc[q][v];
                    for (int j = 1; j < NF; ++j) t += red[(q * V + v) * kBlock + j * CL + cl];
                    acc[q][v] = t;
                }
        }
    }
}

// ReLU (slope == 0) / LeakyReLU and their derivative factor, as torch defines them (x > 0 ? x : x*slope)
__device__ __forceinline__ float act_f(float a, float slope) { return a > 0.0f ? a : (slope != 0.0f ? a * slope : 0.0f); }
__device__ __forceinline__ float dact_f(float z, float g, float slope) { return z > 0.0f ? g : g * slope; }

// ------------------------------------------------------------------ batch-norm statistics
// partial[(rb*C + c)*2 + {0,1}] = sum x, sum x^2 over the block's rows (double)
template <int V>
__global__ void bn_stats_kernel(const float* __restrict__ y, int64_t M, int C, int CL,
                                int64_t rows_per_block, double* __restrict__ partial) {
    __shared__ double red[2 * V * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * V;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[2][V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[0][v] = acc[1][v] = 0;
    if (c < C) {
        for (int64_t r = r0 + rl; r < r1; r += RL) {
            float x[V];
            ldv<V>(y + r * C + c, x);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const double d = (double)x[v];
                acc[0][v] += d;
                acc[1][v] += d * d;
            }
        }
    }
    row_lane_reduce<V, 2>(acc, red, CL, RL, cl, rl);
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 0] = acc[0][v];
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 1] = acc[1][v];
        }
    }
}

// one block per 8 channels, 32 record lanes
__global__ void bn_finalize_kernel(const double* __restrict__ partial, int records, int C,
                                   double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, int ema_repeats,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ var_out) {
    __shared__ double red[2 * kBlock];
    const int cl = threadIdx.x & (kFinCh - 1), rl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    double s1 = 0, s2 = 0;
    if (c < C) {
        for (int r = rl; r < records; r += kFinLanes) {
            s1 += partial[((int64_t)r * C + c) * 2 + 0];
            s2 += partial[((int64_t)r * C + c) * 2 + 1];
        }
    }
    s1 = finish_sum(s1, red);
    s2 = finish_sum(s2, red);
    if (rl == 0 && c < C) {
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        const double invstd = 1.0 / sqrt(var + 1e-5);
        const float meanf = (float)mean, invf = (float)invstd;
        mean_out[c] = meanf;
        invstd_out[c] = invf;
        if (var_out) var_out[c] = (float)var;
        const float sc = gamma[c] * invf;
        scale[c] = sc;
        shift[c] = beta[c] - meanf * sc;
        if (running_mean) {
            const double var_u = count > 1 ? var * (count / (count - 1.0)) : var;
            float rm = running_mean[c], rv = running_var[c];
            const float vuf = (float)var_u;
            for (int k = 0; k < ema_repeats; ++k) {
                rm = (1.0f - 0.1f) * rm + 0.1f * meanf;
                rv = (1.0f - 0.1f) * rv + 0.1f * vuf;
            }
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float inv = (float)(1.0 / sqrt((double)rv[c] + 1e-5));
        float sc = gamma[c] * inv;
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}

// ------------------------------------------------------------------ batch-norm backward
// partial[(rb*C+c)*2 + {0,1}] = sum dz, sum dz*xhat;  dz = da * (y*scale+shift > 0)
template <int V>
__global__ void bn_bwd_reduce_kernel(const void* __restrict__ da, int da16, const void* __restrict__ y, int y16, int64_t yps,
                                     int64_t M, int C, int CL, int64_t rows_per_block,
                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     double* __restrict__ partial, float slope) {
    __shared__ double red[2 * V * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * V;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[2][V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[0][v] = acc[1][v] = 0;
    if (c < C) {
        float sc[V], sh[V], mu[V], is[V];
        ldv<V>(scale + c, sc); ldv<V>(shift + c, sh); ldv<V>(mean + c, mu); ldv<V>(invstd + c, is);
        // four rows per trip: the eight loads issue before the first sum (one row per trip left two 16-byte loads in flight
        // per lane: 4 TB/s on the 128 x 128 levels); the rows are summed in the same order as before -- bit-identical records
        constexpr int U = 4;
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * RL) {
            float yv[U][V], dv[U][V];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const int64_t r = rb + (int64_t)k * RL;
                if (r < r1) {
                    ldy<V>(y, y16, r * yps + c, yv[k]);
                    ldy<V>(da, da16, r * C + c, dv[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (rb + (int64_t)k * RL < r1) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float dz = dact_f(yv[k][v] * sc[v] + sh[v], dv[k][v], slope);
                        const float xh = (yv[k][v] - mu[v]) * is[v];
                        acc[0][v] += (double)dz;
                        acc[1][v] += (double)dz * (double)xh;
                    }
                }
            }
        }
    }
    row_lane_reduce<V, 2>(acc, red, CL, RL, cl, rl);
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 0] = acc[0][v];
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 1] = acc[1][v];
        }
    }
}

// ---- the same two passes for the bfloat16 data flow (dA and Y bfloat16, dense / chunk-padded rows; C % 8 == 0): 8 channels per
// lane -- one 16-byte load per tensor, row and lane, ONE 16-byte store of the bf16 dY row piece -- and the per-channel sums
// in float32 over 8 rows at a time, then fp64 (the generic kernels above: 8-byte loads, four 2-byte stores and two to five
// fp64 operations per element, 2.8 - 3.9 TB/s on the 1024 x 1024 x 64 tensors; these: the elementwise kernels' rate)
__device__ __forceinline__ void ld8h(const unsigned short* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
    v[4] = __builtin_bit_cast(float, t.z << 16); v[5] = __builtin_bit_cast(float, t.z & 0xffff0000u);
    v[6] = __builtin_bit_cast(float, t.w << 16); v[7] = __builtin_bit_cast(float, t.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {      // RNE; NaN stays NaN
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__global__ __launch_bounds__(kBlock) void bn_bwd_reduce16_kernel(const unsigned short* __restrict__ da, const unsigned short* __restrict__ y, int64_t yps, int64_t M,
                                       int C, int CL, int64_t rows_per_block, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                       double* __restrict__ partial, float slope) {
    __shared__ double red[2 * 8 * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * 8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[2][8];
#pragma unroll
    for (int v = 0; v < 8; ++v) acc[0][v] = acc[1][v] = 0;
    if (c < C) {
        float sc[8], sh[8], mu[8], is[8];
        ld8f(scale + c, sc); ld8f(shift + c, sh); ld8f(mean + c, mu); ld8f(invstd + c, is);
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)RL * 8) {
            float p1[8], p2[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) p1[v] = p2[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t r = rb + (int64_t)k * RL;
                if (r < r1) {
                    float yv[8], dv[8];
                    ld8h(y + r * yps + c, yv);
                    ld8h(da + r * C + c, dv);
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float dz = dact_f(yv[v] * sc[v] + sh[v], dv[v], slope);
                        p1[v] += dz;
                        p2[v] += dz * ((yv[v] - mu[v]) * is[v]);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                acc[0][v] += (double)p1[v];
                acc[1][v] += (double)p2[v];
            }
        }
    }
    row_lane_reduce<8, 2>(acc, red, CL, RL, cl, rl);
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 0] = acc[0][v];
            partial[((int64_t)blockIdx.x * C + c + v) * 2 + 1] = acc[1][v];
        }
    }
}
// dY as bf16 planes (P = 1); SUMS: partial[rb * C + c] = sum dy (the conv bias gradient; layers without a bias skip it)
template <bool SUMS>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply16_kernel(const unsigned short* __restrict__ da, const unsigned short* __restrict__ y, int64_t yps, int64_t M,
                                      int C, int CL, int64_t rows_per_block, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
                                      const float* __restrict__ gamma, const float* __restrict__ c1, const float* __restrict__ c2,
                                      double* __restrict__ partial, float slope, unsigned short* __restrict__ planes, int64_t pl_stride) {
    __shared__ double red[SUMS ? 8 * kBlock : 1];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * 8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[1][8];
#pragma unroll
    for (int v = 0; v < 8; ++v) acc[0][v] = 0;
    if (c < C) {
        float sc[8], sh[8], mu[8], is[8], g[8], k1[8], k2[8];
        ld8f(scale + c, sc); ld8f(shift + c, sh); ld8f(mean + c, mu); ld8f(invstd + c, is);
        ld8f(gamma + c, g); ld8f(c1 + c, k1); ld8f(c2 + c, k2);
#pragma unroll
        for (int v = 0; v < 8; ++v) g[v] = g[v] * is[v];
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)RL * 8) {
            float p1[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) p1[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t r = rb + (int64_t)k * RL;
                if (r < r1) {
                    float yv[8], dv[8], o[8];
                    ld8h(y + r * yps + c, yv);
                    ld8h(da + r * C + c, dv);
#pragma unroll
                    for (int v = 0; v < 8; ++v) {
                        const float dz = dact_f(yv[v] * sc[v] + sh[v], dv[v], slope);
                        const float xh = (yv[v] - mu[v]) * is[v];
                        o[v] = g[v] * (dz - k1[v] - xh * k2[v]);
                        if (SUMS) p1[v] += o[v];
                    }
                    *reinterpret_cast<uint4*>(planes + r * pl_stride + c) =
                        make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
                }
            }
            if (SUMS) {
#pragma unroll
                for (int v = 0; v < 8; ++v) acc[0][v] += (double)p1[v];
            }
        }
    }
    if constexpr (SUMS) {
        row_lane_reduce<8, 1>(acc, red, CL, RL, cl, rl);
        if (rl == 0 && c < C) {
#pragma unroll
            for (int v = 0; v < 8; ++v) partial[(int64_t)blockIdx.x * C + c + v] = acc[0][v];
        }
    }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partial, int records, int C,
                                       double count, float* c1, float* c2, float* dgamma,
                                       float* dbeta) {
    __shared__ double red[2 * kBlock];
    const int cl = threadIdx.x & (kFinCh - 1), rl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    double s1 = 0, s2 = 0;
    if (c < C) {
        for (int r = rl; r < records; r += kFinLanes) {
            s1 += partial[((int64_t)r * C + c) * 2 + 0];
            s2 += partial[((int64_t)r * C + c) * 2 + 1];
        }
    }
    s1 = finish_sum(s1, red);
    s2 = finish_sum(s2, red);
    if (rl == 0 && c < C) {
        c1[c] = (float)(s1 / count);
        c2[c] = (float)(s2 / count);
        dgamma[c] = (float)s2;
        dbeta[c] = (float)s1;
    }
}

// in place: da <- dy;  partial[rb*C+c] = sum dy (double)
template <int V>
__global__ void bn_bwd_apply_kernel(void* __restrict__ da_, int da16, const void* __restrict__ y, int y16, int64_t yps, int64_t M,
                                    int C, int CL, int64_t rows_per_block,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ c1,
                                    const float* __restrict__ c2, double* __restrict__ partial, float slope,
                                    unsigned short* __restrict__ planes, int64_t pl_stride, int P,
                                    const float* __restrict__ head_dl, const float* __restrict__ head_w) {
    // head_dl != null: da is NOT read -- it is d[pixel] * w[channel], the gradient a one-channel 1x1 head sends down
    // (launch_head_bwd with skip_da); da_ is output only
    __shared__ double red[V * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * V;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[1][V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[0][v] = 0;
    if (c < C) {
        float sc[V], sh[V], mu[V], is[V], g[V], k1[V], k2[V];
        ldv<V>(scale + c, sc); ldv<V>(shift + c, sh); ldv<V>(mean + c, mu); ldv<V>(invstd + c, is);
        ldv<V>(gamma + c, g); ldv<V>(c1 + c, k1); ldv<V>(c2 + c, k2);
#pragma unroll
        for (int v = 0; v < V; ++v) g[v] = g[v] * is[v];
        float hw[V];
#pragma unroll
        for (int v = 0; v < V; ++v) hw[v] = head_dl ? head_w[c + v] : 0.0f;
        // four rows per trip, loads first (as bn_bwd_reduce_kernel; same order of the sums)
        constexpr int U = 4;
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * RL) {
          float yu[U][V], du[U][V];
#pragma unroll
          for (int k = 0; k < U; ++k) {
              const int64_t r = rb + (int64_t)k * RL;
              if (r < r1) {
                  ldy<V>(y, y16, r * yps + c, yu[k]);
                  if (head_dl) {
                      const float d = head_dl[r];
#pragma unroll
                      for (int v = 0; v < V; ++v) du[k][v] = d * hw[v];
                  } else {
                      ldy<V>(da_, da16, r * C + c, du[k]);
                  }
              }
          }
#pragma unroll
          for (int k = 0; k < U; ++k) {
            const int64_t r = rb + (int64_t)k * RL;
            if (r >= r1) break;
            float yv[V], dv[V], o[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { yv[v] = yu[k][v]; dv[v] = du[k][v]; }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float dz = dact_f(yv[v] * sc[v] + sh[v], dv[v], slope);
                const float xh = (yv[v] - mu[v]) * is[v];
                o[v] = g[v] * (dz - k1[v] - xh * k2[v]);
                acc[0][v] += (double)o[v];
            }
            if (planes) {                      // dY leaves as bf16 pieces for the MFMA consumers (planes.hpp)
                unsigned short* q = planes + r * pl_stride + (int64_t)(c >> 4) * (P * 16) + (c & 15);
                float res[V];
#pragma unroll
                for (int v = 0; v < V; ++v) res[v] = o[v];
                for (int p = 0; p < P; ++p) {
                    unsigned short hq[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const __bf16 h = (__bf16)res[v];                                  // RNE; NaN stays NaN
                        hq[v] = __builtin_bit_cast(unsigned short, h);
                        res[v] -= (float)h;
                    }
                    if constexpr (V == 4)              // one 8-byte store per piece (c % 4 == 0: aligned)
                        *reinterpret_cast<uint2*>(q + p * 16) = make_uint2((unsigned)hq[0] | ((unsigned)hq[1] << 16), (unsigned)hq[2] | ((unsigned)hq[3] << 16));
                    else
                        q[p * 16] = hq[0];
                }
            } else {
                stv<V>(static_cast<float*>(da_) + r * C + c, o);        // (float32 tensors only: the launcher checks)
            }
          }
        }
    }
    row_lane_reduce<V, 1>(acc, red, CL, RL, cl, rl);
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < V; ++v) partial[(int64_t)blockIdx.x * C + c + v] = acc[0][v];
    }
}

// out[c] = sum_r partial[r*stride + c] for c < count (double -> float)
__global__ void finish_channel_sum_kernel(const double* __restrict__ partial, int records,
                                          int64_t stride, int count, float* __restrict__ out) {
    __shared__ double red[kBlock];
    const int cl = threadIdx.x & (kFinCh - 1), rl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    const int C = count;
    double s = 0;
    if (c < C)
        for (int r = rl; r < records; r += kFinLanes) s += partial[(int64_t)r * stride + c];
    s = finish_sum(s, red);
    if (rl == 0 && c < C) out[c] = (float)s;
}

// the same for a table of sums in ONE launch (blockIdx.y = entry): the conv-bias gradients of every layer of a backward pass,
// whose per-block partials stay where bn_bwd_apply left them until the end of the pass (nothing reads a bias gradient
// before the optimiser; 18 five-microsecond launches -- each behind a 5-us bubble -- leave the main stream's chain)
__global__ void finish_channel_sum_batched_kernel(const FinishSumDesc* __restrict__ descs) {
    __shared__ double red[kBlock];
    const FinishSumDesc d = descs[blockIdx.y];
    const int cl = threadIdx.x & (kFinCh - 1), rl = threadIdx.x / kFinCh;
    const int c = blockIdx.x * kFinCh + cl;
    if (blockIdx.x * kFinCh >= d.count) return;          // (uniform for the block)
    double s = 0;
    if (c < d.count)
        for (int r = rl; r < d.records; r += kFinLanes) s += d.partial[(int64_t)r * d.stride + c];
    s = finish_sum(s, red);
    if (rl == 0 && c < d.count) d.out[c] = (float)s;
}

template <int V>
__global__ void channel_sum_kernel(const void* __restrict__ v_, int v16, int pstride, int64_t M, int C, int CL,
                                   int64_t rows_per_block, double* __restrict__ partial) {
    __shared__ double red[V * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * V;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    double acc[1][V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[0][v] = 0;
    if (c < C) {
        constexpr int U = 8;          // eight rows per trip, loads first (one load in flight per lane: 3.2 TB/s); same order of the sums
        for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * RL) {
            float x[U][V];
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (rb + (int64_t)k * RL < r1) ldy<V>(v_, v16, (rb + (int64_t)k * RL) * pstride + c, x[k]);
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (rb + (int64_t)k * RL < r1) {
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[0][v] += (double)x[k][v];
                }
        }
    }
    row_lane_reduce<V, 1>(acc, red, CL, RL, cl, rl);
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < V; ++v) partial[(int64_t)blockIdx.x * C + c + v] = acc[0][v];
    }
}

// ------------------------------------------------------------------ pool
// SKIP / POOL: which of the two outputs this launch writes (both: one pass; the model may split them over two streams -- the
// pooled tensor feeds the next conv at once, the skip is not read before the decoder)
template <bool SKIP, bool POOL>
__global__ void bn_relu_pool_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    float* __restrict__ skip, int skip_ps, float* __restrict__ pooled, float slope) {
    const int Hp = H >> 1, Wp = W >> 1;
    const int64_t total = (int64_t)N * Hp * Wp * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int px = (int)(t % Wp);
        t /= Wp;
        const int py = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const float sc = scale[c], sh = shift[c];
        float best = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t pix = ((int64_t)n * H + (2 * py + (k >> 1))) * W + (2 * px + (k & 1));
            float a = y[pix * C + c] * sc + sh;
            a = act_f(a, slope);
            if constexpr (SKIP) skip[pix * skip_ps + c] = a;
            best = (k == 0 || a > best) ? a : best;
        }
        if constexpr (POOL) pooled[i] = best;
    }
}

// rows/cols of an odd-sized map that the floor-mode pool does not cover still need their skip
__global__ void bn_relu_edge_kernel(const float* __restrict__ y, int N, int H, int W, int C,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    float* __restrict__ skip, int skip_ps, float slope) {
    const int64_t total = (int64_t)N * H * W * C;
    const int He = H & ~1, We = W & ~1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        if (yy >= He || x >= We) {
            float a = y[pix * C + c] * scale[c] + shift[c];
            skip[pix * skip_ps + c] = act_f(a, slope);
        }
    }
}

__global__ void pool_bwd_merge_kernel(const void* __restrict__ y, int y16, int64_t yps, int N, int H, int W, int C,
                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                      const float* __restrict__ dskip, int dskip_ps,
                                      const float* __restrict__ dpool, float* __restrict__ da, float slope) {
    const int Hp = H >> 1, Wp = W >> 1;
    const int64_t total = (int64_t)N * Hp * Wp * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int px = (int)(t % Wp);
        t /= Wp;
        const int py = (int)(t % Hp);
        const int n = (int)(t / Hp);
        const float sc = scale[c], sh = shift[c];
        float best = 0.0f;
        int arg = 0;
        int64_t pixk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pixk[k] = ((int64_t)n * H + (2 * py + (k >> 1))) * W + (2 * px + (k & 1));
            float a = ldy1(y, y16, pixk[k] * yps + c) * sc + sh;
            a = act_f(a, slope);
            if (k == 0 || a > best) {
                best = a;
                arg = k;
            }
        }
        const float g = dpool[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            da[pixk[k] * C + c] = dskip[pixk[k] * dskip_ps + c] + (k == arg ? g : 0.0f);
    }
}
// Four channels per thread (16-byte accesses), optionally (SUMS) also leaving the BatchNorm-backward sums of the layer
// whose activated output da is the gradient of (sum dz, sum dz * xhat per channel; what bn_bwd_reduce would re-read da
// and y for): the grid is then chosen so that a thread keeps ONE channel group over its grid-stride iterations; fp64 per
// thread, fixed-order LDS fold per block, one record per block (C / 4 <= 256) or per group of C / 1024 blocks
template <bool SUMS>
__global__ __launch_bounds__(256) void pool_bwd_merge_vec_kernel(
    const void* __restrict__ y, int y16, int64_t yps, int N, int H, int W, int C, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const void* __restrict__ dskip, int dskip16, int dskip_ps, const void* __restrict__ dpool, int dp16, float* __restrict__ da, float slope,
    double* __restrict__ records, unsigned short* __restrict__ da16) {
    // da16 != null: the merged gradient is stored as bfloat16 there (da unused) and the sums are those of the stored values
    __shared__ double red[SUMS ? 8 * kBlock : 1];
    const int Hp = H >> 1, Wp = W >> 1, C4 = C >> 2;
    const int64_t total = (int64_t)N * Hp * Wp * C4;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int64_t i = first; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;            // (SUMS: the same for every iteration of this thread)
        int64_t t = i / C4;
        const int px = (int)(t % Wp);
        t /= Wp;
        const int py = (int)(t % Hp);
        const int n = (int)(t / Hp);
        float sc[4], sh[4], mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
        ldv<4>(scale + c, sc); ldv<4>(shift + c, sh);
        if (SUMS) { ldv<4>(mean + c, mu); ldv<4>(invstd + c, is); }
        float best[4], yv[4][4], zv[4][4];
        int arg[4] = {0, 0, 0, 0};
        int64_t pixk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pixk[k] = ((int64_t)n * H + (2 * py + (k >> 1))) * W + (2 * px + (k & 1));
            ldy<4>(y, y16, pixk[k] * yps + c, yv[k]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                zv[k][v] = yv[k][v] * sc[v] + sh[v];
                const float a = act_f(zv[k][v], slope);
                if (k == 0 || a > best[v]) {
                    best[v] = a;
                    arg[v] = k;
                }
            }
        }
        float g[4];
        ldy<4>(dpool, dp16, i * 4, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d[4];
            ldy<4>(dskip, dskip16, pixk[k] * dskip_ps + c, d);
#pragma unroll
            for (int v = 0; v < 4; ++v) d[v] += (k == arg[v] ? g[v] : 0.0f);
            if (da16) round_store_bf16x4(da16 + pixk[k] * C + c, d);
            else stv<4>(da + pixk[k] * C + c, d);
            if (SUMS) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float dz = dact_f(zv[k][v], d[v], slope);
                    const float xh = (yv[k][v] - mu[v]) * is[v];
                    s1[v] += (double)dz;
                    s2[v] += (double)dz * (double)xh;
                }
            }
        }
    }
    if constexpr (SUMS) {
        const int c = (int)(first % C4) * 4;
        if (C4 <= kBlock) {                         // 256 % C4 == 0: threads g, g + C4, ... hold channel group g
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                red[(2 * v) * kBlock + threadIdx.x] = s1[v];
                red[(2 * v + 1) * kBlock + threadIdx.x] = s2[v];
            }
            __syncthreads();
            if ((int)threadIdx.x < C4) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double t1 = 0.0, t2 = 0.0;
                    for (int k = threadIdx.x; k < kBlock; k += C4) {
                        t1 += red[(2 * v) * kBlock + k];
                        t2 += red[(2 * v + 1) * kBlock + k];
                    }
                    records[((int64_t)blockIdx.x * C + c + v) * 2 + 0] = t1;
                    records[((int64_t)blockIdx.x * C + c + v) * 2 + 1] = t2;
                }
            }
        } else {                                    // C4 % 256 == 0: C4 / 256 consecutive blocks make one record
            const int per = C4 / kBlock;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                records[((int64_t)(blockIdx.x / per) * C + c + v) * 2 + 0] = s1[v];
                records[((int64_t)(blockIdx.x / per) * C + c + v) * 2 + 1] = s2[v];
            }
        }
    }
}

__global__ void copy_edge_kernel(int N, int H, int W, int C, const float* __restrict__ dskip,
                                 int dskip_ps, float* __restrict__ da) {
    const int64_t total = (int64_t)N * H * W * C;
    const int He = H & ~1, We = W & ~1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int x = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        if (yy >= He || x >= We) da[pix * C + c] = dskip[pix * dskip_ps + c];
    }
}

// ------------------------------------------------------------------ head (1x1 conv) + loss
__global__ void head_fwd_kernel(const float* __restrict__ y, int64_t M, int C,
                                const float* __restrict__ scale, const float* __restrict__ shift,
                                const float* __restrict__ w, const float* __restrict__ b, int Cout,
                                float* __restrict__ logits, float slope) {
    // one wave-quarter (16 lanes) per pixel: lanes stride over channels, shuffle-reduce
    const int lane16 = threadIdx.x & 15;
    const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (m >= M) return;   // whole 16-lane group leaves together
    for (int o = 0; o < Cout; ++o) {
        float acc = 0.0f;
        for (int c = lane16; c < C; c += 16) {
            float a = y[m * C + c] * scale[c] + shift[c];
            a = act_f(a, slope);
            acc += a * w[o * C + c];
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 16);
        if (lane16 == 0) logits[m * Cout + o] = acc + b[o];
    }
}

// C % 4 == 0: L lanes per pixel (L a power of two <= 16), each lane reads 16-byte groups of channels
template <int L>
__global__ __launch_bounds__(256) void head_fwd_vec_kernel(const void* __restrict__ y, int y16, int64_t yps, int64_t M, int C,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ w, const float* __restrict__ b, int Cout,
                                                           float* __restrict__ logits, float slope) {
    const int lane = threadIdx.x & (L - 1);
    const int64_t m = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / L;
    if (m >= M) return;   // whole L-lane group leaves together
    const int C4 = C >> 2;
    for (int o = 0; o < Cout; ++o) {
        float acc = 0.0f;
        for (int c4 = lane; c4 < C4; c4 += L) {
            float yv[4], sc[4], sh[4], wv[4];
            ldy<4>(y, y16, m * yps + 4 * c4, yv); ldv<4>(scale + 4 * c4, sc); ldv<4>(shift + 4 * c4, sh);
            ldv<4>(w + (int64_t)o * C + 4 * c4, wv);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc += act_f(yv[v] * sc[v] + sh[v], slope) * wv[v];
        }
#pragma unroll
        for (int off = L / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, L);
        if (lane == 0) logits[m * Cout + o] = acc + b[o];
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void loss_reduce_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                   int64_t count, double* __restrict__ partial) {
    __shared__ double red[4][kBlock / 64];
    double s[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float x = logits[i];
        const float t = labels[i] ? 1.0f : 0.0f;
        const float bce = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
        const float p = sigmoidf_(x);
        s[0] += (double)bce;
        s[1] += (double)(p * t);
        s[2] += (double)p;
        s[3] += (double)t;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = wave_sum(s[k]);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0;
        for (int wv = 0; wv < kBlock / 64; ++wv) v += red[threadIdx.x][wv];
        partial[(int64_t)blockIdx.x * 4 + threadIdx.x] = v;
    }
}

__global__ void loss_finish_kernel(const double* __restrict__ partial, int blocks, double count,
                                   double* __restrict__ sums4, float* __restrict__ loss_out) {
    __shared__ double red[4][kBlock / 64];
    double s[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < blocks; b += blockDim.x)
        for (int k = 0; k < 4; ++k) s[k] += partial[(int64_t)b * 4 + k];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = wave_sum(s[k]);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int k = 0; k < 4; ++k) {
            t[k] = 0;
            for (int wv = 0; wv < kBlock / 64; ++wv) t[k] += red[k][wv];
            sums4[k] = t[k];
        }
        // float32 arithmetic for the final combination, as the reference's fp32 tensors do
        const float bce = (float)(t[0] / count);
        const float inter = (float)t[1], sp = (float)t[2], st = (float)t[3];
        const float dice = 1.0f - (2.0f * inter + 1.0f) / (sp + st + 1.0f);
        loss_out[0] = bce + dice;
    }
}

__global__ void loss_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                int64_t count, const double* __restrict__ sums4,
                                float* __restrict__ dlogits) {
    const float inter = (float)sums4[1], sp = (float)sums4[2], st = (float)sums4[3];
    const float D = sp + st + 1.0f;
    const float num = 2.0f * inter + 1.0f;
    const float invD2 = 1.0f / (D * D);
    const float invN = (float)(1.0 / (double)count);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float x = logits[i];
        const float t = labels[i] ? 1.0f : 0.0f;
        const float p = sigmoidf_(x);
        // d dice / d p_i = -(2 t D - num) / D^2
        const float ddice = -(2.0f * t * D - num) * invD2;
        dlogits[i] = (p - t) * invN + ddice * p * (1.0f - p);
    }
}

// ---- focal loss (NOT in the reference, which trains with BCE + dice only).  Builder-
// defined as the published sigmoid focal loss (Lin et al. 2017; torchvision.ops.sigmoid_focal_loss, mean):
//   p = sigmoid(x), p_t = p t + (1-p)(1-t), FL = alpha_t (1 - p_t)^gamma * BCEwithLogits(x, t), mean over elements
__device__ __forceinline__ float focal_elem(float x, float t, float alpha, float gamma) {
    const float p = sigmoidf_(x);
    const float ce = fmaxf(x, 0.0f) - x * t + log1pf(expf(-fabsf(x)));
    const float pt = p * t + (1.0f - p) * (1.0f - t);
    const float at = alpha >= 0.0f ? alpha * t + (1.0f - alpha) * (1.0f - t) : 1.0f;
    return at * powf(1.0f - pt, gamma) * ce;
}
__global__ void focal_reduce_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                    int64_t count, float alpha, float gamma, double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    double s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)focal_elem(logits[i], labels[i] ? 1.0f : 0.0f, alpha, gamma);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < kBlock / 64; ++k) t += red[k];
        partial[blockIdx.x] = t;
    }
}
__global__ void focal_finish_kernel(const double* __restrict__ partial, int blocks, double count,
                                    float* __restrict__ loss_out) {
    __shared__ double red[kBlock / 64];
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += blockDim.x) s += partial[b];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < kBlock / 64; ++k) t += red[k];
        loss_out[0] = (float)(t / count);
    }
}
// d FL / d x (closed form), times 1/N
__global__ void focal_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels, int64_t count,
                                 float alpha, float gamma, float* __restrict__ dlogits) {
    const float invN = (float)(1.0 / (double)count);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = logits[i];
        const bool pos = labels[i] != 0;
        const float p = sigmoidf_(x);
        // log p and log(1-p) without cancellation: -softplus(-x), -softplus(x)
        const float sp_neg = fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));     // softplus(-x) = -log p
        const float sp_pos = fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));      // softplus(x)  = -log(1-p)
        float g;
        if (pos) {       // FL = -a (1-p)^g log p
            const float a = alpha >= 0.0f ? alpha : 1.0f;
            g = a * powf(1.0f - p, gamma) * (gamma * p * (-sp_neg) - (1.0f - p));
        } else {         // FL = -(1-a) p^g log(1-p)
            const float a = alpha >= 0.0f ? 1.0f - alpha : 1.0f;
            g = a * powf(p, gamma) * (p - gamma * (1.0f - p) * (-sp_pos));
        }
        dlogits[i] = g * invN;
    }
}

// UNetOverfit's head (models/unet.py:196): the model RETURNS sigmoid(logits), and train_model.py:120 still
// feeds that to BCE-with-logits + dice -- the loss kernels then see x = sigmoid(z) as their "logits"
__global__ void sigmoid_fwd_kernel(const float* __restrict__ z, int64_t n, float* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        x[i] = sigmoidf_(z[i]);
}
// dz = dx * x (1 - x), in place on dx
__global__ void sigmoid_bwd_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = x[i];
        d[i] = d[i] * (p * (1.0f - p));
    }
}

// da[m][c] = sum_o dl[m][o] * w[o][c];  partial dw/db per block (double)
__global__ void head_bwd_kernel(const float* __restrict__ y, int64_t M, int C, int CL,
                                int64_t rows_per_block, const float* __restrict__ scale,
                                const float* __restrict__ shift, const float* __restrict__ w, int Cout,
                                const float* __restrict__ dl, float* __restrict__ da,
                                double* __restrict__ partial, float slope) {
    // partial layout per row block: [Cout][C] dw then [Cout] db  (stride Cout*C + Cout)
    __shared__ double red[kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = blockIdx.y * CL + cl;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const int64_t pstride = (int64_t)Cout * C + Cout;
    for (int o = 0; o < Cout; ++o) {
        double sw = 0, sb = 0;
        if (c < C) {
            const float sc = scale[c], sh = shift[c], wv = w[o * C + c];
            for (int64_t r = r0 + rl; r < r1; r += RL) {
                const float d = dl[r * Cout + o];
                float a = y[r * C + c] * sc + sh;
                a = act_f(a, slope);
                sw += (double)d * (double)a;
                sb += (double)d;
                if (o == 0) da[r * C + c] = d * wv;
                else da[r * C + c] += d * wv;
            }
        }
        red[threadIdx.x] = sw;
        __syncthreads();
        if (rl == 0 && c < C) {
            for (int k = 1; k < RL; ++k) sw += red[k * CL + cl];
            partial[(int64_t)blockIdx.x * pstride + (int64_t)o * C + c] = sw;
        }
        __syncthreads();
        if (blockIdx.y == 0 && cl == 0) {   // db: channel lane 0 of the first channel block
            red[rl] = sb;
        }
        __syncthreads();
        if (blockIdx.y == 0 && threadIdx.x == 0) {
            double t = 0;
            for (int k = 0; k < RL; ++k) t += red[k];
            partial[(int64_t)blockIdx.x * pstride + (int64_t)Cout * C + o] = t;
        }
        __syncthreads();
    }
}

// the same with four channels per lane (C % 4 == 0): 16-byte loads of y and stores of da
__global__ __launch_bounds__(256) void head_bwd_vec_kernel(const void* __restrict__ y, int y16, int64_t yps, int64_t M, int C, int CL,
                                                           int64_t rows_per_block, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ w, int Cout,
                                                           const float* __restrict__ dl, float* __restrict__ da,
                                                           double* __restrict__ partial, float slope,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           double* __restrict__ bn_records, unsigned short* __restrict__ da16,
                                                           int store_da) {
    // store_da == 0 (Cout == 1, bn_records != null): da is not written -- bn_bwd_apply's head form recomputes d * w from
    // the logit gradients (1 float per pixel) instead of reading C floats per pixel back
    // da16 != null (Cout == 1): da is stored as bfloat16 there and the BatchNorm-backward sums are those of the stored values
    // bn_records != null (Cout == 1): also the BatchNorm-backward sums of the layer below (sum dz, sum dz * xhat with
    // dz = da * act'), one fp64 record per row block -- bn_bwd_reduce's pass over da and y disappears
    __shared__ double red[4 * kBlock];
    const int RL = kBlock / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    const int c = (blockIdx.y * CL + cl) * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    const int64_t pstride = (int64_t)Cout * C + Cout;
    double q1[4] = {0, 0, 0, 0}, q2[4] = {0, 0, 0, 0};
    for (int o = 0; o < Cout; ++o) {
        double sw[4] = {0, 0, 0, 0}, sb = 0;
        if (c < C) {
            float sc[4], sh[4], wv[4], mu[4] = {0, 0, 0, 0}, is[4] = {0, 0, 0, 0};
            ldv<4>(scale + c, sc); ldv<4>(shift + c, sh); ldv<4>(w + (int64_t)o * C + c, wv);
            if (bn_records) { ldv<4>(mean + c, mu); ldv<4>(invstd + c, is); }
            constexpr int U = 4;      // four rows per trip, loads first (same order of the sums)
            for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)U * RL) {
              float du[U], yu[U][4], gu[U][4];
#pragma unroll
              for (int k = 0; k < U; ++k) {
                  const int64_t r = rb + (int64_t)k * RL;
                  if (r < r1) {
                      du[k] = dl[r * Cout + o];
                      ldy<4>(y, y16, r * yps + c, yu[k]);
                      if (o != 0) ldv<4>(da + r * C + c, gu[k]);
                  }
              }
#pragma unroll
              for (int k = 0; k < U; ++k) {
                const int64_t r = rb + (int64_t)k * RL;
                if (r >= r1) break;
                const float d = du[k];
                float yv[4], g[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) { yv[v] = yu[k][v]; g[v] = gu[k][v]; }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float z = yv[v] * sc[v] + sh[v];
                    const float a = act_f(z, slope);
                    sw[v] += (double)d * (double)a;
                    g[v] = o == 0 ? d * wv[v] : g[v] + d * wv[v];
                }
                if (da16) round_store_bf16x4(da16 + r * C + c, g);
                else if (store_da) stv<4>(da + r * C + c, g);
                if (bn_records) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float dz = dact_f(yv[v] * sc[v] + sh[v], g[v], slope);
                        const float xh = (yv[v] - mu[v]) * is[v];
                        q1[v] += (double)dz;
                        q2[v] += (double)dz * (double)xh;
                    }
                }
                sb += (double)d;
              }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) red[v * kBlock + threadIdx.x] = sw[v];
        __syncthreads();
        if (rl == 0 && c < C) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double t = sw[v];
                for (int k = 1; k < RL; ++k) t += red[v * kBlock + k * CL + cl];
                partial[(int64_t)blockIdx.x * pstride + (int64_t)o * C + c + v] = t;
            }
        }
        __syncthreads();
        if (blockIdx.y == 0 && cl == 0) red[rl] = sb;   // db: channel lane 0 of the first channel block
        __syncthreads();
        if (blockIdx.y == 0 && threadIdx.x == 0) {
            double t = 0;
            for (int k = 0; k < RL; ++k) t += red[k];
            partial[(int64_t)blockIdx.x * pstride + (int64_t)Cout * C + o] = t;
        }
        __syncthreads();
    }
    if (bn_records) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int v = 0; v < 4; ++v) red[v * kBlock + threadIdx.x] = q == 0 ? q1[v] : q2[v];
            __syncthreads();
            if (rl == 0 && c < C) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    double t = q == 0 ? q1[v] : q2[v];
                    for (int k = 1; k < RL; ++k) t += red[v * kBlock + k * CL + cl];
                    bn_records[((int64_t)blockIdx.x * C + c + v) * 2 + q] = t;
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ layouts
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int N, int C, int H, int W,
                                    float* __restrict__ dst) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int x = (int)(t % W);
        t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        dst[i] = src[(((int64_t)n * C + c) * H + yy) * W + x];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int N, int C, int H, int W,
                                    float* __restrict__ dst) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        int64_t t = i / W;
        const int yy = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        dst[i] = src[(((int64_t)n * H + yy) * W + x) * C + c];
    }
}
__global__ void weight_to_dgrad_kernel(const float* __restrict__ wf, int taps, int Cout, int Cin,
                                       int flip, float* __restrict__ wd) {
    const int64_t total = (int64_t)taps * Cout * Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cout);
        int64_t t = i / Cout;
        const int ci = (int)(t % Cin);
        const int tp = (int)(t / Cin);
        const int ts = flip ? (taps - 1 - tp) : tp;
        wd[i] = wf[((int64_t)ts * Cout + co) * Cin + ci];
    }
}
// all layers in one launch: one 32 x 32 tile of one tap's [cout][cin] matrix per block, transposed through LDS so
// that both the reads (cin contiguous) and the writes (cout contiguous) are coalesced; the block finds its layer in
// the tile prefix of the descriptor table
__global__ __launch_bounds__(256) void weight_to_dgrad_batched_kernel(const RelayoutDesc* __restrict__ descs, int n,
                                                                     const float* __restrict__ src, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.x;
    int l = 0;
    while (l + 1 < n && descs[l + 1].tile0 <= b) ++l;
    const RelayoutDesc d = descs[l];
    const float* __restrict__ wf = src + d.src_off;
    float* __restrict__ wd = dst + d.dst_off;
    const int tci = (d.cin + 31) / 32, tco = (d.cout + 31) / 32;
    const int t = (int)(b - d.tile0);
    const int tp = t / (tci * tco), r = t % (tci * tco);
    const int bco = r / tci, bci = r % tci;
    const int ts = d.flip ? (d.taps - 1 - tp) : tp;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = bco * 32 + ty + 8 * k, ci = bci * 32 + tx;
        if (co < d.cout && ci < d.cin) tile[ty + 8 * k][tx] = wf[((int64_t)ts * d.cout + co) * d.cin + ci];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = bci * 32 + ty + 8 * k, co = bco * 32 + tx;
        if (co < d.cout && ci < d.cin) wd[((int64_t)tp * d.cin + ci) * d.cout + co] = tile[tx][ty + 8 * k];
    }
}
__global__ void relu_bwd_kernel(float4* __restrict__ dA, const float4* __restrict__ Y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 g = dA[i];
        const float4 y = Y[i];
        g.x = y.x > 0.0f ? g.x : 0.0f;
        g.y = y.y > 0.0f ? g.y : 0.0f;
        g.z = y.z > 0.0f ? g.z : 0.0f;
        g.w = y.w > 0.0f ? g.w : 0.0f;
        dA[i] = g;
    }
}
__global__ void pad_channels_kernel(const float* __restrict__ s, int64_t M, int c, int cp,
                                    float* __restrict__ d) {
    const int64_t total = M * cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % cp);
        d[i] = k < c ? s[(i / cp) * c + k] : 0.0f;
    }
}
__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ s, int64_t n, float* __restrict__ d) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        d[i] = (float)s[i];
}

// ------------------------------------------------------------------ optimiser
__global__ void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    double s = 0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += (double)v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < kBlock / 64; ++k) t += red[k];
        partial[blockIdx.x] = t;
    }
}
__global__ void sumsq_finish_kernel(const double* __restrict__ partial, int blocks,
                                    double* __restrict__ out) {
    __shared__ double red[kBlock / 64];
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += blockDim.x) s += partial[b];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int k = 0; k < kBlock / 64; ++k) t += red[k];
        out[0] = t;
    }
}

// One Adam step with torch.optim.Adam's CPU (single-tensor) rounding sequence, which the golden
// trajectories pin bit-for-bit on m and v: g' = fma(wd, p, g); m = lerp(m, g', 1-b1) = fma(1-b1, g'-m, m);
// v = fma((1-b2) g', g', b2 v); p += (-(lr/bc1) m) / (sqrt(v)/sqrt(bc2) + eps).  The scalars are
// formed in double on the host (python floats in the reference) and rounded to float once.
__global__ void adam_kernel(AdamArgs a) {
    const float norm = (float)sqrt(a.sumsq[0]) * a.grad_scale;
    float coef = a.max_norm / (norm + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (a.max_norm <= 0.0f) coef = 1.0f;   // clipping disabled
    const float gs = a.grad_scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float p = a.p[i];
        float g = (a.g[i] * gs) * coef;
        g = __fmaf_rn(a.wd, p, g);
        const float m0 = a.m[i];
        const float m = __fmaf_rn(a.one_minus_beta1, g - m0, m0);
        const float v = __fmaf_rn(a.one_minus_beta2 * g, g, a.beta2 * a.v[i]);
        a.m[i] = m;
        a.v[i] = v;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.p[i] = p + (a.neg_step * m) / denom;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.norm_out) a.norm_out[0] = norm;
}

// out[i] = sum_{k<count} slabs[(base + k*kstride) * n + i]; grid.y selects the group:
// base = blockIdx.y * group, out row = blockIdx.y * out_rowstride (in units of n)
__global__ void scale_inplace_kernel(float* __restrict__ x, int64_t n, float f) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= f;
}
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int nslabs, int group, int kstride,
                                    int64_t n, float* __restrict__ out, int64_t out_rowstride) {
    const int base = blockIdx.y * group;
    int cnt = nslabs - base;
    if (cnt > group) cnt = group;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < cnt; ++k) s += slabs[((int64_t)base + (int64_t)k) * kstride * n + i];
        out[(int64_t)blockIdx.y * out_rowstride * n + i] = s;
    }
}

// the same sums four elements per thread (16-byte loads, the k loop unrolled so that eight loads are in flight): the
// scalar form reached 1.4-1.7 TB/s of its slab reads (0.52 ms of the float32 U-Net step, round 3).  Element order of every
// sum unchanged (k ascending): bit-identical results
__global__ __launch_bounds__(256) void reduce_slabs_vec4_kernel(const float* __restrict__ slabs, int nslabs, int group, int kstride,
                                                               int64_t n4, float* __restrict__ out, int64_t out_rowstride) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int base = blockIdx.y * group;
    int cnt = nslabs - base;
    if (cnt > group) cnt = group;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(slabs) + (int64_t)base * kstride * n4;
    f32x4* __restrict__ dst = reinterpret_cast<f32x4*>(out) + (int64_t)blockIdx.y * out_rowstride * n4;
    const int64_t step = (int64_t)kstride * n4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 8 <= cnt; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + (int64_t)(k + u) * step + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < cnt; ++k) s += __builtin_nontemporal_load(src + (int64_t)k * step + i);
        dst[i] = s;
    }
}

static int grid_for(int64_t total, int cap = 256 * 16) {
    int64_t b = cdiv(total, kBlock);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

constexpr int kMaxRowBlocks = 1024;

}  // namespace

// ====================================================================== launch wrappers
size_t bn_stats_ws_floats(int C) { return (size_t)kMaxRowBlocks * C * 2 * 2; }

// records actually used for (M, C)
static ChanGeom geom_rows(int64_t M, int C, bool allow_vec = true) {
    return chan_geom(M, C, kMaxRowBlocks, allow_vec);
}

void launch_bn_stats(rfi_ctx* ctx, const float* y, int64_t M, int C, float* partial_ws) {
    ChanGeom g = geom_rows(M, C);
    ProfScope ps(ctx, FAM_BN, 0, (double)M * C * 4);
    if (g.V == 4)
        hipLaunchKernelGGL(bn_stats_kernel<4>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, y, M,
                           C, g.CL, g.rows_per_block, reinterpret_cast<double*>(partial_ws));
    else
        hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, y, M,
                           C, g.CL, g.rows_per_block, reinterpret_cast<double*>(partial_ws));
    check_launch("bn_stats");
}

void launch_bn_finalize(rfi_ctx* ctx, const float* partial, int64_t M, int C, const float* gamma,
                          const float* beta, float* running_mean, float* running_var,
                          int ema_repeats, float* mean, float* invstd, float* scale, float* shift,
                          float* var_out, int records, hipEvent_t done) {
    ChanGeom g = geom_rows(M, C);
    ProfScope ps(ctx, FAM_BN);
    hipExtLaunchKernelGGL(bn_finalize_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0, ctx->stream, nullptr, done, 0,
                          reinterpret_cast<const double*>(partial), records > 0 ? records : g.rblocks, C, (double)M, gamma, beta,
                          running_mean, running_var, ema_repeats, mean, invstd, scale, shift, var_out);
    check_launch("bn_finalize");
}

void launch_bn_eval_coeffs(rfi_ctx* ctx, int C, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, float* scale,
                           float* shift) {
    ProfScope ps(ctx, FAM_BN);
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((int)cdiv(C, 256)), dim3(256), 0, ctx->stream, C,
                       gamma, beta, running_mean, running_var, scale, shift);
    check_launch("bn_eval_coeffs");
}

size_t bn_bwd_ws_floats(int64_t M, int C) {
    (void)M;
    return (size_t)kMaxRowBlocks * C * 2 * 2;   // doubles counted as 2 floats
}

void launch_bn_bwd_finalize_records(rfi_ctx* ctx, const float* partial_ws, int records, int64_t M, int C, float* c1,
                                    float* c2, float* dgamma, float* dbeta) {
    ProfScope ps(ctx, FAM_BN);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0, ctx->stream,
                       reinterpret_cast<const double*>(partial_ws), records, C, (double)M, c1, c2, dgamma, dbeta);
    check_launch("bn_bwd_finalize");
}

void launch_bn_bwd_reduce(rfi_ctx* ctx, YRef da, YRef y, int64_t M, int C,
                          const float* scale, const float* shift, const float* mean,
                          const float* invstd, float* partial_ws, float* c1, float* c2,
                          float* dgamma, float* dbeta, float slope) {
    ChanGeom g = geom_rows(M, C);
    static const bool no16 = getenv("RFI_NO_BN16") != nullptr;          // A/B runs: the generic kernels for bfloat16 tensors too
    const bool fast16 = !no16 && da.bf16 && y.bf16 && C % 8 == 0 && da.stride(C) == C && y.stride(C) % 8 == 0 &&
                        !((reinterpret_cast<uintptr_t>(da.p) | reinterpret_cast<uintptr_t>(y.p) | reinterpret_cast<uintptr_t>(scale) |
                           reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(invstd)) & 15);
    if (fast16) {
        g = chan_geom(M, C, kMaxRowBlocks, true, 8);
        ProfScope ps(ctx, FAM_BN, 0, (double)M * C * 4);
        hipLaunchKernelGGL(bn_bwd_reduce16_kernel, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream,
                           static_cast<const unsigned short*>(da.p), static_cast<const unsigned short*>(y.p), y.stride(C), M, C, g.CL,
                           g.rows_per_block, scale, shift, mean, invstd, reinterpret_cast<double*>(partial_ws), slope);
        check_launch("bn_bwd_reduce16");
    } else {
        RFI_REQUIRE(da.stride(C) == C && (!da.bf16 || g.V == 4), "bn_bwd_reduce: the gradient tensor must be dense (bfloat16: C % 4 == 0)");
        ProfScope ps(ctx, FAM_BN, 0, (double)M * C * ((y.bf16 ? 2 : 4) + (da.bf16 ? 2 : 4)));
        if (g.V == 4)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0,
                               ctx->stream, da.p, da.bf16, y.p, y.bf16, y.stride(C), M, C, g.CL, g.rows_per_block, scale, shift, mean, invstd,
                               reinterpret_cast<double*>(partial_ws), slope);
        else
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0,
                               ctx->stream, da.p, da.bf16, y.p, y.bf16, y.stride(C), M, C, g.CL, g.rows_per_block, scale, shift, mean, invstd,
                               reinterpret_cast<double*>(partial_ws), slope);
        check_launch("bn_bwd_reduce");
    }
    {
        ProfScope ps(ctx, FAM_BN);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0, ctx->stream,
                           reinterpret_cast<const double*>(partial_ws), g.rblocks, C, (double)M, c1, c2,
                           dgamma, dbeta);
        check_launch("bn_bwd_finalize");
    }
}

void launch_bn_bwd_apply(rfi_ctx* ctx, YRef da_inout, YRef y, int64_t M, int C,
                         const float* scale, const float* shift, const float* mean,
                         const float* invstd, const float* gamma, const float* c1, const float* c2,
                         float* partial_ws, float* dbias, float slope, unsigned short* planes_out,
                         int64_t planes_pstride, int planes_P, hipEvent_t done, bool finish_dbias, const float* head_dl,
                         const float* head_w) {
    ChanGeom g = geom_rows(M, C);
    static const bool no16 = getenv("RFI_NO_BN16") != nullptr;
    const bool fast16 = !no16 && da_inout.bf16 && y.bf16 && planes_out && planes_P == 1 && !head_dl && C % 8 == 0 && da_inout.stride(C) == C &&
                        y.stride(C) % 8 == 0 && planes_pstride % 8 == 0 &&
                        !((reinterpret_cast<uintptr_t>(da_inout.p) | reinterpret_cast<uintptr_t>(y.p) | reinterpret_cast<uintptr_t>(planes_out) |
                           reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(mean) |
                           reinterpret_cast<uintptr_t>(invstd) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(c1) |
                           reinterpret_cast<uintptr_t>(c2)) & 15);
    if (fast16) {
        const ChanGeom g4 = g;
        g = chan_geom(M, C, kMaxRowBlocks, true, 8);
        if (dbias) {          // the row blocks of the generic kernel: the records the batched finisher's table expects (bn_bwd_apply_records)
            g.rows_per_block = g4.rows_per_block;     // -- and the same partial sums whether the finishing is deferred or not (a step with the
            g.rblocks = g4.rblocks;                   // gradient exchange finishes at once: bit-identical bias gradients either way)
        }
        {
            ProfScope ps(ctx, FAM_BN, 0, (double)M * C * 6);
            auto launch = [&](auto kernel) {
                hipExtLaunchKernelGGL(kernel, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, nullptr, done, 0,
                                      static_cast<const unsigned short*>(da_inout.p), static_cast<const unsigned short*>(y.p), y.stride(C), M, C, g.CL,
                                      g.rows_per_block, scale, shift, mean, invstd, gamma, c1, c2, reinterpret_cast<double*>(partial_ws), slope,
                                      planes_out, planes_pstride);
            };
            if (dbias) launch(bn_bwd_apply16_kernel<true>);
            else launch(bn_bwd_apply16_kernel<false>);
            check_launch("bn_bwd_apply16");
        }
        if (dbias && finish_dbias) {
            ProfScope ps(ctx, FAM_BN);
            hipLaunchKernelGGL(finish_channel_sum_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0, ctx->stream,
                               reinterpret_cast<const double*>(partial_ws), g.rblocks, (int64_t)C, C, dbias);
            check_launch("finish_channel_sum");
        }
        return;
    }
    {
        RFI_REQUIRE(da_inout.stride(C) == C && (!da_inout.bf16 || (planes_out && g.V == 4)),
                    "bn_bwd_apply: a bfloat16 gradient tensor needs the plane output and C % 4 == 0");
        ProfScope ps(ctx, FAM_BN, 0, (double)M * C * ((planes_out ? 8 + 2 * planes_P : 12) - (y.bf16 ? 2 : 0) - (da_inout.bf16 ? 2 : 0)));
        auto launch = [&](auto kernel) {
            hipExtLaunchKernelGGL(kernel, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, nullptr, done, 0,
                                  const_cast<void*>(da_inout.p), da_inout.bf16, y.p, y.bf16, y.stride(C), M, C, g.CL, g.rows_per_block, scale, shift, mean, invstd, gamma,
                                  c1, c2, reinterpret_cast<double*>(partial_ws), slope, planes_out, planes_pstride, planes_P, head_dl, head_w);
        };
        if (g.V == 4) launch(bn_bwd_apply_kernel<4>);
        else launch(bn_bwd_apply_kernel<1>);
        check_launch("bn_bwd_apply");
    }
    if (dbias && finish_dbias) launch_bn_bwd_apply_finish(ctx, partial_ws, M, C, dbias);
}
void launch_bn_bwd_apply_finish(rfi_ctx* ctx, const float* partial_ws, int64_t M, int C, float* dbias) {
    ChanGeom g = geom_rows(M, C);
    ProfScope ps(ctx, FAM_BN);
    hipLaunchKernelGGL(finish_channel_sum_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0,
                       ctx->stream, reinterpret_cast<const double*>(partial_ws), g.rblocks, (int64_t)C, C, dbias);
    check_launch("finish_channel_sum");
}

int bn_bwd_apply_records(int64_t M, int C) { return geom_rows(M, C).rblocks; }
void launch_finish_channel_sums_batched(rfi_ctx* ctx, const FinishSumDesc* descs_dev, int n, int max_count) {
    if (n <= 0) return;
    ProfScope ps(ctx, FAM_BN);
    hipLaunchKernelGGL(finish_channel_sum_batched_kernel, dim3((int)cdiv(max_count, kFinCh), n), dim3(kBlock), 0, ctx->stream, descs_dev);
    check_launch("finish_channel_sum_batched");
}

size_t channel_sum_ws_floats(int64_t M, int C) {
    (void)M;
    return (size_t)kMaxRowBlocks * C * 2;
}
void launch_channel_sum(rfi_ctx* ctx, YRef v, int64_t M, int C, float* partial_ws, float* out, bool finish) {
    ChanGeom g = geom_rows(M, C);
    const int pstride = (int)v.stride(C);
    const bool vec = g.V == 4 && pstride % 4 == 0 && (reinterpret_cast<uintptr_t>(v.p) & (v.bf16 ? 7 : 15)) == 0;
    RFI_REQUIRE(finish || vec, "channel_sum: deferred finishing needs the vector path (C % 4 == 0, an aligned view)");
    RFI_REQUIRE(vec || !v.bf16, "channel_sum: a bfloat16 view needs C % 4 == 0 and 8-byte aligned rows");
    if (!vec && g.V == 4) g = geom_rows(M, C, false);   // unaligned view: scalar lanes
    {
        ProfScope ps(ctx, FAM_REDUCE, 0, (double)M * C * (v.bf16 ? 2 : 4));
        if (g.V == 4)
            hipLaunchKernelGGL(channel_sum_kernel<4>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream,
                               v.p, v.bf16, pstride, M, C, g.CL, g.rows_per_block,
                               reinterpret_cast<double*>(partial_ws));
        else
            hipLaunchKernelGGL(channel_sum_kernel<1>, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream,
                           v.p, v.bf16, pstride, M, C, g.CL, g.rows_per_block,
                           reinterpret_cast<double*>(partial_ws));
        check_launch("channel_sum");
    }
    if (finish) {
        ProfScope ps(ctx, FAM_REDUCE);
        hipLaunchKernelGGL(finish_channel_sum_kernel, dim3((int)cdiv(C, kFinCh)), dim3(kBlock), 0,
                           ctx->stream, reinterpret_cast<const double*>(partial_ws), g.rblocks, (int64_t)C, C, out);
        check_launch("finish_channel_sum");
    }
}
void launch_channel_sum(rfi_ctx* ctx, View v, int64_t M, int C, float* partial_ws, float* out, bool finish) {
    launch_channel_sum(ctx, YRef(v.p, (int64_t)v.pstride), M, C, partial_ws, out, finish);
}

void launch_bn_relu_pool(rfi_ctx* ctx, const float* y, int N, int H, int W, int C,
                         const float* scale, const float* shift, MutView skip, float* pooled, float slope) {
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    RFI_REQUIRE(skip.p || pooled, "bn_relu_pool: nothing to write");
    {
        ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * (skip.p ? 8 : 4) + (pooled ? (double)total * 4 : 0.0));
        if (skip.p && pooled)
            hipLaunchKernelGGL((bn_relu_pool_kernel<true, true>), dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, y,
                               N, H, W, C, scale, shift, skip.p, skip.pstride, pooled, slope);
        else if (pooled)      // (skip null: the pooled tensor only)
            hipLaunchKernelGGL((bn_relu_pool_kernel<false, true>), dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, y,
                               N, H, W, C, scale, shift, nullptr, 0, pooled, slope);
        else                  // (pooled null: the skip only)
            hipLaunchKernelGGL((bn_relu_pool_kernel<true, false>), dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, y,
                               N, H, W, C, scale, shift, skip.p, skip.pstride, nullptr, slope);
        check_launch("bn_relu_pool");
    }
    if (skip.p && ((H & 1) || (W & 1))) {
        ProfScope ps(ctx, FAM_ELEMWISE);
        hipLaunchKernelGGL(bn_relu_edge_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(kBlock), 0,
                           ctx->stream, y, N, H, W, C, scale, shift, skip.p, skip.pstride, slope);
        check_launch("bn_relu_edge");
    }
}

static bool pool_vec_ok(YRef y, int C, const float* scale, const float* shift, YRef dskip, YRef dpool_, const float* da) {
    const void* dpool = dpool_.p;
    return C % 4 == 0 && dskip.stride(C) % 4 == 0 && y.stride(C) % 4 == 0 &&
           !((reinterpret_cast<uintptr_t>(y.p) | reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
              reinterpret_cast<uintptr_t>(dskip.p) | reinterpret_cast<uintptr_t>(dpool) | reinterpret_cast<uintptr_t>(da)) & 15);
}
int launch_pool_bwd_merge_sums(rfi_ctx* ctx, YRef y, int N, int H, int W, int C, const float* scale,
                               const float* shift, const float* mean, const float* invstd, YRef dskip, YRef dpool,
                               float* da, float slope, float* partial_ws, unsigned short* da16) {
    if (!pool_vec_ok(y, C, scale, shift, dskip, dpool, da16 ? reinterpret_cast<const float*>(da16) : da) || dpool.stride(C) != C ||
        (reinterpret_cast<uintptr_t>(da16) & 7) ||
        ((reinterpret_cast<uintptr_t>(mean) | reinterpret_cast<uintptr_t>(invstd)) & 15))
        return 0;
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C4;
    const bool small = C4 <= kBlock && kBlock % C4 == 0, wide = C4 > kBlock && C4 % kBlock == 0;
    if ((H & 1) || (W & 1) || !(small || wide) || total < (int64_t)kBlock * 16) return 0;      // caller: separate reduction
    int grid = grid_for(total, kMaxRowBlocks);
    int records = grid;
    if (wide) {
        const int per = C4 / kBlock;
        grid = std::max(per, grid / per * per);
        records = grid / per;
    }
    if ((int64_t)grid * kBlock > total) return 0;   // (every thread must own at least one element)
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * (y.bf16 ? 10 : 12) + (double)total * (dpool.bf16 ? 8 : 16));
    hipLaunchKernelGGL(pool_bwd_merge_vec_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, y.p, y.bf16, y.stride(C), N, H, W,
                       C, scale, shift, mean, invstd, dskip.p, dskip.bf16, (int)dskip.stride(C), dpool.p, dpool.bf16, da, slope, reinterpret_cast<double*>(partial_ws), da16);
    check_launch("pool_bwd_merge_sums");
    return records;
}

void launch_pool_bwd_merge(rfi_ctx* ctx, YRef y, int N, int H, int W, int C,
                           const float* scale, const float* shift, YRef dskip, YRef dpool,
                           float* da, float slope, unsigned short* da16) {
    const bool vec = pool_vec_ok(y, C, scale, shift, dskip, dpool, da16 ? reinterpret_cast<const float*>(da16) : da) && dpool.stride(C) == C &&
                     !(reinterpret_cast<uintptr_t>(da16) & 7);
    RFI_REQUIRE(vec || !(dpool.bf16 || da16 || dskip.bf16), "pool_bwd_merge: bfloat16 gradient tensors need C % 4 == 0 and aligned tensors");
    RFI_REQUIRE(!dskip.bf16 || !((H & 1) || (W & 1)), "pool_bwd_merge: a bfloat16 skip gradient needs even H and W");
    RFI_REQUIRE(!da16 || !((H & 1) || (W & 1)), "pool_bwd_merge: a bfloat16 output needs even H and W");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    {
        ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)N * H * W * C * (y.bf16 ? 10 : 12) + (double)total * (dpool.bf16 ? 2 : 4));
        if (vec)
            hipLaunchKernelGGL(pool_bwd_merge_vec_kernel<false>, dim3(grid_for(total / 4)), dim3(kBlock), 0, ctx->stream, y.p, y.bf16,
                               y.stride(C), N, H, W, C, scale, shift, nullptr, nullptr, dskip.p, dskip.bf16, (int)dskip.stride(C), dpool.p, dpool.bf16, da, slope, nullptr, da16);
        else
            hipLaunchKernelGGL(pool_bwd_merge_kernel, dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, y.p, y.bf16, y.stride(C),
                               N, H, W, C, scale, shift, static_cast<const float*>(dskip.p), (int)dskip.stride(C), static_cast<const float*>(dpool.p), da, slope);
        check_launch("pool_bwd_merge");
    }
    if ((H & 1) || (W & 1)) {
        ProfScope ps(ctx, FAM_ELEMWISE);
        hipLaunchKernelGGL(copy_edge_kernel, dim3(grid_for((int64_t)N * H * W * C)), dim3(kBlock), 0,
                           ctx->stream, N, H, W, C, static_cast<const float*>(dskip.p), (int)dskip.stride(C), da);
        check_launch("copy_edge");
    }
}

void launch_head_fwd(rfi_ctx* ctx, YRef yr, int64_t M, int C, const float* scale,
                     const float* shift, const float* w, const float* b, int Cout, float* logits, float slope) {
    ProfScope ps(ctx, FAM_ELEMWISE, 2.0 * M * C * Cout, (double)M * C * (yr.bf16 ? 2 : 4) + (double)M * Cout * 4);
    const bool vec = C % 4 == 0 && !((reinterpret_cast<uintptr_t>(yr.p) | reinterpret_cast<uintptr_t>(scale) |
                                      reinterpret_cast<uintptr_t>(shift) | reinterpret_cast<uintptr_t>(w)) & 15) && yr.stride(C) % 4 == 0;
    RFI_REQUIRE(vec || !yr.bf16, "head_fwd: a bfloat16 input needs C % 4 == 0");
    const float* y = static_cast<const float*>(yr.p);
    if (vec) {
        int L = 1;
        while (L * 2 <= C / 4 && L < 16) L *= 2;
        const unsigned blocks = (unsigned)cdiv(M * L, kBlock);
#define RFI_HF(L_) hipLaunchKernelGGL(head_fwd_vec_kernel<L_>, dim3(blocks), dim3(kBlock), 0, ctx->stream, yr.p, yr.bf16, yr.stride(C), M, C, scale, shift, w, b, Cout, logits, slope)
        if (L == 16) RFI_HF(16); else if (L == 8) RFI_HF(8); else if (L == 4) RFI_HF(4); else if (L == 2) RFI_HF(2); else RFI_HF(1);
#undef RFI_HF
    } else {
        const int64_t threads = M * 16;
        hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)cdiv(threads, kBlock)), dim3(kBlock), 0,
                           ctx->stream, y, M, C, scale, shift, w, b, Cout, logits, slope);
    }
    check_launch("head_fwd");
}

size_t loss_ws_doubles(int64_t count) {
    (void)count;
    return 4 * 1024 + 8;
}
void launch_loss_reduce(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count,
                        double* partial_ws, double* sums4, float* loss_out) {
    int blocks = grid_for(count, 1024);
    {
        ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)count * 5);
        hipLaunchKernelGGL(loss_reduce_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, logits, labels,
                           count, partial_ws);
        check_launch("loss_reduce");
    }
    {
        ProfScope ps(ctx, FAM_ELEMWISE);
        hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial_ws, blocks,
                           (double)count, sums4, loss_out);
        check_launch("loss_finish");
    }
}
void launch_focal_reduce(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count, float alpha,
                         float gamma, double* partial_ws, float* loss_out) {
    int blocks = grid_for(count, 1024);
    {
        ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)count * 5);
        hipLaunchKernelGGL(focal_reduce_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, logits, labels, count,
                           alpha, gamma, partial_ws);
        check_launch("focal_reduce");
    }
    {
        ProfScope ps(ctx, FAM_ELEMWISE);
        hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial_ws, blocks,
                           (double)count, loss_out);
        check_launch("focal_finish");
    }
}
void launch_focal_bwd(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count, float alpha,
                      float gamma, float* dlogits) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)count * 9);
    hipLaunchKernelGGL(focal_bwd_kernel, dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, logits, labels,
                       count, alpha, gamma, dlogits);
    check_launch("focal_bwd");
}
void launch_loss_bwd(rfi_ctx* ctx, const float* logits, const uint8_t* labels, int64_t count,
                     const double* sums4, float* dlogits) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)count * 9);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(grid_for(count)), dim3(kBlock), 0, ctx->stream, logits,
                       labels, count, sums4, dlogits);
    check_launch("loss_bwd");
}

void launch_sigmoid_fwd(rfi_ctx* ctx, const float* z, int64_t n, float* x) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 8);
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, z, n, x);
    check_launch("sigmoid_fwd");
}
void launch_sigmoid_bwd(rfi_ctx* ctx, const float* x, int64_t n, float* d_inout) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 12);
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, x, n, d_inout);
    check_launch("sigmoid_bwd");
}

size_t head_bwd_ws_floats(int64_t M, int C, int Cout) {
    (void)M;
    return (size_t)kMaxRowBlocks * ((size_t)Cout * C + Cout) * 2;
}
int launch_head_bwd(rfi_ctx* ctx, YRef yr, int64_t M, int C, const float* scale,
                    const float* shift, const float* w, int Cout, const float* dlogits, float* da,
                    float* partial_ws, float* dw, float* db, float slope, const float* bn_mean, const float* bn_invstd,
                    float* bn_records_ws, unsigned short* da16, bool* skip_da, bool finish) {
    // skip_da (in: the caller can do without da; out: it was not written -- only together with the BatchNorm-backward sums)
    const bool may_skip = skip_da && *skip_da;
    if (skip_da) *skip_da = false;
    int bn_records = 0;
    const bool vec = C % 4 == 0 && !((reinterpret_cast<uintptr_t>(yr.p) | reinterpret_cast<uintptr_t>(da) |
                                      reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift) |
                                      reinterpret_cast<uintptr_t>(w)) & 15) && yr.stride(C) % 4 == 0;
    RFI_REQUIRE(vec || !yr.bf16, "head_bwd: a bfloat16 input needs C % 4 == 0");
    // finish = false leaves the records for a batched finisher whose table was filled with the VECTOR kernel's record geometry
    RFI_REQUIRE(finish || vec, "head_bwd: deferred finishing needs the vector path (C % 4 == 0, 16-byte aligned tensors)");
    RFI_REQUIRE(!da16 || (vec && Cout == 1 && !(reinterpret_cast<uintptr_t>(da16) & 7)), "head_bwd: a bfloat16 gradient tensor needs C % 4 == 0 and one output channel");
    const float* y = static_cast<const float*>(yr.p);
    ChanGeom g = geom_rows(M, C, vec);     // scalar kernel: one channel per lane; vector kernel: four
    {
        ProfScope ps(ctx, FAM_ELEMWISE, 4.0 * M * C * Cout, (double)M * C * (yr.bf16 ? 6 : 8));
        if (vec) {
            const bool sums = bn_records_ws && bn_mean && bn_invstd && Cout == 1 &&
                              !((reinterpret_cast<uintptr_t>(bn_mean) | reinterpret_cast<uintptr_t>(bn_invstd)) & 15);
            hipLaunchKernelGGL(head_bwd_vec_kernel, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, yr.p, yr.bf16,
                               yr.stride(C), M, C, g.CL, g.rows_per_block, scale, shift, w, Cout, dlogits, da,
                               reinterpret_cast<double*>(partial_ws), slope, bn_mean, bn_invstd,
                               sums ? reinterpret_cast<double*>(bn_records_ws) : nullptr, da16, (sums && may_skip && !da16) ? 0 : 1);
            if (sums) bn_records = g.rblocks;
            if (sums && may_skip && !da16) *skip_da = true;
        }
        else
            hipLaunchKernelGGL(head_bwd_kernel, dim3(g.rblocks, g.cblocks), dim3(kBlock), 0, ctx->stream, y,
                               M, C, g.CL, g.rows_per_block, scale, shift, w, Cout, dlogits, da,
                               reinterpret_cast<double*>(partial_ws), slope);
        check_launch("head_bwd");
    }
    if (finish) {
        // dw (Cout*C values) followed by db (Cout values) are contiguous in each partial record
        const int n = Cout * C + Cout;
        ProfScope ps(ctx, FAM_REDUCE);
        // dw and db are adjacent in the flat gradient buffer only by construction of the caller;
        // finish into dw[0..Cout*C) and db[0..Cout) separately
        hipLaunchKernelGGL(finish_channel_sum_kernel, dim3((int)cdiv(Cout * C, kFinCh)), dim3(kBlock), 0,
                           ctx->stream, reinterpret_cast<const double*>(partial_ws), g.rblocks, (int64_t)n, Cout * C, dw);
        check_launch("head_bwd_finish_dw");
    }
    if (finish) {
        const int n = Cout * C + Cout;
        ProfScope ps(ctx, FAM_REDUCE);
        hipLaunchKernelGGL(finish_channel_sum_kernel, dim3((int)cdiv(Cout, kFinCh)), dim3(kBlock), 0,
                           ctx->stream, reinterpret_cast<const double*>(partial_ws) + (size_t)Cout * C,
                           g.rblocks, (int64_t)n, Cout, db);
        check_launch("head_bwd_finish_db");
    }
    return bn_records;
}

void launch_nchw_to_nhwc(rfi_ctx* ctx, const float* src, int N, int C, int H, int W, float* dst) {
    const int64_t total = (int64_t)N * C * H * W;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 8);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, src, N,
                       C, H, W, dst);
    check_launch("nchw_to_nhwc");
}
void launch_nhwc_to_nchw(rfi_ctx* ctx, const float* src, int N, int C, int H, int W, float* dst) {
    const int64_t total = (int64_t)N * C * H * W;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 8);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, src, N,
                       C, H, W, dst);
    check_launch("nhwc_to_nchw");
}
void launch_weight_to_dgrad(rfi_ctx* ctx, const float* wf, int taps, int Cout, int Cin, int flip,
                            float* wd) {
    const int64_t total = (int64_t)taps * Cout * Cin;
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)total * 8);
    hipLaunchKernelGGL(weight_to_dgrad_kernel, dim3(grid_for(total)), dim3(kBlock), 0, ctx->stream, wf,
                       taps, Cout, Cin, flip, wd);
    check_launch("weight_to_dgrad");
}
int64_t relayout_assign_tiles(RelayoutDesc* descs, int n) {
    int64_t t = 0;
    for (int i = 0; i < n; ++i) {
        descs[i].tile0 = t;
        t += (int64_t)descs[i].taps * ((descs[i].cout + 31) / 32) * ((descs[i].cin + 31) / 32);
    }
    return t;
}
void launch_weight_to_dgrad_batched(rfi_ctx* ctx, const RelayoutDesc* descs_dev, int n, const float* src,
                                    float* dst, double total_bytes, int64_t total_tiles) {
    RFI_REQUIRE(total_tiles > 0 && total_tiles < ((int64_t)1 << 31), "weight_to_dgrad_batched: bad tile count");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, total_bytes);
    hipLaunchKernelGGL(weight_to_dgrad_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, ctx->stream, descs_dev, n,
                       src, dst);
    check_launch("weight_to_dgrad_batched");
}
void launch_relu_bwd(rfi_ctx* ctx, float* dA, const float* Y, int64_t n) {
    RFI_REQUIRE(n % 4 == 0, "relu_bwd: element count must be a multiple of 4");
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 12);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(kBlock), 0, ctx->stream,
                       reinterpret_cast<float4*>(dA), reinterpret_cast<const float4*>(Y), n / 4);
    check_launch("relu_bwd");
}
void launch_pad_channels(rfi_ctx* ctx, const float* src, int64_t M, int c, int cp, float* dst) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)M * (c + cp) * 4);
    hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_for(M * cp)), dim3(kBlock), 0, ctx->stream, src, M, c,
                       cp, dst);
    check_launch("pad_channels");
}
void launch_u8_to_f32(rfi_ctx* ctx, const uint8_t* src, int64_t n, float* dst) {
    ProfScope ps(ctx, FAM_ELEMWISE, 0, (double)n * 5);
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, src, n, dst);
    check_launch("u8_to_f32");
}

size_t sumsq_ws_doubles(int64_t n) {
    (void)n;
    return 1024 + 8;
}
void launch_sumsq(rfi_ctx* ctx, const float* g, int64_t n, double* partial_ws, double* sumsq) {
    int blocks = grid_for(n / 4 + 1, 1024);
    {
        ProfScope ps(ctx, FAM_OPTIM, 0, (double)n * 4);
        hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, g, n, partial_ws);
        check_launch("sumsq");
    }
    {
        ProfScope ps(ctx, FAM_OPTIM);
        hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, partial_ws, blocks,
                           sumsq);
        check_launch("sumsq_finish");
    }
}
void launch_adam(rfi_ctx* ctx, const AdamArgs& a) {
    ProfScope ps(ctx, FAM_OPTIM, 0, (double)a.n * 28);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(a.n)), dim3(kBlock), 0, ctx->stream, a);
    check_launch("adam");
}

void launch_scale_inplace(rfi_ctx* ctx, float* x, int64_t n, float f) {
    if (n <= 0) return;
    ProfScope ps(ctx, FAM_COMM, 0, (double)n * 8);
    hipLaunchKernelGGL(scale_inplace_kernel, dim3(grid_for(n)), dim3(kBlock), 0, ctx->stream, x, n, f);
    check_launch("scale_inplace");
}
void launch_reduce_slabs(rfi_ctx* ctx, const float* slabs, int nslabs, int64_t n, float* out) {
    // fold groups of 32 slabs in place (each group's sum lands in the group's first slab) until
    // at most 32 partial slabs remain, then sum those into `out`.  Fixed order -> reproducible.
    float* s = const_cast<float*>(slabs);
    int count = nslabs;      // live partial slabs, `stride` slabs apart
    int stride = 1;
    const bool vec = n % 4 == 0 && !((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(out)) & 15);
    while (count > 32) {
        const int groups = (int)cdiv(count, 32);
        ProfScope ps(ctx, FAM_REDUCE, 0, (double)n * 4 * (count + groups));
        if (vec)
            hipLaunchKernelGGL(reduce_slabs_vec4_kernel, dim3(grid_for(n / 4, 1024), groups), dim3(kBlock), 0, ctx->stream, s, count, 32,
                               stride, n / 4, s, (int64_t)32 * stride);
        else
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(n, 1024), groups), dim3(kBlock), 0,
                           ctx->stream, s, count, 32, stride, n, s, (int64_t)32 * stride);
        check_launch("reduce_slabs_fold");
        // group g wrote slab index g*32*stride; element i of it is only touched by the thread
        // that summed element i of that group, so the in-place fold is race free
        count = groups;
        stride *= 32;
    }
    ProfScope ps(ctx, FAM_REDUCE, 0, (double)n * 4 * (count + 1));
    if (vec)
        hipLaunchKernelGGL(reduce_slabs_vec4_kernel, dim3(grid_for(n / 4, 1024), 1), dim3(kBlock), 0, ctx->stream, s, count, 32, stride,
                           n / 4, out, (int64_t)0);
    else
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(n, 1024), 1), dim3(kBlock), 0, ctx->stream, s,
                       count, 32, stride, n, out, (int64_t)0);
    check_launch("reduce_slabs");
}

}  // namespace rfi
