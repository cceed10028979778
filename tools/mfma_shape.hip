// Microbenchmark (GPU box): the consumer loop of the wave-specialised kernels (LDS fragment reads + the 3 x bf16 block
// product) in the two bf16 MFMA shapes, on random data, with and without a VALU partner wave per SIMD.  Question: under
// the power cap, does v_mfma_f32_16x16x32_bf16 deliver more FLOP/s than v_mfma_f32_32x32x16_bf16 at equal cycles per
// FLOP (MI355X guide, "DVFS give-back" item 7)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDS_BYTES = 64 * 1024;

// SHAPE 32: per iteration 12 fragment reads + 24 MFMAs 32x32x16 (a 64 x 64 wave tile, one 16-channel chunk of one tap)
// SHAPE 16: per iteration 20 fragment reads + 48 MFMAs 16x16x32 (the same tile; planes paired along K)
// PARTNER: waves 4-7 run a split-like VALU loop on registers (the producers' instruction mix, no memory)
template <int SHAPE, bool PARTNER>
__global__ __launch_bounds__(PARTNER ? 512 : 256) void k(const u32x4* src, float* out, int iters, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<u32x4*>(smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (PARTNER && wave >= 4) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
        unsigned acc = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {        // ~8 instructions per value pair: cvt, shift / and, 2 sub, cvt, ...
                const float a = v[i], b = v[i + 1];
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, b2));
                const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
                const unsigned m = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{ra, rb}, b2));
                acc ^= h + m;
                v[i] = ra * 1.5f + 0.25f;
                v[i + 1] = rb * 1.25f + 0.5f;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = v[0] + (float)acc;
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned base = lane * 16;
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
        bf16x8 fr[2][12];
        auto load = [&](int it, bf16x8 (&f)[12]) {
#pragma unroll
            for (int j = 0; j < 12; ++j) f[j] = *reinterpret_cast<const bf16x8*>(smem + ((base + j * 1024 + it * 12288) & (LDS_BYTES - 1)));
        };
        load(0, fr[0]);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                load(it + u + 1, fr[(u + 1) & 1]);
                const bf16x8(&f)[12] = fr[u];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        f32x16 c = acc[mt][nt];
                        const bf16x8 *a = &f[mt * 3], *b = &f[6 + nt * 3];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
                        acc[mt][nt] = c;
                    }
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) acc[i >> 2][i & 3][r] = 0.f;
        bf16x8 fr[2][20];
        auto load = [&](int it, bf16x8 (&f)[20]) {
#pragma unroll
            for (int j = 0; j < 20; ++j) f[j] = *reinterpret_cast<const bf16x8*>(smem + ((base + j * 1024 + it * 20480) & (LDS_BYTES - 1)));
        };
        load(0, fr[0]);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                load(it + u + 1, fr[(u + 1) & 1]);
                const bf16x8(&f)[20] = fr[u];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        f32x4 c = acc[mt][nt];
                        const bf16x8 *a = &f[mt * 2], *b = &f[8 + nt * 3];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);      // [l | h] . [h | l]
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);      // [m | h] . [m | m]
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], c, 0, 0, 0);      // [m | h] . [h | h]
                        acc[mt][nt] = c;
                    }
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if ((i + 1) * 20 / 48 - i * 20 / 48 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 16; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i >> 2][i & 3][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * (PARTNER ? 512 : 256) + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE, bool PARTNER>
void run(const u32x4* src, int iters, int reps) {
    const int blocks = 256, threads = PARTNER ? 512 : 256;
    float* out;
    unsigned long long* st;
    CHECK(hipMalloc(&out, blocks * 512 * 4));
    CHECK(hipMalloc(&st, blocks * 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, PARTNER>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL((k<SHAPE, PARTNER>), dim3(blocks), dim3(threads), LDS_BYTES, 0, src, out, 200, st);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<SHAPE, PARTNER>), dim3(blocks), dim3(threads), LDS_BYTES, 0, src, out, iters, st);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CHECK(hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (int b = 0; b < blocks; ++b) {
        clk.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 100.0);
        cyc.push_back((double)h[b * 2] / iters);
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop = (double)reps * blocks * 4 * iters * 786432.0;
    printf("shape %2d%s: %8.2f ms  %7.1f TFLOP/s (3xbf16-equivalent float32 work: %6.1f)  clock %.0f MHz (median)  %.1f cycles/iter (ideal 768)\n", SHAPE,
           PARTNER ? " + VALU partner" : "               ", ms, flop / (ms * 1e-3) / 1e12, flop / 6.0 / (ms * 1e-3) / 1e12, clk[blocks / 2], cyc[blocks / 2]);
    fflush(stdout);
    CHECK(hipFree(out));
    CHECK(hipFree(st));
}

int main() {
    std::vector<unsigned short> h(LDS_BYTES / 2);
    srand(1);
    for (auto& v : h) {
        const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    u32x4* src;
    CHECK(hipMalloc(&src, LDS_BYTES));
    CHECK(hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int round = 0; round < 2; ++round) {
        run<32, false>(src, iters, 30);
        run<16, false>(src, iters, 30);
        run<32, true>(src, iters, 30);
        run<16, true>(src, iters, 30);
    }
    return 0;
}
