// Microbenchmark (GPU box): issue rate of the MFMA instructions this project uses or considers.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    s16x8 a8 = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[i], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b8), acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 6 x 32x32x16 bf16 MFMAs + NV independent fp32 VALU ops per iteration: do the pipes overlap?
template <int NV>
__global__ __launch_bounds__(256) void kmix(float* out, int iters) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a8 = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b8), acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 6 x 32x32x16 bf16 MFMAs + NL ds_read_b128 per iteration (reads feed a dependent xor chain only at the end)
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int NL>
__global__ __launch_bounds__(256) void klds(float* out, int iters) {
    __shared__ f32x4m sm[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = f32x4m{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    s16x8 a8 = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7}, b8 = {7, 6, 5, 4, 3, 2, 1, (short)threadIdx.x};
    f32x4m v = {0.f, 0.f, 0.f, 0.f};
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a8), __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b8), acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const f32x4m t = sm[(idx + i * 7 + it) & 1023];
            v.x += t.x;
        }
    }
    float s = v.x;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NL>
void runlds(int blocks) {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 20000;
    hipLaunchKernelGGL(klds<NL>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(klds<NL>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("6 mfma16 + %3d ds_read_b128 (+1 add each), %4d blocks: %8.2f ms  ~%6.0f cycles/iter/wave-slot\n", NL, blocks, ms,
           ms * 1e-3 * 2.3e9 / iters / (blocks / 256.0));
    hipFree(d);
}
template <int NV>
void runmix(int blocks) {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    const int iters = 20000;
    hipLaunchKernelGGL(kmix<NV>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kmix<NV>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_cycles_per_wave_slot = ms * 1e-3 * 2.3e9 / iters / (blocks / 256.0);
    printf("6 mfma16 + %3d valu, %4d blocks (%d waves/SIMD): %8.2f ms  ~%6.0f cycles/iter/wave-slot\n", NV, blocks,
           blocks / 256, ms, per_iter_cycles_per_wave_slot);
    hipFree(d);
}
template <int KIND>
void run(const char* name, double flop_per_mfma) {
    float* d;
    hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 20000, blocks = 1024;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 /*waves*/ * iters * 4;
    printf("%-28s %8.2f ms  %8.1f TFLOP/s\n", name, ms, n * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    run<0>("f32 32x32x2", 4096.0);
    run<1>("bf16_1k 32x32x8", 16384.0);
    run<2>("bf16 32x32x16", 32768.0);
    runmix<0>(256); runmix<40>(256); runmix<80>(256);
    runmix<0>(512); runmix<40>(512); runmix<80>(512);
    runlds<0>(512); runlds<8>(512); runlds<16>(512); runlds<32>(512);
    return 0;
}
