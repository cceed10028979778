// Probe (GPU box): is  v_dot2c_f32_bf16(h_pk, (-1, 0), a)  ==  a - float(h.lo)  bit for bit?  (the residual of the
// float32 -> 3 x bf16 split in ONE instruction instead of shift + subtract).  Build: hipcc --offload-arch=gfx950 -O3 -o build/probe_dot2 tools/probe_dot2_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// (the builtin is not used: for the multiplier (-1, 0) the compiler emits the inline constant -1.0, which the hardware
// reads as the 32-bit pattern 0xbf800000 = (0, -1))
__device__ __forceinline__ float dot2c(unsigned h, unsigned mult, float acc) {
    asm("v_dot2c_f32_bf16_e32 %0, %1, %2" : "+v"(acc) : "s"(mult), "v"(h));
    return acc;
}
__global__ void k(const float* a, unsigned* bad, unsigned* first, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 2 + 1 >= n) return;
    float v0 = a[i * 2], v1 = a[i * 2 + 1];
    f2 v = {v0, v1};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    unsigned hb = __builtin_bit_cast(unsigned, h);
    float r0 = dot2c(hb, 0x0000bf80u, v0);
    float r1 = dot2c(hb, 0xbf800000u, v1);
    float e0 = v0 - __builtin_bit_cast(float, hb << 16), e1 = v1 - __builtin_bit_cast(float, hb & 0xffff0000u);
    // second level: residual of the residual
    f2 rr = {e0, e1};
    bf16x2 m = __builtin_convertvector(rr, bf16x2);
    unsigned mb = __builtin_bit_cast(unsigned, m);
    float s0 = dot2c(mb, 0x0000bf80u, e0), s1 = dot2c(mb, 0xbf800000u, e1);
    float g0 = e0 - __builtin_bit_cast(float, mb << 16), g1 = e1 - __builtin_bit_cast(float, mb & 0xffff0000u);
    bool b = __builtin_bit_cast(unsigned, r0) != __builtin_bit_cast(unsigned, e0) || __builtin_bit_cast(unsigned, r1) != __builtin_bit_cast(unsigned, e1) ||
             __builtin_bit_cast(unsigned, s0) != __builtin_bit_cast(unsigned, g0) || __builtin_bit_cast(unsigned, s1) != __builtin_bit_cast(unsigned, g1);
    if (b) { if (atomicAdd(bad, 1u) == 0) { first[0] = __builtin_bit_cast(unsigned, v0); first[1] = __builtin_bit_cast(unsigned, r0); first[2] = __builtin_bit_cast(unsigned, e0);
             first[3] = __builtin_bit_cast(unsigned, v1); first[4] = __builtin_bit_cast(unsigned, r1); first[5] = __builtin_bit_cast(unsigned, e1); } }
}
int main() {
    const int n = 1 << 26;
    std::vector<float> h(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {          // random bit patterns: every exponent incl. denormals; NaN / inf excluded
        s = s * 1664525u + 1013904223u; unsigned b = s; s = s * 1664525u + 1013904223u; b ^= s >> 7;
        if (((b >> 23) & 0xff) >= 0xfe) b &= ~(1u << 30);      // (|v| >= 2^127 rounds to inf in bf16: outside the domain of either form)
        if (i % 3 == 0) b = (b & 0x807fffffu) | ((100u + (b >> 23) % 56u) << 23);      // a third: ordinary magnitudes
        memcpy(&h[i], &b, 4);
    }
    float* d; unsigned *bad, *first;
    hipMalloc(&d, n * 4); hipMalloc(&bad, 4); hipMalloc(&first, 32);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4); hipMemset(first, 0, 32);
    k<<<n / 2 / 256, 256>>>(d, bad, first, n);
    unsigned nb, f[8];
    hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
    printf("pairs %d mismatching %u first: v0 %08x dot %08x sub %08x | v1 %08x dot %08x sub %08x\n", n / 2, nb, f[0], f[1], f[2], f[3], f[4], f[5]);
    return 0;
}
