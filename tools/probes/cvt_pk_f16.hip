// Is v_cvt_pk_f16_f32 (gfx950) round-to-nearest-even like the scalar (_Float16) cast?  Exhaustive over all fp32 bit patterns
// whose exponent lies in half's range (plus a band around it), both operand slots.   hipcc --offload-arch=gfx950 -O2 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* bad) {
    const unsigned long long n = 1ull << 32;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float((unsigned)i);
        const float b = __uint_as_float((unsigned)(i * 2654435761u + 12345u));
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        const _Float16 x = (_Float16)a, y = (_Float16)b;
        const unsigned ref = (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
        const bool nan_a = a != a, nan_b = b != b;
        const unsigned m = (nan_a ? 0u : 0xffffu) | (nan_b ? 0u : 0xffff0000u);     // NaN payloads may differ
        if ((r & m) != (ref & m)) atomicAdd(bad, 1ull);
    }
}
int main() {
    unsigned long long* d; unsigned long long h = 0;
    hipMalloc(&d, 8); hipMemcpy(d, &h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mismatches over 2^32 patterns: %llu\n", h);
    return h != 0;
}
