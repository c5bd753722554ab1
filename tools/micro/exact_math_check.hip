// Device check of the two exact shortcuts of csrc/eikonal_core.h against the compiler's IEEE expansions, over every float pattern of
// their domains: sqrt_pos(x) == sqrtf(x) for all x in [1e-30, 1e30], div3(x) == x / 3.0f for all finite x except -0.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/exact_math_check tools/micro/exact_math_check.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../dsurftomo_amd/csrc/eikonal_core.h"
__global__ void k_check(unsigned long long* bad)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long b0 = 0, b1 = 0;
    for (unsigned long long u = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((unsigned)u);
        if (x >= 1e-30f && x <= 1e30f && __float_as_uint(dsa::sqrt_pos(x)) != __float_as_uint(sqrtf(x))) ++b0;
        if (fabsf(x) < __builtin_inff() && (unsigned)u != 0x80000000u && __float_as_uint(dsa::div3(x)) != __float_as_uint(x / 3.0f)) ++b1;
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
}
int main()
{
    unsigned long long* d; unsigned long long h[2] = { 0, 0 };
    hipMalloc(&d, 16); hipMemset(d, 0, 16);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("sqrt_pos vs sqrtf over [1e-30, 1e30]: %llu mismatches; div3 vs x / 3.0f over all finite x (except -0): %llu mismatches\n", h[0], h[1]);
    return (h[0] || h[1]) ? 1 : 0;
}
