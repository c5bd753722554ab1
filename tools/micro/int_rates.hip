// Issue cost of the integer instructions the solve kernel leans on, relative to v_add_u32 (gfx950): 8 independent chains per lane,
// 4 waves per SIMD, so the loop is issue bound.  hipcc --offload-arch=gfx950 -O3 -o tools/micro/int_rates tools/micro/int_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters)
{
    unsigned a[8];
    unsigned long long b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 977u + i * 131u; b[i] = ((unsigned long long)a[i] << 17) | 5u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + seed;
            if (OP == 1) a[i] = a[i] * 2654435761u;
            if (OP == 2) a[i] = __umulhi(a[i], 2654435761u) + 1u;
            if (OP == 3) b[i] = (b[i] << (seed & 7)) | 1ull;
            if (OP == 4) b[i] = (b[i] >> 1) & 0x7f7f7f7f7f7f7f7full;
            if (OP == 5) a[i] = (a[i] << 3) ^ seed;
            if (OP == 6) a[i] = __popcll(b[i] ^ a[i]);
            if (OP == 7) a[i] = (unsigned)(__ffsll((long long)(b[i] | a[i]))) + a[i];
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)b[i] + (unsigned)(b[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> float run(unsigned* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3u, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(1024), dim3(256), 0, 0, d, 3u, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    unsigned* d; hipMalloc(&d, 1024 * 256 * 4);
    const int iters = 20000;
    const char* names[8] = { "v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32 + add", "64-bit shift left (variable) | 1", "64-bit shift right 1 & mask", "v_lshlrev_b32 ^", "popcount 64 (of xor)", "ffs 64 + add" };
    float t[8] = { run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters), run<6>(d, iters), run<7>(d, iters) };
    for (int i = 0; i < 8; ++i) printf("%-36s %8.3f ms  x%.2f of v_add_u32\n", names[i], t[i], t[i] / t[0]);
    return 0;
}
