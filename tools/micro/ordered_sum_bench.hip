// Micro-benchmark of the in-order reductions of lsmr.hip: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/osb tools/micro/ordered_sum_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

template <int MODE, int KB>
__global__ __launch_bounds__(64) void k_dot(int n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    __shared__ float lds_acc;
    if (threadIdx.x == 0) lds_acc = 0.0f;
    __syncthreads();
    float acc = 0.0f;
    const long long c0 = clock64(), w0 = wall_clock64();
    float cur[KB], nxt[KB];
#pragma unroll
    for (int u = 0; u < KB; ++u) { const int i = u * 64 + lane; cur[u] = i < n ? a[i] * b[i] : 0.0f; }
    for (int base = 0; base < n; base += 64 * KB) {
#pragma unroll
        for (int u = 0; u < KB; ++u) { const int i = base + 64 * KB + u * 64 + lane; nxt[u] = i < n ? a[i] * b[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (MODE == 0) {             // loads only
                acc += cur[u];
            } else if (MODE == 1) {      // DPP chain
                float v = lane == 0 ? acc + cur[u] : cur[u];
#pragma unroll
                for (int k = 1; k < 64; ++k)
                    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(cur[u]));
                acc = __shfl(v, 63);
            } else if (MODE == 3) {      // LDS atomic: same-address float adds are applied one lane after the other
                __hip_atomic_fetch_add(&lds_acc, cur[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {                     // readlane chain
#pragma unroll
                for (int i = 0; i < 64; ++i) acc = acc + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur[u]), i));
            }
        }
#pragma unroll
        for (int u = 0; u < KB; ++u) cur[u] = nxt[u];
    }
    __syncthreads();
    if (MODE == 3) acc = lds_acc;
    if (lane == 0) { out[0] = acc; out[1] = (float)(clock64() - c0); out[2] = (float)(wall_clock64() - w0); }
}

int main()
{
    const int n = 133128;
    std::vector<float> h(n, 1.0f), g(n, 1.0f);
    unsigned st = 12345u; for (int i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((st >> 8) & 0xffff) / 65536.0f - 0.5f; st = st * 1664525u + 1013904223u; g[i] = ((st >> 8) & 0xffff) / 32768.0f - 1.0f; }
    float want = 0.0f; for (int i = 0; i < n; ++i) { volatile float p = h[i] * g[i]; want = want + p; }
    printf("host in-order sum %.9g\n", want);
    float *a, *b, *o;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, 64);
    hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(b, g.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern) {
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, n, a, b, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float r3[3]; hipMemcpy(r3, o, 12, hipMemcpyDeviceToHost); float r0 = r3[0];
            printf("%-28s %.3f ms per call (%.1f ns per element) result %.9g | shader cycles %.0f (%.1f per element), 100 MHz ticks %.0f -> %.0f MHz\n", name, ms / 10, 1e6 * ms / 10 / n, r0, r3[1], r3[1] / n, r3[2], r3[1] / r3[2] * 100.0);
        }
    };
    run("loads only, 4 batches", k_dot<0, 4>);
    run("loads only, 16 batches", k_dot<0, 16>);
    run("dpp chain, 4 batches", k_dot<1, 4>);
    run("dpp chain, 16 batches", k_dot<1, 16>);
    run("readlane chain, 4 batches", k_dot<2, 4>);
    run("lds atomic, 4 batches", k_dot<3, 4>);
    run("lds atomic, 16 batches", k_dot<3, 16>);
    return 0;
}
