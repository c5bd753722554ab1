// Device check of the lean per-lane quadrant evaluation (exact_march.h: x_quad_lane) against the literal per-quadrant form
// (x_quad_candidates) on random stencil states, as the march kernel calls them: four lanes per node, minimum by DPP.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I dsurftomo_amd/csrc tools/micro/quad_check.hip -o /tmp/quad_check && /tmp/quad_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "exact_march.h"
using namespace dsa;
__device__ unsigned long long rng(unsigned long long& st) { st = st * 6364136223846793005ull + 1442695040888963407ull; return st >> 33; }
__global__ void k_check(unsigned long long seed, int iters, unsigned long long* bad, float* sample)
{
    const int lane = threadIdx.x & 3, j = lane >> 1, k = lane & 1;
    unsigned long long st = seed + (blockIdx.x * blockDim.x + threadIdx.x) / 4 * 7919ull;     // the four lanes of a node share the stream
    for (int it = 0; it < iters; ++it) {
        auto uni = [&]() { return (float)(rng(st) & 0xffffff) / 16777216.0f; };
        NodeGeom g; g.ri = 6371.0f; g.risti = g.ri * (0.3f + 0.6f * uni()); g.dnx = 2.18e-5f; g.dnz = g.dnx;
        const float slown = 1.0f / (2.5f + uni());
        const float hx = g.ri * g.dnx * slown, hz = g.risti * g.dnz * slown;
        const float t0 = 30.0f * uni();
        bool ej[2], ek[2], aj[2], ak[2], oj[2], ok[2]; float tj[2], tk[2], tj2[2], tk2[2];
        for (int d = 0; d < 2; ++d) {
            const unsigned r = (unsigned)rng(st);
            ej[d] = (r & 15) != 0; ek[d] = ((r >> 4) & 15) != 0;
            aj[d] = ej[d] && ((r >> 8) & 3) != 0; ak[d] = ek[d] && ((r >> 10) & 3) != 0;
            oj[d] = ((r >> 12) & 3) != 0; ok[d] = ((r >> 14) & 3) != 0;
            const float u1 = uni(), u2 = uni(), u3 = uni(), u4 = uni();
            tj[d] = aj[d] ? t0 + hx * (2.4f * u1 - 1.2f) : ((r >> 20) & 1 ? 0.0f : t0 + 3 * hx * u1);
            tk[d] = ak[d] ? t0 + hz * (2.4f * u2 - 1.2f) : ((r >> 21) & 1 ? 0.0f : t0 + 3 * hz * u2);
            if ((r >> 16) & 1) { if (aj[d] && ak[d]) tk[d] = tj[d]; }
            tj2[d] = oj[d] ? tj[d] + hx * (1.6f * u3 - 1.1f) : 0.0f;
            tk2[d] = ok[d] ? tk[d] + hz * (1.6f * u4 - 1.1f) : 0.0f;
        }
        XQuadState q;
        q.ej = ej[j]; q.aj = aj[j]; q.oj = oj[j]; q.tj = tj[j]; q.tj2 = tj2[j];
        q.ek = ek[k]; q.ak = ak[k]; q.ok = ok[k]; q.tk = tk[k]; q.tk2 = tk2[k];
        const bool k_dead = (ek[0] && !ak[0]) || (ek[1] && !ak[1]), j_dead = (ej[0] && !aj[0]) || (ej[1] && !aj[1]);
        float a = x_quad_lane(q, k_dead, j_dead, j, k, slown, g);
        XQuadState l = q;
        l.tj = q.aj ? q.tj : kInf; l.tj2 = q.oj ? q.tj2 : kInf; l.tk = q.ak ? q.tk : kInf; l.tk2 = q.ok ? q.tk2 : kInf;
        float b = x_quad_candidates(l, k_dead, j_dead, slown, g);
        for (int m = 1; m < 4; m <<= 1) { const float oa = __shfl_xor(a, m, 4), ob = __shfl_xor(b, m, 4); a = oa < a ? oa : a; b = ob < b ? ob : b; }
        if (__float_as_uint(a) != __float_as_uint(b) && lane == 0) { const unsigned long long n = atomicAdd(bad, 1ull); if (n < 4) { sample[2 * n] = a; sample[2 * n + 1] = b; } }
    }
}
int main()
{
    unsigned long long* bad; float* sample;
    hipMalloc(&bad, 8); hipMalloc(&sample, 64); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(1024), dim3(256), 0, 0, 20261003ull, 2000, bad, sample);
    unsigned long long h = 0; float hs[8];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 32, hipMemcpyDeviceToHost);
    printf("quad_check: %llu of %llu node evaluations differ between x_quad_lane and x_quad_candidates on the device\n", h, 1024ull * 64 * 2000);
    for (int i = 0; i < 4 && i < (int)h; ++i) printf("  lane form %.9g literal %.9g\n", hs[2 * i], hs[2 * i + 1]);
    return h ? 1 : 0;
}
