// K7: dispersion curves and depth kernels for every column of the Vs model (reference depthkernel
// CalSurfG.f90:1-169, caldespersion :2866-2927, on top of surfdisp96.f).
//
// One lane per curve: (column, perturbation) with perturbation 0 = the model itself and
// 1 + 6 i + 2 q + s = parameter q (Vs, Vp, rho) of depth i scaled by 1 -+ 0.005.  Columns vary
// fastest so that a wavefront holds neighbouring columns under the same perturbation: similar
// models, similar trip counts.  The curves land in a (perturbation, period, column) buffer and a
// second, trivial kernel turns them into pv and the central differences.
#include "kernels.h"
#include "dispersion_core.h"

namespace dsa {

constexpr int kMaxDepths = 64;

// DSA_DISP_WAVES: wavefronts per SIMD the register allocation aims at (0: the compiler's choice, 177 VGPRs = 2 for the Rayleigh kernel)
#ifndef DSA_DISP_WAVES
#define DSA_DISP_WAVES 0
#endif
#if DSA_DISP_WAVES > 0
#define DSA_DISP_OCC __attribute__((amdgpu_waves_per_eu(DSA_DISP_WAVES)))
#else
#define DSA_DISP_OCC
#endif
template <int IFUNC>
__global__ __launch_bounds__(64) DSA_DISP_OCC void k_dispersion(const LayerGeom* __restrict__ G, const float* __restrict__ vels, int ncol,
                                                   int npert, int igr, int kmax, const double* __restrict__ t,
                                                   float* __restrict__ ws, size_t nlanes, double* __restrict__ curves, int layers_in_lds,
                                                   int gshift, unsigned long long* __restrict__ diag, unsigned long long* __restrict__ fail_list, int fail_cap)
{
    // gshift > 0 (few curves): 2^gshift neighbouring lanes share one curve -- see Layers::gsize.  A curve is one
    // dependent chain of ~20 000 layer matrices; with one lane per curve a call with 324 columns keeps a quarter of the
    // SIMDs busy with one wavefront each.
    const size_t lane_id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tid = lane_id >> gshift;                  // curve
    if (tid >= (size_t)ncol * npert) return;
    const int p = (int)(tid / ncol), c = (int)(tid - (size_t)p * ncol);
    const int nz = G->nz;
    float vs[kMaxDepths], vp[kMaxDepths], rho[kMaxDepths];
    for (int k = 0; k < nz; ++k) {
        vs[k] = vels[(size_t)k * ncol + c];
        brocher_vp_rho(vs[k], &vp[k], &rho[k]);
    }
    if (p > 0) {
        const int idx = p - 1, i = idx / 6, q = (idx % 6) >> 1, s = idx & 1;
        float* arr = q == 0 ? vs : (q == 1 ? vp : rho);
        const float base = arr[i];
        const float dln = 0.01f;
        arr[i] = s ? base + 0.5f * dln * base : base - 0.5f * dln * base;
    }
    Layers m;
    extern __shared__ __attribute__((aligned(16))) float lds_layers[];
    if (layers_in_lds) {
        // the layer table of the curves of this wavefront in LDS ([layer][curve]: conflict free): the secular function
        // walks it once per evaluation, ~25 evaluations per root (the lanes of a group write the same table)
        const int slot = (int)(threadIdx.x >> gshift), per = 64 >> gshift, plane = G->rmax * per;
        m.d = lds_layers + slot; m.a = lds_layers + plane + slot; m.b = lds_layers + 2 * plane + slot; m.rho = lds_layers + 3 * plane + slot;
        m.stride = (size_t)per;
        if (gshift > 0) {
            m.gsize = 1 << gshift;
            m.gsub = (int)(threadIdx.x & (m.gsize - 1));
            double* xbase = reinterpret_cast<double*>(lds_layers + 4 * plane);                    // 16 * plane bytes in: 8-byte aligned
            m.xch = xbase + (size_t)slot * m.gsize * 15;
        }
    } else {
        const size_t plane = (size_t)G->rmax * nlanes;
        m.d = ws + tid; m.a = ws + plane + tid; m.b = ws + 2 * plane + tid; m.rho = ws + 3 * plane + tid;
        m.stride = nlanes;
    }
    build_layers<IFUNC>(*G, vs, vp, rho, m);
    const int kfail = dispersion_curve<IFUNC>(m, igr, kmax, t, curves + (size_t)p * kmax * ncol + c, (size_t)ncol);
    // diagnostics of the boundary (surfdisp96.f:308-339): how many curves ended without a root, and the first of them (by curve number)
    if (kfail && diag && (gshift == 0 || (threadIdx.x & ((1u << gshift) - 1u)) == 0u)) {
        const unsigned long long slot = atomicAdd(diag, 1ull);
        atomicMin(diag + 1, ((unsigned long long)tid << 16) | (unsigned long long)kfail);
        if (fail_list && slot < (unsigned long long)fail_cap) fail_list[slot] = ((unsigned long long)tid << 8) | (unsigned long long)kfail;
    }
}

void launch_dispersion(int iwave, const LayerGeom* d_geom, const float* d_vels, int ncol, int npert, int igr, int kmax,
                       const double* d_t, float* d_ws, size_t nlanes, double* d_curves, int rmax, int layers_in_lds, int gshift,
                       unsigned long long* d_diag, unsigned long long* d_fail_list, int fail_cap, hipStream_t stream)
{
    const size_t n = ((size_t)ncol * npert) << gshift;
    if (n == 0) return;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    const int per = 64 >> gshift;
    size_t lds = layers_in_lds ? (size_t)4 * rmax * per * sizeof(float) : 0;
    if (gshift > 0) lds += 8 + (size_t)64 * 15 * sizeof(double);
    if (iwave == 1) hipLaunchKernelGGL(k_dispersion<1>, grid, block, lds, stream, d_geom, d_vels, ncol, npert, igr, kmax, d_t, d_ws, nlanes, d_curves, layers_in_lds, gshift, d_diag, d_fail_list, fail_cap);
    else hipLaunchKernelGGL(k_dispersion<2>, grid, block, lds, stream, d_geom, d_vels, ncol, npert, igr, kmax, d_t, d_ws, nlanes, d_curves, layers_in_lds, gshift, d_diag, d_fail_list, fail_cap);
}

// Host replay of one curve (kernels.h): the kernel's set-up of the column and its perturbation, the same layer table, the loop-nest form of the
// curve (the state machine of the Rayleigh kernel gives the same bits, dispersion_core.h) with the trace the reference's unit-66 block prints.
int disp_replay_failure(const LayerGeom& G, const float* vs_in, int pert, int iwave, int igr, int kmax, const double* t,
                        int* mmax, float* table, double* cc_cm_c1, double* c)
{
    float vs[kMaxDepths], vp[kMaxDepths], rho[kMaxDepths];
    for (int k = 0; k < G.nz; ++k) { vs[k] = vs_in[k]; brocher_vp_rho(vs[k], &vp[k], &rho[k]); }
    if (pert > 0) {
        const int idx = pert - 1, i = idx / 6, q = (idx % 6) >> 1, sgn = idx & 1;
        float* arr = q == 0 ? vs : (q == 1 ? vp : rho);
        const float base = arr[i];
        const float dln = 0.01f;
        arr[i] = sgn ? base + 0.5f * dln * base : base - 0.5f * dln * base;
    }
    Layers m;
    m.d = table; m.a = table + kMaxLayers; m.b = table + 2 * kMaxLayers; m.rho = table + 3 * kMaxLayers;
    m.stride = 1;
    double cg[kMaxPeriods];
    DispTrace tr{};
    int kfail;
    if (iwave == 1) { build_layers<1>(G, vs, vp, rho, m); kfail = dispersion_curve_nested<1>(m, igr, kmax, t, cg, 1, &tr); }
    else { build_layers<2>(G, vs, vp, rho, m); kfail = dispersion_curve_nested<2>(m, igr, kmax, t, cg, 1, &tr); }
    *mmax = m.mmax;
    cc_cm_c1[0] = tr.cc; cc_cm_c1[1] = tr.cm; cc_cm_c1[2] = tr.c1;
    for (int k = 0; k < kMaxPeriods; ++k) c[k] = k < kfail - 1 ? tr.c[k] : 0.0;
    return kfail;
}

// pv(c, k) = curve 0; sen_q(c, slot0 + k, i) = (cg(+) - cg(-)) / dble(dln * base_q(i)), CalSurfG.f90:76-150
__global__ void k_depth_kernels(const float* __restrict__ vels, int ncol, int nz, int kmax, const double* __restrict__ curves,
                                int with_kernels, double* __restrict__ pv, double* __restrict__ sen_vs, double* __restrict__ sen_vp,
                                double* __restrict__ sen_rho, int kmax_total, int slot0)
{
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)ncol * kmax) return;
    const int k = (int)(id / ncol), c = (int)(id - (size_t)k * ncol);
    pv[id] = curves[id];
    if (!with_kernels) return;
    const float dln = 0.01f;
    for (int i = 0; i < nz; ++i) {
        const float v = vels[(size_t)i * ncol + c];
        float base[3];
        base[0] = v;
        brocher_vp_rho(v, &base[1], &base[2]);
        double* out[3] = { sen_vs, sen_vp, sen_rho };
        for (int q = 0; q < 3; ++q) {
            const size_t pm = (size_t)(1 + 6 * i + 2 * q) * kmax * ncol + id;
            const double cg1 = curves[pm], cg2 = curves[pm + (size_t)kmax * ncol];
            out[q][((size_t)i * kmax_total + slot0 + k) * ncol + c] = (cg2 - cg1) / (double)(dln * base[q]);
        }
    }
}

void launch_depth_kernels(const float* d_vels, int ncol, int nz, int kmax, const double* d_curves, int with_kernels, double* d_pv,
                          double* d_sen_vs, double* d_sen_vp, double* d_sen_rho, int kmax_total, int slot0, hipStream_t stream)
{
    const size_t n = (size_t)ncol * kmax;
    if (n == 0) return;
    hipLaunchKernelGGL(k_depth_kernels, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_vels, ncol, nz, kmax, d_curves,
                       with_kernels, d_pv, d_sen_vs, d_sen_vp, d_sen_rho, kmax_total, slot0);
}

// velv = real(pv) for the maps of a call (CalSurfG.f90:1492)
__global__ void k_to_float(const double* __restrict__ in, float* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

void launch_to_float(const double* d_in, float* d_out, size_t n, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_to_float, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_in, d_out, n);
}

}  // namespace dsa
