// LSMR of the inversion step on the device-resident matrix (reference lsmrModule.f90:36-750 as shipped with
// DSurfTomo: single precision, local reorthogonalisation over the last `localSize` v's; called once per outer
// iteration at main.f90:487).  SURVEY.md 8f rank 1.
//
// Two placements of the vectors, same results (option "lsmr_device_vectors"):
//   0 (default)  the two matrix products run on the device (190 M entries at the headline size: HBM-bound, 1.2 ms
//                each); u and v cross PCIe once per product (1 MB + 0.5 MB) and the ordered reductions and the
//                element-wise updates run on the host.  Why: the reference's sums are serial fp32 chains, and one
//                dependent v_add_f32 costs ~18 cycles on a lone wavefront (tools/micro/ordered_sum_bench.hip:
//                7.4 ns per element at 2.4 GHz, readlane or DPP alike) against ~1 ns on a host core.  Measured at
//                the headline size: 27 ms per LSMR iteration with the vectors on the device, see DESIGN.md.
//   1            all vectors resident on the device, reductions by the single-wavefront kernels below.
//
// The reference's results depend on the order of its fp32 sums, so the order is kept:
//   * the two matrix products add every output element's entries in storage order (spmv.hip);
//   * dnrm2 (lsmrblas.f90:247-277) is a running (scale, ssq) recurrence and dot_product an in-order sum: one
//     wavefront forms the per-element terms in parallel -- the divisions and squares of dnrm2, the products of
//     the dot -- and only the final additions run as a serial chain, the running sum handed from lane to lane by a
//     DPP wave shift (wave_ordered_sum: one dependent v_add per element).  dnrm2's scale only changes at a new running maximum; a batch of 64
//     that contains one takes a scalar path, all others use the scale they start with;
//   * everything else is element-wise.
// The scalar recurrences (plane rotations, norm estimates, stopping rules) run on the host in fp32 exactly as
// written in the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"
#include "spmv_state.h"

namespace dsa {

namespace {

__device__ __forceinline__ float lane_value(float v, int i) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); }

// acc + t(0) + t(1) + ... + t(cnt-1), added strictly in that order (one rounding per addition), cnt in 1..64, wave-uniform.
// The running sum travels from lane to lane: `v_add_f32 v, v, t wave_shr:1` gives every lane "left neighbour's v + own
// t" (lane 0, which has no left neighbour, keeps its value), so after k steps lane k holds the correct prefix sum and
// keeps reproducing it.  One dependent DPP add per element (plus the two wait states a DPP read of a fresh VGPR
// needs) instead of a readlane / add pair through scalar registers.
__device__ __forceinline__ float wave_ordered_sum(float acc, float t, int cnt)
{
    float v = (threadIdx.x & 63) == 0 ? acc + t : t;
#pragma unroll
    for (int k = 1; k < 64; ++k)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(t));
    return __shfl(v, cnt - 1);
}

__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

constexpr int kBatch = 16;      // 64-element batches in flight per step: the loads of the next 1024 elements overlap this step's chain (~3 us)

// out[0] = dnrm2(n, x, 1), lsmrblas.f90:247-277.  One wavefront.
__global__ __launch_bounds__(64) void k_nrm2(int n, const float* __restrict__ x, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    if (n < 1) { if (lane == 0) out[0] = 0.0f; return; }
    if (n == 1) { if (lane == 0) out[0] = fabsf(x[0]); return; }
    float scale = 0.0f, ssq = 1.0f;                    // wave-uniform
    float cur[kBatch], nxt[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) { const int i = u * 64 + lane; cur[u] = i < n ? fabsf(x[i]) : 0.0f; }
    for (int base = 0; base < n; base += 64 * kBatch) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { const int i = base + 64 * kBatch + u * 64 + lane; nxt[u] = i < n ? fabsf(x[i]) : 0.0f; }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const float a = cur[u];                    // |x(i)|; zero for skipped (zero) elements and beyond n
            const float bmax = wave_max(a);
            if (bmax == 0.0f) continue;                // nothing but zeros: the reference skips them
            if (bmax <= scale) {
                // no new maximum in this batch (ties take the reference's else branch too): ssq += (|x|/scale)**2,
                // and a zero element adds +0 to ssq >= 1, which changes nothing
                const float q = a / scale;
                const float t = q * q;
                ssq = wave_ordered_sum(ssq, t, 64);
            } else {
                for (int i = 0; i < 64; ++i) {
                    const float xi = lane_value(a, i);
                    if (xi != 0.0f) {
                        if (scale < xi) { const float q = scale / xi; ssq = 1.0f + ssq * (q * q); scale = xi; }
                        else { const float q = xi / scale; ssq = ssq + q * q; }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) cur[u] = nxt[u];
    }
    if (lane == 0) out[0] = scale * sqrtf(ssq);
}

// out[0] = sum over i of a(i) * b(i), added in order from 0 (the reference's inlined dot_product, lsmrModule.f90:744)
__global__ __launch_bounds__(64) void k_dot(int n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out)
{
    const int lane = threadIdx.x;
    float acc = 0.0f;
    float cur[kBatch], nxt[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) { const int i = u * 64 + lane; cur[u] = i < n ? a[i] * b[i] : 0.0f; }
    for (int base = 0; base < n; base += 64 * kBatch) {
#pragma unroll
        for (int u = 0; u < kBatch; ++u) { const int i = base + 64 * kBatch + u * 64 + lane; nxt[u] = i < n ? a[i] * b[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int left = n - (base + u * 64);
            if (left > 0) acc = wave_ordered_sum(acc, cur[u], left < 64 ? left : 64);
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) cur[u] = nxt[u];
    }
    if (lane == 0) out[0] = acc;
}

__global__ void k_scal(int n, float sa, float* __restrict__ x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = sa * x[i];
}
// v = v - d * lv, d from device memory (localVOrtho, lsmrModule.f90:745)
__global__ void k_axmy(int n, const float* __restrict__ d, const float* __restrict__ lv, float* __restrict__ v)
{
    const float dd = d[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = v[i] - dd * lv[i];
}
// lsmrModule.f90:545-547: hbar = h - c1*hbar; x = x + c2*hbar; h = v - c3*h
__global__ void k_update(int n, float c1, float c2, float c3, const float* __restrict__ v, float* __restrict__ h,
                         float* __restrict__ hbar, float* __restrict__ x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float hb = h[i] - c1 * hbar[i];
        hbar[i] = hb;
        x[i] = x[i] + c2 * hb;
        h[i] = v[i] - c3 * h[i];
    }
}

// lsmrModule.f90:686-711
float d2norm(float a, float b)
{
    const float scale = fabsf(a) + fabsf(b);
    if (scale == 0.0f) return 0.0f;
    const float p = a / scale, q = b / scale;
    return scale * sqrtf(p * p + q * q);
}

#define LS_TRY(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) { (e)->fail(DSA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_r)); return DSA_ERR_DEVICE; } \
    } while (0)

}  // namespace

}  // namespace dsa

using dsa::Engine;
using dsa::SpmvState;

namespace {

// The vector side of LSMR.  Every method returns 0 or an error code (engine status set).
struct DeviceVectors {
    Engine* e; SpmvState& S; int m, n, localVecs; hipStream_t st;
    float host_scalar = 0.0f;
    DeviceVectors(Engine* e_, SpmvState& S_, int lv) : e(e_), S(S_), m(S_.m), n(S_.n), localVecs(lv), st(e_->stream) {}
    static dim3 grid(int len) { return dim3((unsigned)std::min(4096, std::max(1, (len + 255) / 256))); }
    int setup(const float* b)
    {
        if (e->ensure(S.u, (size_t)m) || e->ensure(S.v, (size_t)n) || e->ensure(S.h, (size_t)n) || e->ensure(S.hbar, (size_t)n) ||
            e->ensure(S.xs, (size_t)n) || e->ensure(S.localV, std::max<size_t>((size_t)n * (size_t)localVecs, 1)) || e->ensure(S.scal, 16)) return e->status;
        LS_TRY(e, hipMemcpyAsync(S.u.p, b, (size_t)m * 4, hipMemcpyHostToDevice, st));            // u = b, v = 0, x = 0 (:383-385)
        LS_TRY(e, hipMemsetAsync(S.v.p, 0, (size_t)n * 4, st));
        LS_TRY(e, hipMemsetAsync(S.xs.p, 0, (size_t)n * 4, st));
        LS_TRY(e, hipMemsetAsync(S.hbar.p, 0, (size_t)n * 4, st));
        return 0;
    }
    int nrm2(int len, const float* p, float* out)
    {
        hipLaunchKernelGGL(dsa::k_nrm2, dim3(1), dim3(64), 0, st, len, p, S.scal.p);
        if (hipMemcpyAsync(&host_scalar, S.scal.p, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            e->fail(DSA_ERR_DEVICE, "lsmr: norm failed: %s", hipGetErrorString(hipGetLastError())); return DSA_ERR_DEVICE; }
        *out = host_scalar;
        return 0;
    }
    int norm_u(float* out) { return nrm2(m, S.u.p, out); }
    int norm_v(float* out) { return nrm2(n, S.v.p, out); }
    int norm_x(float* out) { return nrm2(n, S.xs.p, out); }
    int scal_u(float a) { hipLaunchKernelGGL(dsa::k_scal, grid(m), dim3(256), 0, st, m, a, S.u.p); return 0; }
    int scal_v(float a) { hipLaunchKernelGGL(dsa::k_scal, grid(n), dim3(256), 0, st, n, a, S.v.p); return 0; }
    int aprod1() { dsa::spmv_device(e, 1, S.v.p, S.u.p); return 0; }                               // u += A v
    int aprod2() { dsa::spmv_device(e, 2, S.v.p, S.u.p); return 0; }                               // v += A'u
    int enqueue(int slot) { LS_TRY(e, hipMemcpyAsync(S.localV.p + (size_t)slot * (size_t)n, S.v.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st)); return 0; }
    int ortho(int lim)                                                                             // localVOrtho, :731-748
    {
        for (int k = 0; k < lim; ++k) {
            const float* lv = S.localV.p + (size_t)k * (size_t)n;
            hipLaunchKernelGGL(dsa::k_dot, dim3(1), dim3(64), 0, st, n, (const float*)S.v.p, lv, S.scal.p + 1);
            hipLaunchKernelGGL(dsa::k_axmy, grid(n), dim3(256), 0, st, n, (const float*)(S.scal.p + 1), lv, S.v.p);
        }
        return 0;
    }
    int h_from_v() { LS_TRY(e, hipMemcpyAsync(S.h.p, S.v.p, (size_t)n * 4, hipMemcpyDeviceToDevice, st)); return 0; }
    int update(float c1, float c2, float c3)
    {
        hipLaunchKernelGGL(dsa::k_update, grid(n), dim3(256), 0, st, n, c1, c2, c3, (const float*)S.v.p, S.h.p, S.hbar.p, S.xs.p);
        return 0;
    }
    int fetch_x(float* x)
    {
        LS_TRY(e, hipMemcpyAsync(x, S.xs.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        LS_TRY(e, hipGetLastError());
        LS_TRY(e, hipStreamSynchronize(st));
        return 0;
    }
};

// lsmrblas.f90:247-277 on the host
float host_nrm2(int n, const float* x)
{
    if (n < 1) return 0.0f;
    if (n == 1) return fabsf(x[0]);
    float scale = 0.0f, ssq = 1.0f;
    for (int i = 0; i < n; ++i) {
        const float a = fabsf(x[i]);
        if (a == 0.0f) continue;
        if (scale < a) { const float q = scale / a; ssq = 1.0f + ssq * (q * q); scale = a; }
        else { const float q = a / scale; ssq = ssq + q * q; }
    }
    return scale * sqrtf(ssq);
}

struct HostVectors {
    Engine* e; SpmvState& S; int m, n, localVecs; hipStream_t st;
    float* u = nullptr; float* v = nullptr;        // pinned (SpmvState): they cross PCIe with every product
    std::vector<float> h, hbar, xs, localV;
    HostVectors(Engine* e_, SpmvState& S_, int lv) : e(e_), S(S_), m(S_.m), n(S_.n), localVecs(lv), st(e_->stream) {}
    int pinned(float** p, size_t* cap, size_t need)
    {
        if (*cap >= need) return 0;
        if (*p) (void)hipHostFree(*p);
        *p = nullptr; *cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(p), need * sizeof(float), hipHostMallocDefault) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "lsmr: pinned host allocation of %zu floats failed", need); return DSA_ERR_DEVICE; }
        *cap = need;
        return 0;
    }
    int setup(const float* b)
    {
        if (e->ensure(S.u, (size_t)m) || e->ensure(S.v, (size_t)n)) return e->status;
        if (pinned(&S.hu, &S.hu_cap, (size_t)m) || pinned(&S.hv, &S.hv_cap, (size_t)n)) return e->status;
        u = S.hu; v = S.hv;
        std::memcpy(u, b, (size_t)m * 4); std::fill(v, v + n, 0.0f);
        h.assign((size_t)n, 0.0f); hbar.assign((size_t)n, 0.0f); xs.assign((size_t)n, 0.0f);
        localV.resize((size_t)n * (size_t)localVecs);
        return 0;
    }
    int norm_u(float* out) { *out = host_nrm2(m, u); return 0; }
    int norm_v(float* out) { *out = host_nrm2(n, v); return 0; }
    int norm_x(float* out) { *out = host_nrm2(n, xs.data()); return 0; }
    int scal_u(float a) { for (int i = 0; i < m; ++i) u[i] = a * u[i]; return 0; }
    int scal_v(float a) { for (int i = 0; i < n; ++i) v[i] = a * v[i]; return 0; }
    int product(int mode)
    {
        LS_TRY(e, hipMemcpyAsync(S.u.p, u, (size_t)m * 4, hipMemcpyHostToDevice, st));
        LS_TRY(e, hipMemcpyAsync(S.v.p, v, (size_t)n * 4, hipMemcpyHostToDevice, st));
        dsa::spmv_device(e, mode, S.v.p, S.u.p);
        if (mode == 1) LS_TRY(e, hipMemcpyAsync(u, S.u.p, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        else LS_TRY(e, hipMemcpyAsync(v, S.v.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        LS_TRY(e, hipGetLastError());
        LS_TRY(e, hipStreamSynchronize(st));
        return 0;
    }
    int aprod1() { return product(1); }
    int aprod2() { return product(2); }
    int enqueue(int slot) { std::memcpy(localV.data() + (size_t)slot * (size_t)n, v, (size_t)n * 4); return 0; }
    int ortho(int lim)
    {
        for (int k = 0; k < lim; ++k) {
            const float* lv = localV.data() + (size_t)k * (size_t)n;
            float d = 0.0f;
            for (int i = 0; i < n; ++i) d = d + v[i] * lv[i];                                      // in order (dot_product, :744)
            for (int i = 0; i < n; ++i) v[i] = v[i] - d * lv[i];
        }
        return 0;
    }
    int h_from_v() { h.assign(v, v + n); return 0; }
    int update(float c1, float c2, float c3)
    {
        for (int i = 0; i < n; ++i) {
            const float hb = h[i] - c1 * hbar[i];
            hbar[i] = hb;
            xs[i] = xs[i] + c2 * hb;
            h[i] = v[i] - c3 * h[i];
        }
        return 0;
    }
    int fetch_x(float* x) { std::memcpy(x, xs.data(), (size_t)n * 4); return 0; }
};

#define LS_DO(call) do { if ((rc = (call)) != 0) return rc; } while (0)

// lsmrModule.f90:380-651 over a vector backend V
template <class V>
int lsmr_loop(V& W, const float* b, float damp, float atol, float btol, float conlim, int itnlim, int localVecs,
              float* x, int* istop, int* itn, float* normA, float* condA, float* normr, float* normAr, float* normx)
{
    int rc = 0;
    LS_DO(W.setup(b));
    float alpha = 0.0f, beta = 0.0f;
    LS_DO(W.norm_u(&beta));
    if (beta > 0.0f) {
        LS_DO(W.scal_u(1.0f / beta));
        LS_DO(W.aprod2());                                                                       // v = A'u
        LS_DO(W.norm_v(&alpha));
    }
    if (alpha > 0.0f) LS_DO(W.scal_v(1.0f / alpha));
    *itn = 0; *istop = 0; *normA = 0.0f; *condA = 0.0f; *normx = 0.0f;
    *normr = beta;
    *normAr = alpha * beta;
    const bool damped = damp > 0.0f;
    if (*normAr != 0.0f) {
        bool localOrtho = false, localVQueueFull = false;
        int localPointer = 0;
        if (localVecs > 0) {                                                                     // :408-413
            localPointer = 1; localOrtho = true;
            LS_DO(W.enqueue(0));
        }
        float zetabar = alpha * beta, alphabar = alpha, rho = 1.0f, rhobar = 1.0f, cbar = 1.0f, sbar = 0.0f;
        LS_DO(W.h_from_v());
        float betadd = beta, betad = 0.0f, rhodold = 1.0f, tautildeold = 0.0f, thetatilde = 0.0f, zeta = 0.0f, d = 0.0f;
        float normA2 = alpha * alpha, maxrbar = 0.0f, minrbar = 1e+30f;
        const float normb = beta;
        float ctol = 0.0f;
        if (conlim > 0.0f) ctol = 1.0f / conlim;
        for (;;) {                                                                               // :480
            *itn += 1;
            LS_DO(W.scal_u(-alpha));
            LS_DO(W.aprod1());                                                                   // u = A v - alpha u
            LS_DO(W.norm_u(&beta));
            if (beta > 0.0f) {
                LS_DO(W.scal_u(1.0f / beta));
                if (localOrtho) {                                                                // localVEnqueue, :715-727
                    if (localPointer < localVecs) localPointer += 1;
                    else { localPointer = 1; localVQueueFull = true; }
                    LS_DO(W.enqueue(localPointer - 1));
                }
                LS_DO(W.scal_v(-beta));
                LS_DO(W.aprod2());                                                               // v = A'u - beta v
                if (localOrtho) LS_DO(W.ortho(localVQueueFull ? localVecs : localPointer));
                LS_DO(W.norm_v(&alpha));
                if (alpha > 0.0f) LS_DO(W.scal_v(1.0f / alpha));
            }
            // plane rotations and estimates, :516-600, in the reference's order
            const float alphahat = dsa::d2norm(alphabar, damp);
            const float chat = alphabar / alphahat, shat = damp / alphahat;
            const float rhoold = rho;
            rho = dsa::d2norm(alphahat, beta);
            const float c = alphahat / rho, s = beta / rho;
            const float thetanew = s * alpha;
            alphabar = c * alpha;
            const float rhobarold = rhobar, zetaold = zeta;
            const float thetabar = sbar * rho, rhotemp = cbar * rho;
            rhobar = dsa::d2norm(cbar * rho, thetanew);
            cbar = cbar * rho / rhobar;
            sbar = thetanew / rhobar;
            zeta = cbar * zetabar;
            zetabar = -sbar * zetabar;
            LS_DO(W.update(thetabar * rho / (rhoold * rhobarold), zeta / (rho * rhobar), thetanew / rho));   // :545-547
            const float betaacute = chat * betadd, betacheck = -shat * betadd;
            const float betahat = c * betaacute;
            betadd = -s * betaacute;
            const float thetatildeold = thetatilde;
            const float rhotildeold = dsa::d2norm(rhodold, thetabar);
            const float ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
            thetatilde = stildeold * rhobar;
            rhodold = ctildeold * rhobar;
            betad = -stildeold * betad + ctildeold * betahat;
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
            const float taud = (zeta - thetatilde * tautildeold) / rhodold;
            d = d + betacheck * betacheck;
            {
                const float e1 = betad - taud;
                *normr = sqrtf(d + e1 * e1 + betadd * betadd);
            }
            normA2 = normA2 + beta * beta;
            *normA = sqrtf(normA2);
            normA2 = normA2 + alpha * alpha;
            maxrbar = maxrbar > rhobarold ? maxrbar : rhobarold;
            if (*itn > 1) minrbar = minrbar < rhobarold ? minrbar : rhobarold;
            *condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
            *normAr = fabsf(zetabar);
            LS_DO(W.norm_x(normx));
            const float test1 = *normr / normb;
            const float test2 = *normAr / (*normA * *normr);
            const float test3 = 1.0f / *condA;
            const float t1 = test1 / (1.0f + *normA * *normx / normb);
            const float rtol = btol + atol * *normA * *normx / normb;
            if (*itn >= itnlim) *istop = 7;                                                      // :607-613
            if (1.0f + test3 <= 1.0f) *istop = 6;
            if (1.0f + test2 <= 1.0f) *istop = 5;
            if (1.0f + t1 <= 1.0f) *istop = 4;
            if (test3 <= ctol) *istop = 3;
            if (test2 <= atol) *istop = 2;
            if (test1 <= rtol) *istop = 1;
            if (*istop != 0) break;
        }
    }
    if (damped && *istop == 2) *istop = 3;                                                       // :651
    return W.fetch_x(x);
}

}  // namespace

extern "C" {

// LSMR on the matrix of the last dsa_spmv_load.  Arguments and results as lsmrModule.f90:36-50 (without the matrix,
// which is resident, and without nout).  b: m values (host, not modified), x: n values (host, out).
int dsa_lsmr(dsa_engine* h_, const float* b, float damp, float atol, float btol, float conlim, int itnlim, int localSize,
             float* x, int* istop, int* itn, float* normA, float* condA, float* normr, float* normAr, float* normx)
{
    if (!h_) return DSA_ERR_ARGUMENT;
    Engine* e = reinterpret_cast<Engine*>(h_);
    if (!e->spmv) { e->fail(DSA_ERR_STATE, "lsmr: call dsa_spmv_load first"); return DSA_ERR_STATE; }
    if (!b || !x || !istop || !itn || !normA || !condA || !normr || !normAr || !normx) { e->fail(DSA_ERR_ARGUMENT, "lsmr: null argument"); return DSA_ERR_ARGUMENT; }
    SpmvState& S = *e->spmv;
    const int localVecs = std::max(0, std::min(localSize, std::min(S.m, S.n)));                  // :365
    LS_TRY(e, hipSetDevice(e->device));
    if (e->lsmr_device_vectors) {
        DeviceVectors W(e, S, localVecs);
        return lsmr_loop(W, b, damp, atol, btol, conlim, itnlim, localVecs, x, istop, itn, normA, condA, normr, normAr, normx);
    }
    HostVectors W(e, S, localVecs);
    return lsmr_loop(W, b, damp, atol, btol, conlim, itnlim, localVecs, x, istop, itn, normA, condA, normr, normAr, normx);
}

}  // extern "C"
