// K5 / K6: ray back-trace and Frechet row assembly (reference rpaths, CalSurfG.f90:1771-2318, and the
// row loop :1383-1432).
//
// K5 runs one ray per lane: a ray is a serial chain of ~2 steps per node crossed with data-dependent
// addresses, and a call has tens of thousands of them.  Each ray owns a dense slab of vertex sums in
// HBM (the reference's fdm array) that it touches once per vertex-cell crossing (ray_core.h).
// K6 runs one wavefront per ray: lanes walk the slab in the reference's column order, compact with
// ballot / popcount, and write COO entries at offsets from a prefix sum over rays, so the output is
// in the reference's order without atomics.
#include "kernels.h"
#include "ray_core.h"

namespace dsa {

template <int LPR>
__global__ __launch_bounds__(64) void k_rays(GridDesc g, BatchPtrs b, int unit_base, const RayDesc* __restrict__ rays,
                                             const int* __restrict__ trace_ids, int n, const float* __restrict__ veln_all,
                                             size_t field_stride, float dpl, float* __restrict__ slabs, size_t slab_stride,
                                             int32_t* __restrict__ rayinfo, int32_t* __restrict__ err,
                                             float* __restrict__ paths, int path_cap, int* __restrict__ path_n)
{
    const int lane_id = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = lane_id / LPR, sub = lane_id % LPR;          // LPR = 4: four lanes trace ray t together (ray_core.h: PatchAcc)
    if (t >= n) return;
    const int r = trace_ids[t];
    const RayDesc rd = rays[r];
    const int slot = rd.src - unit_base;
    const SourceDesc sd = b.src[slot];
    const size_t rr = (size_t)kRefMax * kRefMax;
    RayFields f;
    f.F = b.T_c + (size_t)slot * g.nbx * g.nbz * kTileRecs;
    f.veln = veln_all + (size_t)sd.period * field_stride;
    f.Tr = b.Tfin_r + slot * rr;
    f.Sr = b.S_r + slot * rr;
    int flags = 0, steps = 0;
    RayPath path;
    if (paths && sub == 0) { path.pts = paths + (size_t)t * (size_t)path_cap * 2; path.cap = path_cap; }
    const int rc = trace_ray<LPR>(g, sd, f, rd.rx, rd.rz, dpl, slabs + (size_t)t * slab_stride, &flags, &steps, paths ? &path : nullptr, sub);
    if (sub != 0) return;
    if (paths) path_n[t] = path.n;
    if (rc != 0) atomicExch(err, r + 1);
    rayinfo[2 * t] = flags;
    rayinfo[2 * t + 1] = steps;
}

void launch_rays(const GridDesc& g, const BatchPtrs& b, int unit_base, const RayDesc* d_rays, const int* d_trace_ids, int n,
                 const float* d_veln_all, size_t field_stride, float dpl, float* d_slabs, size_t slab_stride,
                 int32_t* d_rayinfo, int32_t* d_err, float* d_paths, int path_cap, int* d_path_n, hipStream_t stream, int lanes_per_ray)
{
    if (n <= 0) return;
    // A ray is a chain of ~2 steps per node crossed, ~3 000 instructions a step of which the sixteen vertex sums are half: a small launch (one
    // wavefront or less per SIMD) is bound by that chain, and four lanes per ray shorten it; a large one is bound by the instructions issued, of
    // which four lanes per ray issue three times as many per ray (profiles/r05_ab_rays.log)
    if (lanes_per_ray == 4)
        hipLaunchKernelGGL(k_rays<4>, dim3((unsigned)(((size_t)n * 4 + 63) / 64)), dim3(64), 0, stream, g, b, unit_base, d_rays, d_trace_ids, n, d_veln_all,
                           field_stride, dpl, d_slabs, slab_stride, d_rayinfo, d_err, d_paths, path_cap, d_path_n);
    else
        hipLaunchKernelGGL(k_rays<1>, dim3((n + 63) / 64), dim3(64), 0, stream, g, b, unit_base, d_rays, d_trace_ids, n, d_veln_all,
                           field_stride, dpl, d_slabs, slab_stride, d_rayinfo, d_err, d_paths, path_cap, d_path_n);
}

// ---------------------------------------------------------------------------------------------
__global__ void k_sen_combine(int ncol, int kmax, int nz, const float* __restrict__ vels, const double* __restrict__ sen_vs,
                              const double* __restrict__ sen_vp, const double* __restrict__ sen_rho, int shallow,
                              double* __restrict__ S)
{
    const size_t n = (size_t)ncol * kmax * (nz - 1);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % ncol);
    const int k = (int)(i / ((size_t)ncol * kmax));
    float a, r;
    brocher_chain(vels[(size_t)k * ncol + c], shallow != 0, &a, &r);
    S[i] = (sen_vp[i] * (double)a + sen_rho[i] * (double)r) + sen_vs[i];
}

void launch_sen_combine(int ncol, int kmax, int nz, const float* d_vels, const double* d_sen_vs, const double* d_sen_vp,
                        const double* d_sen_rho, int shallow, double* d_S, hipStream_t stream)
{
    const size_t n = (size_t)ncol * kmax * (nz - 1);
    if (n == 0) return;
    hipLaunchKernelGGL(k_sen_combine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ncol, kmax, nz, d_vels, d_sen_vs,
                       d_sen_vp, d_sen_rho, shallow, d_S);
}

// ---------------------------------------------------------------------------------------------
namespace {
constexpr float kFtol = 1e-4f;     // reference ftol (CalSurfG.f90:1031)
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull));
}
}  // namespace

// interior vertices with |fdm| >= ftol, in the reference's loop order (jj = 1..nvz outer, kk = 1..nvx inner)
__global__ __launch_bounds__(256) void k_row_list(GridDesc g, RowArgs a)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= a.n) return;
    const float* slab = a.slabs + (size_t)t * a.slab_stride;
    int* vl = a.vlist + (size_t)t * a.vlist_stride;
    const int nvx = g.nvx, nint = g.nvx * g.nvz, ldx = g.nvx + 2;
    int n = 0;
    for (int base = 0; base < nint; base += 64) {
        const int i = base + lane;
        bool keep = false;
        if (i < nint) {
            const int jj = i / nvx + 1, kk = i - (jj - 1) * nvx + 1;
            keep = fabsf(slab[(size_t)jj * ldx + kk]) >= kFtol;
        }
        const unsigned long long m = __ballot(keep);
        if (keep) vl[n + lanes_below(m)] = i;
        n += __popcll(m);
    }
    if (lane == 0) a.nv[t] = n;
}

void launch_row_list(const GridDesc& g, const RowArgs& a, hipStream_t stream)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_row_list, dim3((a.n + 3) / 4), dim3(256), 0, stream, g, a);
}

// row(n) = real(S * fdm) for the kept vertices, layers outer; entries with |row| > ftol survive
template <bool WRITE>
__global__ __launch_bounds__(256) void k_row_emit(GridDesc g, RowArgs a)
{
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= a.n) return;
    const RayDesc rd = a.rays[a.trace_ids[t]];
    const int slot = a.src[rd.src - a.unit_base].sen_slot;
    const float* slab = a.slabs + (size_t)t * a.slab_stride;
    const int* vl = a.vlist + (size_t)t * a.vlist_stride;
    const int nv = a.nv[t];
    const int nvx = g.nvx, ldx = g.nvx + 2, ncol = g.nx * g.ny, layer = g.nvx * g.nvz;
    const long long off = WRITE ? a.offsets[t] : 0;
    int cnt = 0;
    for (int k = 0; k < a.nz - 1; ++k) {
        const double* Sk = a.S + ((size_t)k * a.kmax + slot) * ncol;
        for (int base = 0; base < nv; base += 64) {
            const int e = base + lane;
            bool keep = false;
            float val = 0.0f;
            int i = 0;
            if (e < nv) {
                i = vl[e];
                const int jj = i / nvx + 1, kk = i - (jj - 1) * nvx + 1;
                const float f = slab[(size_t)jj * ldx + kk];
                val = (float)(Sk[jj * g.nx + kk] * (double)f);
                keep = fabsf(val) > kFtol;
            }
            const unsigned long long m = __ballot(keep);
            if (WRITE && keep) {
                const long long p = off + cnt + lanes_below(m);
                a.rw[p] = val;
                a.iw[p] = rd.data + 1;
                a.col[p] = k * layer + i + 1;
            }
            cnt += __popcll(m);
        }
    }
    if (!WRITE && lane == 0) a.counts[t] = cnt;
}

void launch_row_emit(const GridDesc& g, const RowArgs& a, bool write, hipStream_t stream)
{
    if (a.n <= 0) return;
    if (write) hipLaunchKernelGGL(k_row_emit<true>, dim3((a.n + 3) / 4), dim3(256), 0, stream, g, a);
    else hipLaunchKernelGGL(k_row_emit<false>, dim3((a.n + 3) / 4), dim3(256), 0, stream, g, a);
}

// one workgroup: per-thread segment sums, scan of the 1024 sums in LDS, segment rewrite
__global__ __launch_bounds__(1024) void k_scan(const int* __restrict__ counts, int n, long long* __restrict__ offsets)
{
    __shared__ long long s_sum[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    s_sum[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = tid >= d ? s_sum[tid - d] : 0;
        __syncthreads();
        s_sum[tid] += v;
        __syncthreads();
    }
    long long run = s_sum[tid] - s;
    for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
    if (tid == 1023) offsets[n] = s_sum[1023];
}

void launch_scan(const int* d_counts, int n, long long* d_offsets, hipStream_t stream)
{
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, stream, d_counts, n, d_offsets);
}

}  // namespace dsa
