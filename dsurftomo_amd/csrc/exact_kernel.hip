// K3x: exact mode of the eikonal solve -- the reference's Fast Marching replayed literally, one wavefront per unit
// (exact_march.h; reference CalSurfG.f90:288-487 `travel`, :587-759 `fouds2`, :768-921 the tree, :1287-1349 the hand-off).
// The engine runs it for the units whose fixed-point solve met an exact time tie (option exact_ties = 1) or for all of them
// (exact_ties = 2); it replaces the unit's refined snapshot (Tfin_r, S_r) and its compact coarse field with the march's own.
#include "kernels.h"

#include "exact_march.h"

namespace dsa {

__global__ __launch_bounds__(64) void k_exact(GridDesc g, BatchPtrs b, const int* __restrict__ units, int n,
                                             const float* __restrict__ slow_all, size_t field_stride, const float* __restrict__ risti_c,
                                             XRec* pool, size_t pool_stride, XEntry* heap_pool, int gcap, int lcap, int32_t* xinfo)
{
    extern __shared__ unsigned char x_lds[];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot >= n) return;
    const int s = units[slot];
    const SourceDesc sd = b.src[s];
    DSA_LDS XEntry* const hl = (DSA_LDS XEntry*)x_lds;                         // slots 1..lcap
    DSA_LDS XLog* const log = (DSA_LDS XLog*)(hl + lcap + 1);
    DSA_LDS int* const stage_st = (DSA_LDS int*)(log + kXLogCap);
    DSA_LDS float* const stage_T = (DSA_LDS float*)(stage_st + kXStage);
    const size_t rr = (size_t)kRefMax * kRefMax;

    XMarch m;
    m.hl = hl; m.lcap = lcap; m.hg = heap_pool + (size_t)slot * gcap; m.gcap = gcap; m.log = log;
    m.ntr = 0; m.error = 0; m.nlog = 0; m.pops = 0u;
    m.ri = g.earth;

    // ---- refined stage: travel(urg = 1) on the box
    XRec* const Fr = (XRec*)(b.F_r + (size_t)s * kRefRecs);
    for (int i = lane; i < kRefRecs; i += 64) Fr[i] = XRec{ 0.0f, -1 };
    __threadfence_block();
    m.F = Fr; m.slow = b.slow_r + (size_t)s * kRefRecs; m.risti = b.risti_r + (size_t)s * kRefMax;
    x_set_grid(m, sd.nbz_r, sd.rnx, sd.rnz); m.dnx = sd.rdnx; m.dnz = sd.rdnz;
    x_refined_start(m, sd, b.vcorner + (size_t)s * 4);
    x_march<true>(m, sd);
    __threadfence_block();
    const unsigned pops_r = m.pops;
    int err = m.error;
    // the snapshot the ray tracer reads (reference ttnr / nstsr, :1287-1288) and the hand-off: every sgdl-th refined node, status
    // and -- for status >= 0 -- value, onto the propagation grid (:1293-1303)
    float* const Tfin = b.Tfin_r + (size_t)s * rr;
    int8_t* const Sr = b.S_r + (size_t)s * rr;
    const int nref = sd.rnx * sd.rnz;
    for (int id = lane; id < nref; id += 64) {
        const int ix = id / sd.rnz, iz = id - ix * sd.rnz;
        const XRec r = Fr[rec_index(sd.nbz_r, iz, ix)];
        Sr[id] = (int8_t)(r.st < 0 ? -1 : r.st == 0 ? 0 : 1);
        Tfin[id] = r.st >= 0 ? r.T : kInf;
    }
    const int bxn = (sd.rnx - 1) / kSgdl + 1, bzn = (sd.rnz - 1) / kSgdl + 1;
    for (int q = lane; q < bxn * bzn; q += 64) {
        const int l = (q / bzn) * kSgdl, k = (q - (q / bzn) * bzn) * kSgdl;      // 0-based refined node
        const XRec r = Fr[rec_index(sd.nbz_r, k, l)];
        stage_st[q] = r.st < 0 ? -1 : r.st == 0 ? 0 : 1;
        stage_T[q] = r.T;
    }
    __syncthreads();
    // alive nodes that touch a far node go back into the narrow band (:1332-1349); nodes outside the box are far
    unsigned promote = 0u;                                                      // bit t: this lane's t-th node
    for (int q = lane, t = 0; q < bxn * bzn; q += 64, ++t) {
        if (stage_st[q] != 0) continue;
        const int bx = q / bzn, bz = q - bx * bzn;
        const int cx = sd.vnl + bx, cz = sd.vnt + bz;                           // 1-based node of the propagation grid
        const int dx[4] = { -1, 1, 0, 0 }, dz[4] = { 0, 0, -1, 1 };
        for (int d = 0; d < 4; ++d) {
            const int nx = cx + dx[d], nz = cz + dz[d];
            if (nx < 1 || nx > g.nnx || nz < 1 || nz > g.nnz) continue;
            const int ox = bx + dx[d], oz = bz + dz[d];
            const bool inbox = ox >= 0 && ox < bxn && oz >= 0 && oz < bzn;
            if (!inbox || stage_st[ox * bzn + oz] == -1) promote |= 1u << t;
        }
    }
    __syncthreads();
    for (int q = lane, t = 0; q < bxn * bzn; q += 64, ++t) if ((promote >> t) & 1u) stage_st[q] = 1;
    __syncthreads();

    // ---- coarse stage: travel(urg = 2) from the injected state
    XRec* const Fc = pool + (size_t)slot * pool_stride;
    const int nrec = g.nbx * g.nbz * kTileRecs;
    for (int i = lane; i < nrec; i += 64) Fc[i] = XRec{ 0.0f, -1 };
    __threadfence_block();
    m.F = Fc; m.slow = slow_all + (size_t)sd.period * field_stride; m.risti = risti_c;
    x_set_grid(m, g.nbz, g.nnx, g.nnz); m.dnx = g.dnx; m.dnz = g.dnz;
    m.ntr = 0; m.nlog = 0; m.pops = 0u;
    for (int q = lane; q < bxn * bzn; q += 64)
        if (stage_st[q] == 0) {
            const int bx = q / bzn, bz = q - bx * bzn;
            Fc[rec_index(g.nbz, sd.vnt + bz - 1, sd.vnl + bx - 1)] = XRec{ stage_T[q], 0 };
        }
    __threadfence_block();
    // tree start in the reference's scan order: ix outer, iz inner (:341-347)
    for (int q = 0; q < bxn * bzn; ++q) {
        if (x_uni(stage_st[q]) <= 0) continue;
        const int bx = q / bzn, bz = q - bx * bzn;
        const int id = rec_index(g.nbz, sd.vnt + bz - 1, sd.vnl + bx - 1);
        const float tq = x_unif(stage_T[q]);
        if (lane == 0) Fc[id].T = tq;
        x_add(m, id, tq);
    }
    x_march<false>(m, sd);
    __threadfence_block();
    err = err ? err : m.error;
    // the unit's compact coarse field: plain values, no exceptional nodes (every node was accepted once, in order)
    float* const T_c = b.T_c + (size_t)s * g.nbx * g.nbz * kTileRecs;
    for (int i = lane; i < nrec; i += 64) {
        const XRec r = Fc[i];
        T_c[i] = r.st == 0 ? r.T : kInf;
    }
    if (lane == 0) { xinfo[4 * s + 0] = (int)pops_r; xinfo[4 * s + 1] = (int)m.pops; xinfo[4 * s + 2] = err; xinfo[4 * s + 3] = 0; }
}

size_t exact_lds_bytes(int lcap) { return (size_t)(lcap + 1) * sizeof(XEntry) + (size_t)kXLogCap * sizeof(XLog) + (size_t)kXStage * 8; }

void launch_exact(const GridDesc& g, const BatchPtrs& b, const int* d_units, int n, const float* d_slow_all, size_t field_stride,
                  const float* d_risti_c, void* d_pool, size_t pool_stride, void* d_heap_pool, int gcap, int lcap, int32_t* d_xinfo,
                  hipStream_t stream)
{
    if (n <= 0) return;
    const size_t lds = exact_lds_bytes(lcap);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // (per device: set every time)
    hipLaunchKernelGGL(k_exact, dim3(n), dim3(64), lds, stream, g, b, d_units, n, d_slow_all, field_stride, d_risti_c,
                       (XRec*)d_pool, pool_stride, (XEntry*)d_heap_pool, gcap, lcap, d_xinfo);
}

}  // namespace dsa
