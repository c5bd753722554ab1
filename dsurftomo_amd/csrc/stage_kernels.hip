// Kernels around the fixed-point solve: period-level velocity tables, the per-source stages of
// source_stage.h, and the receiver gather.  See kernels.h for the data layout.
#include "kernels.h"
#include "receiver_core.h"
#include "exact_march.h"

namespace dsa {

namespace {

__device__ __forceinline__ SourceScratch scratch_of(const BatchPtrs& b, int s)
{
    SourceScratch w;
    const size_t rr = (size_t)kRefMax * kRefMax;
    w.slow_r = b.slow_r + (size_t)s * kRefRecs;
    w.F_r = b.F_r + (size_t)s * kRefRecs;
    w.S_r = b.S_r + s * rr;
    w.risti_r = b.risti_r + (size_t)s * kRefMax;
    w.vcorner = b.vcorner + (size_t)s * 4;
    w.rst = b.rst + (size_t)s * kRWin * kRWin;
    w.cst = b.cst + (size_t)s * kCWinMax * kCWinMax;
    w.cinit = b.cinit + (size_t)s * kCWinMax * kCWinMax;
    w.heap = b.heap + (size_t)s * kHeapCap;
    w.flags = b.flags + (size_t)s * 4;
    return w;
}

// queue node (iz, ix) (1-based) as a seed of the fixed-point solve; single writer per source
__device__ __forceinline__ void seed_node(Rec* F, int nbz, int* seed, int* nseed, int cap, int iz, int ix)
{
    const int id = rec_index(nbz, iz - 1, ix - 1);
    unsigned* bits = reinterpret_cast<unsigned*>(&F[id].tau);
    if (*bits & kQueuedBit) return;
    *bits |= kQueuedBit;
    if (*nseed < cap) seed[*nseed] = (int)id;
    *nseed += 1;      // a count above cap makes the solve kernel rescan the field
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// K1: dice one period's vertex map onto the propagation grid; z fastest -> lanes walk iz
__global__ void k_gridder(GridDesc g, const float* __restrict__ velv, const float* __restrict__ basis,
                          float* __restrict__ veln, float* __restrict__ slow)
{
    const int nrec = g.nbx * g.nbz * kTileRecs;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;      // tiled record index: lanes walk a tile
    if (id >= nrec) return;
    int iz0, ix0;
    rec_coords(g.nbz, id, &iz0, &ix0);
    if (iz0 >= g.nnz || ix0 >= g.nnx) { slow[id] = 1.0f; return; }        // padding of the last tiles
    const float v = coarse_velocity(g, velv, basis, iz0 + 1, ix0 + 1);
    veln[(size_t)ix0 * g.nnz + iz0] = v;
    slow[id] = 1.0f / v;
}

void launch_gridder(const GridDesc& g, const float* d_velv, const float* d_basis, float* d_veln, float* d_slow,
                    hipStream_t stream)
{
    const int nrec = g.nbx * g.nbz * kTileRecs;
    hipLaunchKernelGGL(k_gridder, dim3((nrec + 255) / 256), dim3(256), 0, stream, g, d_velv, d_basis, d_veln, d_slow);
}

// ---------------------------------------------------------------------------------------------
// K2a: refined velocities of every source box; also resets the refined field and block state
__global__ void k_refine(GridDesc g, BatchPtrs b, const float* __restrict__ velv_all, size_t velv_stride,
                         const float* __restrict__ rbasis)
{
    const int s = blockIdx.y;
    const SourceDesc sd = b.src[s];
    const int id = blockIdx.x * blockDim.x + threadIdx.x;       // tiled record index inside the box storage
    if (id < 4 && blockIdx.x == 0) b.flags[(size_t)s * 4 + id] = 0;
    if (id == 0 && blockIdx.x == 0) { b.nseed_r[s] = 0; b.nseed_c[s] = 0; }
    if (id >= kRefRecs) return;
    const size_t at = (size_t)s * kRefRecs + id;
    b.F_r[at] = Rec{ kInf, kInf };
    int kz0, lx0;
    rec_coords(sd.nbz_r, id, &kz0, &lx0);
    if (id >= sd.nbx_r * sd.nbz_r * kTileRecs || kz0 >= sd.rnz || lx0 >= sd.rnx) { b.slow_r[at] = 1.0f; return; }
    const int lx = lx0 + 1, kz = kz0 + 1;
    const float* velv = velv_all + (size_t)sd.period * velv_stride;
    const float v = refined_velocity(g, sd, velv, rbasis, kz, lx);
    b.slow_r[at] = 1.0f / v;
    if ((lx == sd.isx_r || lx == sd.isx_r + 1) && (kz == sd.isz_r || kz == sd.isz_r + 1))
        b.vcorner[(size_t)s * 4 + (lx - sd.isx_r) * 2 + (kz - sd.isz_r)] = v;
}

void launch_refine(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_velv_all, size_t velv_stride,
                   const float* d_rbasis, hipStream_t stream)
{
    if (nsrc <= 0) return;
    const int per = (kRefRecs + 255) / 256;
    hipLaunchKernelGGL(k_refine, dim3(per, nsrc), dim3(256), 0, stream, g, b, d_velv_all, velv_stride, d_rbasis);
}

// ---------------------------------------------------------------------------------------------
// K2b: start-up march, one lane per source (a few accept steps each; the control flow is the same
// for every source, so the lanes of a wave stay together)
__global__ void k_refined_startup(GridDesc g, BatchPtrs b, int nsrc)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const SourceDesc sd = b.src[s];
    SourceScratch w = scratch_of(b, s);
    const int ended = refined_startup(g, sd, w);
    refined_encode(sd, w, ended);
    if (ended) return;
    // the unpinned neighbours of every pinned node start the fixed-point solve
    int* seed = b.seed_r + (size_t)s * kSeedR;
    int nseed = 0;
    for (int lx = 0; lx < kRWin; ++lx)
        for (int lz = 0; lz < kRWin; ++lz) {
            if (w.rst[lx * kRWin + lz] != 0) continue;
            const int ix = sd.rwx0 + lx + 1, iz = sd.rwz0 + lz + 1;
            const int nx[4] = { ix - 1, ix + 1, ix, ix }, nz[4] = { iz, iz, iz - 1, iz + 1 };
            for (int q = 0; q < 4; ++q) {
                if (nx[q] < 1 || nx[q] > sd.rnx || nz[q] < 1 || nz[q] > sd.rnz) continue;
                if (t_pinned(w.F_r[rec_index(sd.nbz_r, nz[q] - 1, nx[q] - 1)].T)) continue;
                seed_node(w.F_r, sd.nbz_r, seed, &nseed, kSeedR, nz[q], nx[q]);
            }
        }
    b.nseed_r[s] = nseed;
}

void launch_refined_startup(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream)
{
    if (nsrc <= 0) return;
    hipLaunchKernelGGL(k_refined_startup, dim3((nsrc + 63) / 64), dim3(64), 0, stream, g, b, nsrc);
}

// ---------------------------------------------------------------------------------------------
// K2c: hand-off, one workgroup per source
// the hand-off of ONE unit by the threads of a workgroup (k_handoff: 256 of them; k_handoff_replay: 64)
__device__ __forceinline__ void handoff_unit(const GridDesc& g, const BatchPtrs& b, int s, unsigned long long& s_best, int& s_first)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const SourceDesc sd = b.src[s];
    SourceScratch w = scratch_of(b, s);
    const int ended = w.flags[0];
    const int n = sd.rnx * sd.rnz;
    const size_t rr = (size_t)kRefMax * kRefMax;
    if (tid == 0) { s_best = ~0ull; s_first = 0x7fffffff; }
    __syncthreads();
    // first open-edge node in acceptance order; exact ties resolved by scan order (ix outer, iz inner: the smallest id).  Round 5: the
    // lanes walk the box's PERIMETER (2 rnx + 2 rnz positions, the corners twice -- a minimum does not mind) instead of testing all rnx x rnz
    // nodes for an edge: two passes of 3 trips instead of 65; with the stencil records of handoff_node fetched together: stages 12.1 -> 8.8 ms for 16 000 sources
    if (!ended) {
        const int nper = 2 * sd.rnz + 2 * sd.rnx;
        auto edge_node = [&](int q, int* iz, int* ix) {
            if (q < sd.rnz) { *ix = 1; *iz = q + 1; }
            else if (q < 2 * sd.rnz) { *ix = sd.rnx; *iz = q - sd.rnz + 1; }
            else if (q < 2 * sd.rnz + sd.rnx) { *iz = 1; *ix = q - 2 * sd.rnz + 1; }
            else { *iz = sd.rnz; *ix = q - 2 * sd.rnz - sd.rnx + 1; }
        };
        for (int q = tid; q < nper; q += nt) {
            int iz, ix;
            edge_node(q, &iz, &ix);
            if (!is_open_edge(sd, iz, ix)) continue;
            const Rec r = w.F_r[rec_index(sd.nbz_r, iz - 1, ix - 1)];
            if (!(t_value(r.T) < kInf)) continue;
            atomicMin(&s_best, (unsigned long long)accept_rank(r.T, r.tau));
        }
        __syncthreads();
        const unsigned long long best = s_best;
        if (best != ~0ull)
            for (int q = tid; q < nper; q += nt) {
                int iz, ix;
                edge_node(q, &iz, &ix);
                if (!is_open_edge(sd, iz, ix)) continue;
                const Rec r = w.F_r[rec_index(sd.nbz_r, iz - 1, ix - 1)];
                if (!(t_value(r.T) < kInf)) continue;
                if ((unsigned long long)accept_rank(r.T, r.tau) == best) atomicMin(&s_first, (ix - 1) * sd.rnz + (iz - 1));
            }
    }
    __syncthreads();
    const uint64_t rstar = ended ? ~0ull : (uint64_t)s_best;
    int ez = 0, ex = 0;
    if (!ended && s_first != 0x7fffffff) { ex = s_first / sd.rnz + 1; ez = s_first % sd.rnz + 1; }
    if (tid == 0) { w.flags[2] = ez; w.flags[3] = ex; }
    float* Tfin = b.Tfin_r + s * rr;
    {   // (ix, iz) of id = tid, tid + 256, ... without a division per node
        const int dq = nt / sd.rnz, dr = nt % sd.rnz;
        int ix0 = tid / sd.rnz, iz0 = tid % sd.rnz;
        for (int id = tid; id < n; id += nt) {
            float t;
            const int st = handoff_node(g, sd, w, ended, rstar, ez, ex, iz0 + 1, ix0 + 1, &t);
            w.S_r[id] = (int8_t)st;
            Tfin[id] = t;
            iz0 += dr; ix0 += dq;
            if (iz0 >= sd.rnz) { iz0 -= sd.rnz; ix0 += 1; }
        }
    }
    // coarse window: everything far, then every 8th refined node, then band promotion
    const int wn = sd.cwnx * sd.cwnz;
    Rec* W = b.W_c + (size_t)s * kCWinMax * kCWinMax;           // records of the coarse march window, (cwnz, cwnx) column-major
    for (int q = tid; q < wn; q += nt) { w.cst[q] = -1; W[q] = Rec{ kInf, kInf }; }
    __syncthreads();
    const int bxn = (sd.rnx - 1) / kSgdl + 1, bzn = (sd.rnz - 1) / kSgdl + 1;
    for (int q = tid; q < bxn * bzn; q += nt) {
        const int l = (q / bzn) * kSgdl + 1, k = (q % bzn) * kSgdl + 1;
        const int cz = sd.vnt + (k - 1) / kSgdl, cx = sd.vnl + (l - 1) / kSgdl;
        const int id = (l - 1) * sd.rnz + (k - 1);
        const int st = w.S_r[id];
        w.cst[(cx - 1 - sd.cwx0) * sd.cwnz + (cz - 1 - sd.cwz0)] = (int16_t)st;
        if (st >= 0) W[(cx - 1 - sd.cwx0) * sd.cwnz + (cz - 1 - sd.cwz0)].T = Tfin[id];
    }
    __syncthreads();
    // alive nodes that touch a far node go back into the band. A promoted node reads as "not far"
    // before and after, so concurrent promotion is order independent.
    for (int q = tid; q < bxn * bzn; q += nt) {
        const int cx = sd.vnl + q / bzn, cz = sd.vnt + q % bzn;
        int16_t* me = &w.cst[(cx - 1 - sd.cwx0) * sd.cwnz + (cz - 1 - sd.cwz0)];
        if (*me != 0) continue;
        const int nx[4] = { cx - 1, cx + 1, cx, cx }, nz[4] = { cz, cz, cz - 1, cz + 1 };
        bool band = false;
        for (int d = 0; d < 4; ++d) {
            if (nx[d] < 1 || nx[d] > g.nnx || nz[d] < 1 || nz[d] > g.nnz) continue;
            const bool inwin = nz[d] > sd.cwz0 && nz[d] <= sd.cwz0 + sd.cwnz && nx[d] > sd.cwx0 && nx[d] <= sd.cwx0 + sd.cwnx;
            if (!inwin || w.cst[(nx[d] - 1 - sd.cwx0) * sd.cwnz + (nz[d] - 1 - sd.cwz0)] == -1) band = true;
        }
        if (band) *me = 1;
    }
}

__global__ __launch_bounds__(256) void k_handoff(GridDesc g, BatchPtrs b)
{
    __shared__ unsigned long long s_best;
    __shared__ int s_first;
    handoff_unit(g, b, blockIdx.x, s_best, s_first);
}

// (round 6) The hand-off's tie probe, a kernel of its own behind k_handoff (one workgroup per source; it only reads what k_handoff left).  A node of the box
// whose acceptance rank TIES with the terminating node's was accepted before it or not as the reference's tree had it -- ref_alive guesses by scan order.
// Such nodes are collected, and the hand-off's outputs that reach the coarse grid (every kSgdl-th node of the box: status and time) are evaluated once more
// with the tied node's answer inverted: a changed time is the tie's influence (counted into the unit's tie record like the census' ties: any / sum / above
// the threshold); a changed STATUS on the lattice, or more ties than the probe holds, counts in word [6] of the record's refined half, which flags the
// unit by itself (Engine::tie_verdict) -- except in a laterally homogeneous box, where such ties are the grid's symmetry and only count.
__global__ __launch_bounds__(256) void k_handoff_probe(GridDesc g, BatchPtrs b, int32_t* tie, float tie_threshold, int32_t* replay, int replay_cap)
{
    constexpr int kTiedMax = 8;
    __shared__ int s_tied[kTiedMax], s_ntied;
    const int s = blockIdx.x, tid = threadIdx.x;
    const SourceDesc sd = b.src[s];
    SourceScratch w = scratch_of(b, s);
    const int ended = w.flags[0], ez = w.flags[2], ex = w.flags[3];
    if (tid == 0) s_ntied = 0;
    __syncthreads();
    if (ended || ex <= 0) return;                    // (uniform: the refined stage ended by itself, or no open edge was reached -- no terminating node, no tie with it)
    const int n = sd.rnx * sd.rnz, eid = (ex - 1) * sd.rnz + (ez - 1);
    const Rec er = w.F_r[rec_index(sd.nbz_r, ez - 1, ex - 1)];
    const uint64_t rstar = accept_rank(er.T, er.tau);
    {
        const int dq = 256 / sd.rnz, dr = 256 % sd.rnz;
        int ix0 = tid / sd.rnz, iz0 = tid % sd.rnz;
        for (int id = tid; id < n; id += 256) {
            const Rec r = w.F_r[rec_index(sd.nbz_r, iz0, ix0)];
            if (id != eid && !t_pinned(r.T) && t_value(r.T) < kInf && accept_rank(r.T, r.tau) == rstar) { const int k = atomicAdd(&s_ntied, 1); if (k < kTiedMax) s_tied[k] = id; }
            iz0 += dr; ix0 += dq;
            if (iz0 >= sd.rnz) { iz0 -= sd.rnz; ix0 += 1; }
        }
    }
    __syncthreads();
    if (s_ntied == 0) return;
    // (a laterally homogeneous box -- a 1-D starting model: every map of the Taipei example's first iteration -- ties like this BY SYMMETRY, one station of
    // that example in 22 of its 26 periods; there the scan-order rule is the reference's behaviour, checked bit for bit, and a changed status only counts)
    __shared__ unsigned s_smin, s_smax;          // (positive floats order like their bit patterns)
    if (tid == 0) { s_smin = 0x7f800000u; s_smax = 0u; }
    __syncthreads();
    {
        unsigned lo = 0x7f800000u, hi = 0u;
        const int dq = 256 / sd.rnz, dr = 256 % sd.rnz;
        int ix0 = tid / sd.rnz, iz0 = tid % sd.rnz;
        for (int id = tid; id < n; id += 256) {
            const unsigned v = __float_as_uint(w.slow_r[rec_index(sd.nbz_r, iz0, ix0)]);
            lo = v < lo ? v : lo; hi = v > hi ? v : hi;
            iz0 += dr; ix0 += dq;
            if (iz0 >= sd.rnz) { iz0 -= sd.rnz; ix0 += 1; }
        }
        atomicMin(&s_smin, lo); atomicMax(&s_smax, hi);
    }
    __syncthreads();
    // (homogeneous up to the rounding of the B-spline weights that dice a constant map: a few ulps)
    const bool homogeneous = __uint_as_float(s_smax) - __uint_as_float(s_smin) <= 1.0e-5f * __uint_as_float(s_smin);
    if (tid >= 9) return;
    const float* Tfin = b.Tfin_r + (size_t)s * kRefMax * kRefMax;
    int32_t* const tr = tie + (size_t)s * kTieWords;
    const int nt = s_ntied < kTiedMax ? s_ntied : kTiedMax;
    // thread z of the first nine looks at the tied node (z = 0) or at one of its eight stencil nodes
    const int dx = tid == 1 ? -1 : tid == 2 ? 1 : tid == 5 ? -2 : tid == 6 ? 2 : 0, dz = tid == 3 ? -1 : tid == 4 ? 1 : tid == 7 ? -2 : tid == 8 ? 2 : 0;
    if (replay && !homogeneous) {
        // (round 6, last) A tie that changes what the coarse grid receives is not guessed at and not flagged: the unit goes on a list, and
        // k_handoff_replay marches its refined box literally (129^2 nodes: the reference's own tree decides) and hands off from that.  A full list,
        // or a homogeneous box (the symmetric ties of a 1-D model: counted below), keeps the old way.
        bool change = s_ntied > kTiedMax;
        for (int k = 0; k < nt && !change; ++k) {
            const int yid = s_tied[k], zx = yid / sd.rnz + 1 + dx, zz = yid % sd.rnz + 1 + dz;
            if (!(zx >= 1 && zx <= sd.rnx && zz >= 1 && zz <= sd.rnz && (zx - 1) % kSgdl == 0 && (zz - 1) % kSgdl == 0)) continue;
            float t1;
            const int st1 = handoff_node<true>(g, sd, w, 0, rstar, ez, ex, zz, zx, &t1, yid);
            const int zid = (zx - 1) * sd.rnz + (zz - 1);
            change = st1 != w.S_r[zid] || (st1 >= 0 && t1 != Tfin[zid]);
        }
        if (!__any(change)) return;
        int slot = 0;
        if (tid == 0) slot = atomicAdd(replay, 1);
        slot = __shfl(slot, 0);
        if (slot < replay_cap) { if (tid == 0) replay[1 + slot] = s; return; }
    }
    if (s_ntied > kTiedMax && tid == 0) atomicAdd(tr + 6, 1);
    for (int k = 0; k < nt; ++k) {
        const int yid = s_tied[k], zx = yid / sd.rnz + 1 + dx, zz = yid % sd.rnz + 1 + dz;
        // (only what reaches the coarse grid counts: the nodes k_handoff injects)
        if (!(zx >= 1 && zx <= sd.rnx && zz >= 1 && zz <= sd.rnz && (zx - 1) % kSgdl == 0 && (zz - 1) % kSgdl == 0)) continue;
        float t1;
        const int st1 = handoff_node<true>(g, sd, w, 0, rstar, ez, ex, zz, zx, &t1, yid);
        const int zid = (zx - 1) * sd.rnz + (zz - 1);
        const int st0 = w.S_r[zid];
        const float t0 = Tfin[zid];
        // (a changed STATUS on the lattice: the coarse grid starts from another band -- on small grids, where the box is a good part of the field, 2e-4 to
        // 4.5e-4 s at a receiver (profiles/r06_tie_fuzz_symmetric_sources.log) --: the unit is flagged, word [6]; in a homogeneous box it only counts)
        if (st1 != st0) { if (homogeneous) atomicAdd((unsigned*)tr + 2, 1u); else atomicAdd(tr + 6, 1); }
        else if (st0 >= 0 && t1 != t0) {
            const float ti = fabsf(t1 - t0);
            atomicAdd((unsigned*)tr + 2, 1u); atomicAdd((unsigned*)tr + 3, (unsigned)(fminf(ti, 1.0f) * (1.0f / kTieSumUnit)));
            if (ti > tie_threshold) { atomicAdd((unsigned*)tr, 1u); atomicMax((unsigned*)tr + 1, __float_as_uint(ti)); }
        }
    }
}

// (round 6, last) The refined box of a listed unit by the reference's march itself (exact_kernel.hip: launch_refined_replay, k_xmarch<refined> on the unit's
// own records: ~16 000 accepts at 2.3 us); its state then replaces the fixed point's records in the form of a refined stage that ended by itself
// (alive nodes pinned, the narrow band's trial values as they stand: handoff_node's first branch), and the hand-off runs again.  What the coarse grid
// receives is then the reference's, whatever its tree decided about the nodes that rank equal with the terminating one.  A march that runs out of tree
// slots leaves the hand-off as it was and flags the unit (word [6]).
static_assert(sizeof(XRec) == sizeof(Rec) && sizeof(XEntry) == 8, "the march's records lie where the fixed point's do");

__global__ __launch_bounds__(64) void k_handoff_replay(GridDesc g, BatchPtrs b, const int32_t* replay, int replay_cap, const int32_t* xinfo, int32_t* tie)
{
    __shared__ unsigned long long s_best;
    __shared__ int s_first;
    const int j = blockIdx.x, lane = threadIdx.x;
    const int cnt = replay[0] < replay_cap ? replay[0] : replay_cap;
    if (j >= cnt) return;
    const int s = replay[1 + j];
    const SourceDesc sd = b.src[s];
    SourceScratch w = scratch_of(b, s);
    // (a march that ran out of tree slots: the hand-off stays as the fixed point left it and the unit is flagged -- marched whole)
    if (xinfo[4 * s + 2]) { if (lane == 0) atomicAdd(tie + (size_t)s * kTieWords + 6, 1); return; }
    for (int ix = 0; ix < sd.rnx; ++ix)
        for (int iz = lane; iz < sd.rnz; iz += 64) {
            const int id = rec_index(sd.nbz_r, iz, ix);
            const XRec r = reinterpret_cast<const XRec*>(w.F_r)[id];
            w.F_r[id] = r.st == 0 ? Rec{ -r.T, r.T } : r.st > 0 ? Rec{ r.T, r.T } : Rec{ kInf, kInf };
        }
    if (lane == 0) w.flags[0] = 1;
    __threadfence_block();
    __syncthreads();
    handoff_unit(g, b, s, s_best, s_first);
}

void launch_handoff(const GridDesc& g, const BatchPtrs& b, int nsrc, hipStream_t stream, int32_t* d_tie, float tie_threshold, int32_t* d_replay, int replay_cap, void* d_replay_scratch,
                    int32_t* d_xinfo)
{
    if (nsrc <= 0) return;
    hipLaunchKernelGGL(k_handoff, dim3(nsrc), dim3(256), 0, stream, g, b);
    const bool replay = d_tie && d_replay && replay_cap > 0 && d_replay_scratch && d_xinfo;
    if (d_tie) hipLaunchKernelGGL(k_handoff_probe, dim3(nsrc), dim3(256), 0, stream, g, b, d_tie, tie_threshold, replay ? d_replay : nullptr, replay_cap);
    if (replay) {
        launch_refined_replay(g, b, d_replay, replay_cap, d_replay_scratch, kReplayHeap, d_xinfo, stream);
        hipLaunchKernelGGL(k_handoff_replay, dim3(replay_cap), dim3(64), 0, stream, g, b, (const int32_t*)d_replay, replay_cap, (const int32_t*)d_xinfo, d_tie);
    }
}

// ---------------------------------------------------------------------------------------------
// K2d: band march on the coarse grid, then the seeds of the coarse solve.  One wavefront per source: the loops over the status
// window (up to 65 x 65 nodes: tree start, pinning, export into the compact field, seeds) run on all 64 lanes, the tree itself
// (a few dozen accept steps) on lane 0.  [Round 1 ran everything on one lane per source: four serial sweeps of the window in
// global memory, 5.8 ms for the 449 sources of the Taipei example and 10 ms for any number up to 16 000.]
__global__ __launch_bounds__(64) void k_coarse_march(GridDesc g, BatchPtrs b, int nsrc, const float* __restrict__ slow_all,
                                                     size_t field_stride, const float* __restrict__ risti_c, int32_t* tie)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nsrc) return;
    const SourceDesc sd = b.src[s];
    SourceScratch w = scratch_of(b, s);
    Rec* W = b.W_c + (size_t)s * kCWinMax * kCWinMax;
    const float* slow_c = slow_all + (size_t)sd.period * field_stride;      // tiled slowness of this period
    const int nw = sd.cwnx * sd.cwnz;
    const unsigned long long below = (1ull << lane) - 1ull;

    // tree start in the reference's order (ix outer, iz inner = ascending q, CalSurfG.f90:341-347): the lanes classify 64 nodes,
    // lane 0 adds the close ones in order
    MarchView m = band_march_view(g, sd, w, W, slow_c, risti_c);
    int ninit = 0;
    for (int base = 0; base < nw; base += 64) {
        const int q = base + lane;
        int st = -1;
        if (q < nw) {
            st = w.cst[q];
            w.cinit[q] = st > 0 ? 1 : 0;
            if (st == 0) W[q].tau = 0.0f;             // alive before the march
        }
        unsigned long long close = __ballot(st > 0);
        if (lane == 0)
            while (close) {
                const int qq = base + __ffsll((long long)close) - 1;
                close &= close - 1ull;
                ++ninit;
                mv_add(m, sd.cwz0 + qq % sd.cwnz + 1, sd.cwx0 + qq / sd.cwnz + 1);
            }
    }
    __threadfence_block();
    if (lane == 0) band_march_run(m, sd, w, ninit);
    __threadfence_block();
    // (round 6, source_stage.h: band_march_settle) on with the serial march until the tree is a heap: the wavefront checks the tree, lane 0 accepts
    {
        int left = 0;
        for (int extra = 0;; ++extra) {
            m.ntr = __shfl(m.ntr, 0); m.error = __shfl(m.error, 0);
            if (m.ntr <= 1 || m.error) break;
            const bool valid = __all(mv_heap_valid(m, 2 + lane, 64));
            if (valid) break;
            int go = 0;
            if (lane == 0) go = (extra < kBandExtra && mv_root_inside(m)) ? 1 : 0;
            go = __shfl(go, 0);
            if (!go) { left = 1; break; }
            if (lane == 0 && !mv_accept_root(m)) m.error = m.error ? m.error : 1;
            __threadfence_block();
        }
        if (lane == 0) {
            if (m.error && !w.flags[1]) w.flags[1] = 16 + m.error;
            if (left && tie) atomicAdd(tie + (size_t)s * kTieWords + kTieWords / 2 + 6, 1);
        }
    }
    __threadfence_block();
    // pinning: the window's alive nodes get their sign bit; the coarse solve carries them into its field slot and exception table (FimEnds)
    for (int q = lane; q < nw; q += 64) band_march_finish_node(w, W, q);
    // seeds of the fixed-point solve: every node of the grid that is not itself pinned and has a pinned neighbour; they lie in the
    // window or in the ring around it, so each candidate is looked at by one lane and listed once
    int* seed = b.seed_c + (size_t)s * kSeedC;
    int nseed = 0;
    const int enz = sd.cwnz + 2, ncand = (sd.cwnx + 2) * enz;
    auto pinned_at = [&](int lx, int lz) { return lx >= 0 && lx < sd.cwnx && lz >= 0 && lz < sd.cwnz && w.cst[lx * sd.cwnz + lz] == 0; };
    for (int base = 0; base < ncand; base += 64) {
        const int c = base + lane;
        bool is_seed = false;
        int id = 0;
        if (c < ncand) {
            const int lx = c / enz - 1, lz = c - (lx + 1) * enz - 1;
            const int ix = sd.cwx0 + lx + 1, iz = sd.cwz0 + lz + 1;            // 1-based grid indices
            if (ix >= 1 && ix <= g.nnx && iz >= 1 && iz <= g.nnz && !pinned_at(lx, lz) &&
                (pinned_at(lx - 1, lz) || pinned_at(lx + 1, lz) || pinned_at(lx, lz - 1) || pinned_at(lx, lz + 1))) {
                is_seed = true;
                id = rec_index(g.nbz, iz - 1, ix - 1);
            }
        }
        const unsigned long long sm = __ballot(is_seed);
        const int pos = nseed + __popcll(sm & below);
        if (is_seed && pos < kSeedC) seed[pos] = id;
        nseed += __popcll(sm);
    }
    if (lane == 0) b.nseed_c[s] = nseed;
}

void launch_coarse_march(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                         size_t field_stride, const float* d_risti_c, hipStream_t stream, int32_t* d_tie)
{
    if (nsrc <= 0) return;
    hipLaunchKernelGGL(k_coarse_march, dim3(nsrc), dim3(64), 0, stream, g, b, nsrc, d_slow_all, field_stride, d_risti_c, d_tie);
}

// ---------------------------------------------------------------------------------------------
__global__ void k_make_problems(GridDesc g, BatchPtrs b, int nsrc, const float* slow_all, size_t field_stride,
                                const float* risti_c, float window_r, float window_c, FimProblem* prob_r,
                                FimProblem* prob_c, int32_t* info, unsigned long long* clocks, const int* __restrict__ launch_rank,
                                int32_t* tie, float tie_threshold, FimEnds* ends_c, const RayDesc* __restrict__ rays, const float* __restrict__ veln_all,
                                size_t veln_stride, float dpl, float* out, int32_t* err, const int* __restrict__ member_flag, float window_b, int max_rounds_b, float window_t, FimEnds* ends_r)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsrc) return;
    const SourceDesc sd = b.src[s];
    FimProblem r;
    r.Tc = nullptr; r.exc = nullptr; r.exc_log2cap = 0;
    r.F = b.F_r + (size_t)s * kRefRecs; r.slow = b.slow_r + (size_t)s * kRefRecs; r.risti = b.risti_r + (size_t)s * kRefMax;
    r.seed = b.seed_r + (size_t)s * kSeedR; r.seed_count = b.nseed_r + s; r.seed_cap = kSeedR; r.lists = b.lists + (size_t)s * b.lists_stride;
    r.nnx = sd.rnx; r.nnz = sd.rnz; r.nbx = sd.nbx_r; r.nbz = sd.nbz_r;
    r.ri = g.earth; r.dnx = sd.rdnx; r.dnz = sd.rdnz; r.window = window_r;
    r.max_rounds = 64 * (sd.rnx + sd.rnz) + 4096;
    r.clocks = nullptr;
    r.info = info + (size_t)s * 16;
    r.tie = tie ? tie + (size_t)s * kTieWords : nullptr; r.tie_threshold = tie_threshold;
    r.ended = b.flags + (size_t)s * 4;
    // field slot: the unit's own, or (a pool smaller than the launch) the one its workgroup number selects
    const int rank = launch_rank ? launch_rank[s] : s;
    if (ends_r) {
        // (the refined boxes of bundled units are solved in bundles too: the refined problems by launch rank, like the coarse ones -- solo units
        // first --, each with the records its bundle takes the pinned nodes from)
        prob_r[rank] = r;
        FimEnds er{};
        er.Fpin = r.F;
        ends_r[rank] = er;
    } else prob_r[s] = r;
    const bool recycled = b.pool < nsrc;
    const int slot = recycled ? rank % b.pool : s;
    // a member of a bundle (bundle_kernel.hip) is solved inside its bundle's field: it has no slot of its own unless every unit has one
    // (the bundle then leaves the member's field there for the rays)
    const bool member = member_flag && member_flag[s];
    FimProblem c;
    c.F = nullptr;
    c.Tc = b.T_c + (size_t)slot * g.nbx * g.nbz * kTileRecs; c.exc = b.exc_c + ((size_t)slot << b.exc_log2cap); c.exc_log2cap = b.exc_log2cap;
    if (member && recycled) { c.Tc = nullptr; c.exc = nullptr; }
    c.slow = slow_all + (size_t)sd.period * field_stride; c.risti = risti_c;
    c.seed = b.seed_c + (size_t)s * kSeedC; c.seed_count = b.nseed_c + s; c.seed_cap = kSeedC; c.lists = b.lists_c + (size_t)slot * b.lists_c_stride;
    c.nnx = g.nnx; c.nnz = g.nnz; c.nbx = g.nbx; c.nbz = g.nbz;
    c.ri = g.earth; c.dnx = g.dnx; c.dnz = g.dnz; c.window = member ? (member_flag[s] == 2 && window_t > 0.0f ? window_t : window_b) : window_c;
    // (a bundle whose members' fronts have nothing in common re-evaluates without end: it gives up sixteen times sooner and its chunk goes unit by unit)
    c.max_rounds = member ? (max_rounds_b > 0 ? max_rounds_b : 4 * (g.nnx + g.nnz) + 2048) : 64 * (g.nnx + g.nnz) + 4096;
    c.clocks = clocks ? clocks + (size_t)s * kClockSlots : nullptr;
    c.info = info + (size_t)s * 16 + 8;
    c.tie = tie ? tie + (size_t)s * kTieWords + kTieWords / 2 : nullptr; c.tie_threshold = tie_threshold;
    c.ended = nullptr;
    prob_c[rank] = c;      // workgroup launch_rank[s] solves unit s: the longest solves start first
    if (ends_c) {
        FimEnds e;
        e.Fpin = nullptr;
        e.W = b.W_c + (size_t)s * kCWinMax * kCWinMax; e.cwz0 = sd.cwz0; e.cwx0 = sd.cwx0; e.cwnz = sd.cwnz; e.cwnx = sd.cwnx;
        e.slot_busy = recycled && !member ? b.pool_gen : nullptr; e.nslots = b.pool;
        e.Tc_pool = b.T_c; e.exc_pool = b.exc_c; e.lists_pool = b.lists_c; e.lists_stride = (unsigned)b.lists_c_stride;
        e.rays = rays ? rays + sd.first_ray : nullptr; e.nrays = sd.nrec; e.ray0 = sd.first_ray;
        e.veln = veln_all + (size_t)sd.period * veln_stride; e.scx = sd.scx; e.scz = sd.scz; e.dpl = dpl; e.out = out; e.err = err; e.g = g;
        ends_c[rank] = e;
    }
    for (int q = 0; q < 16; ++q) info[(size_t)s * 16 + q] = 0;
    if (tie) for (int q = 0; q < kTieWords; ++q) tie[(size_t)s * kTieWords + q] = 0;
    if (clocks) for (int q = 0; q < kClockSlots; ++q) clocks[(size_t)s * kClockSlots + q] = 0ull;      // probe builds accumulate into them
}

void launch_make_problems(const GridDesc& g, const BatchPtrs& b, int nsrc, const float* d_slow_all,
                          size_t field_stride, const float* d_risti_c, float window_r, float window_c,
                          FimProblem* d_prob_r, FimProblem* d_prob_c, int32_t* d_info, unsigned long long* d_clocks,
                          const int* d_launch_rank, int32_t* d_tie, float tie_threshold, FimEnds* d_ends_c, const RayDesc* d_rays,
                          const float* d_veln_all, size_t veln_stride, float dpl, float* d_out, int32_t* d_err, const int* d_member_flag, float window_b, int max_rounds_b, hipStream_t stream, float window_t, FimEnds* d_ends_r)
{
    if (nsrc <= 0) return;
    hipLaunchKernelGGL(k_make_problems, dim3((nsrc + 63) / 64), dim3(64), 0, stream, g, b, nsrc, d_slow_all,
                       field_stride, d_risti_c, window_r, window_c, d_prob_r, d_prob_c, d_info, d_clocks, d_launch_rank, d_tie, tie_threshold,
                       d_ends_c, d_rays, d_veln_all, veln_stride, dpl, d_out, d_err, d_member_flag, window_b, max_rounds_b, window_t, d_ends_r);
}

// ---------------------------------------------------------------------------------------------
// K4: receiver travel times, one thread per ray; reference srtimes (CalSurfG.f90:1681-1754)
__global__ void k_srtimes(GridDesc g, BatchPtrs b, int unit_base, const RayDesc* __restrict__ rays, int nrays,
                          const float* __restrict__ veln_all, size_t field_stride, float dpl,
                          float* __restrict__ out, int32_t* __restrict__ err)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrays) return;
    const RayDesc rd = rays[r];
    if (!(rd.flags & kRayTime)) return;
    const int slot = rd.src - unit_base;
    const SourceDesc sd = b.src[slot];
    const float* Tc = b.T_c + (size_t)slot * g.nbx * g.nbz * kTileRecs;
    const float* veln = veln_all + (size_t)sd.period * field_stride;
    float trr;
    if (!receiver_time(g, sd.scx, sd.scz, rd, Tc, veln, dpl, &trr)) { atomicExch(err, r + 1); out[rd.data] = 0.0f; return; }
    out[rd.data] = trr;
}

void launch_srtimes(const GridDesc& g, const BatchPtrs& b, int unit_base, const RayDesc* d_rays, int nrays,
                    const float* d_veln_all, size_t field_stride, float dpl, float* d_out, int32_t* d_err,
                    hipStream_t stream)
{
    if (nrays <= 0) return;
    hipLaunchKernelGGL(k_srtimes, dim3((nrays + 255) / 256), dim3(256), 0, stream, g, b, unit_base, d_rays, nrays,
                       d_veln_all, field_stride, dpl, d_out, d_err);
}

}  // namespace dsa
