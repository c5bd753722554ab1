// Per-source stages around the two fixed-point solves of one (period, source) unit:
//
//   refine velocities in the source box -> start-up march (serial, a few accepts)
//     -> fixed point on the refined box -> hand-off (first open-edge arrival, alive/close/far,
//        trial values, every 8th node onto the coarse grid, band promotion)
//     -> band march on the coarse grid (serial, a few dozen accepts) -> fixed point on the coarse grid
//
// It mirrors what the reference does inline per source (CalSurfG.f90:1192-1356) including the
// behaviours that are load-bearing for parity: source-cell start values in radians (:369-375),
// the literal open-edge test that compares coarse box bounds with refined extents (:396-407 with
// :1241-1242), injection of status and value for status >= 0 (:1293-1303) and promotion of alive
// nodes that touch a far node (:1332-1349).
//
// Functions are __host__ __device__ so tests can run them on a CPU; kernels only wrap them.
#pragma once

#include "eikonal_core.h"

namespace dsa {

constexpr int kSgdl = 8;        // source-grid dicing level (reference sgdl)
constexpr int kSgs = 8;         // source-grid half extent in coarse nodes (reference sgs)
constexpr int kRefMax = 129;    // (2*sgs)*sgdl + 1
constexpr int kRefTiles = (kRefMax + 7) / 8;                       // 17
constexpr int kRefRecs = kRefTiles * kRefTiles * 64;               // tiled storage of the largest box
constexpr int kRWin = 32;       // status window (nodes) of the refined start-up march
constexpr int kCMargin = 24;    // margin (coarse nodes) of the coarse band-march window around the box
constexpr int kCWinMax = 2 * kSgs + 1 + 2 * kCMargin;  // 65
constexpr int kHeapCap = 1024;

// coarse propagation grid of one call (shared by all periods)
struct GridDesc {
    int nx, ny, nvx, nvz, gdx, gdz;
    int nnx, nnz;
    float gox, goz, dvx, dvz, dnx, dnz, earth;
    int nbx, nbz;          // 8x8-node blocks per side
};

// one (period slot, source); everything here is exact fp32/int arithmetic done once on the host
struct SourceDesc {
    float scx, scz;
    int period;                   // index of the velocity map
    int vnl, vnr, vnt, vnb;       // refined box in coarse node indices (1-based)
    int rnx, rnz;                 // refined node counts
    float rgox, rgoz, rdnx, rdnz; // refined origin / spacing
    int isx_r, isz_r;             // source cell in the refined grid
    float dsx_r, dsz_r;           // source offset inside that cell
    int open_xlo, open_xhi, open_zlo, open_zhi;  // literal open-edge flags of the refined stage
    int rwz0, rwx0;               // refined march window origin (0-based offset of element (1,1))
    int cwz0, cwx0, cwnz, cwnx;   // coarse march window
    int nbx_r, nbz_r;             // 8x8 tiles of the refined grid
    int first_ray, nrec;          // receivers of this source: rays [first_ray, first_ray+nrec)
    int sen_slot;                 // period slot of the depth kernels used by this unit's Frechet rows
};

// per-source scratch in device memory (all sources of a batch laid out back to back)
struct SourceScratch {
    float* slow_r;      // kRefRecs tiled slowness of the refined box
    Rec* F_r;           // kRefRecs tiled (T, tau) records of the refined solve (eikonal_core.h)
    int8_t* S_r;        // final refined status: -1 far, 0 alive, 1 close (kept for the ray tracer)
    float* risti_r;     // kRefMax
    float* vcorner;     // 4 refined velocities of the source cell, [i][j] i = x offset
    int16_t* rst;       // kRWin*kRWin status window of the start-up march
    int16_t* cst;       // kCWinMax*kCWinMax status window of the band march
    int8_t* cinit;      // same extent: node was in the tree when the band march started
    int32_t* heap;      // kHeapCap
    int32_t* flags;     // [0] refined stage ended inside the start-up march, [1] error code
};

// ---------------------------------------------------------------------------------------------
// refined velocity at refined node (k, l) = (z, x) 1-based of the box; reference bsplrefine
// (CalSurfG.f90:1596-1623). velv: (ny, nx) vertex values as fp32, velv(i,j) = velv[i*nx + j].
// ub/vb: 65x4 basis tables (u = (l-1)/64).
DSA_HD float refined_velocity(const GridDesc& g, const SourceDesc& s, const float* velv,
                              const float* basis, int kz, int lx)
{
    const int nr = g.gdx * kSgdl;                 // refined nodes per vertex cell (gdx == gdz)
    const int gz = (s.vnt - 1) * kSgdl + kz;      // global refined index, 1-based
    const int gx = (s.vnl - 1) * kSgdl + lx;
    int ci = (gz - 1) / nr + 1, k = (gz - 1) % nr + 1;    // vertex cell i (1..nvz-1), in-cell index
    int cj = (gx - 1) / nr + 1, l = (gx - 1) % nr + 1;
    if (ci > g.nvz - 1) { ci = g.nvz - 1; k = nr + 1; }   // last node belongs to the last cell
    if (cj > g.nvx - 1) { cj = g.nvx - 1; l = nr + 1; }
    const float* ub = basis + 4 * (l - 1);
    const float* vb = basis + 4 * (k - 1);
    float sum[4];
    for (int i1 = 1; i1 <= 4; ++i1) {
        float acc = 0.0f;
        for (int j1 = 1; j1 <= 4; ++j1)
            acc = acc + ub[j1 - 1] * velv[(ci - 2 + i1) * g.nx + (cj - 2 + j1)];
        sum[i1 - 1] = vb[i1 - 1] * acc;
    }
    return sum[0] + sum[1] + sum[2] + sum[3];
}

// coarse velocity at node (iz, ix); reference gridder (CalSurfG.f90:1526-1548)
DSA_HD float coarse_velocity(const GridDesc& g, const float* velv, const float* basis, int iz, int ix)
{
    int ci = (iz - 1) / g.gdz + 1, l = (iz - 1) % g.gdz + 1;
    int cj = (ix - 1) / g.gdx + 1, m = (ix - 1) % g.gdx + 1;
    if (ci > g.nvz - 1) { ci = g.nvz - 1; l = g.gdz + 1; }
    if (cj > g.nvx - 1) { cj = g.nvx - 1; m = g.gdx + 1; }
    const float* ui = basis + 4 * (m - 1);
    const float* vi = basis + 4 * (l - 1);
    float sumi = 0.0f;
    for (int i1 = 1; i1 <= 4; ++i1) {
        float sumj = 0.0f;
        for (int j1 = 1; j1 <= 4; ++j1)
            sumj = sumj + ui[j1 - 1] * velv[(ci - 2 + i1) * g.nx + (cj - 2 + j1)];
        sumi = sumi + vi[i1 - 1] * sumj;
    }
    return sumi;
}

// ---------------------------------------------------------------------------------------------
// Start-up march on the refined grid: travel(urg=1) from its beginning until the four corners of
// the source cell are alive (then the regular fixed-point solve takes over), or until the
// reference's own exit fires.  Returns 1 if the refined stage is finished (exit fired / tree
// empty), 0 for hand-over.  On return T_r holds plain values where status >= 0.
DSA_HD int refined_startup(const GridDesc& g, const SourceDesc& s, SourceScratch& w)
{
    MarchView m;
    m.F = w.F_r; m.window = 0; m.slow = w.slow_r; m.nbz = s.nbz_r; m.risti = w.risti_r;
    m.status = w.rst; m.wz0 = s.rwz0; m.wx0 = s.rwx0; m.wnz = kRWin; m.wnx = kRWin;
    m.nnz = s.rnz; m.nnx = s.rnx; m.ri = g.earth; m.dnx = s.rdnx; m.dnz = s.rdnz;
    m.heap = w.heap; m.cap = kHeapCap; m.ntr = 0; m.error = 0; m.clock = 0.0f;
    for (int q = 0; q < kRWin * kRWin; ++q) w.rst[q] = -1;
    const int isx = s.isx_r, isz = s.isz_r;
    float vss[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) vss[i][j] = w.vcorner[i * 2 + j];
    const float vsrc = bilinear4(vss, s.rdnx, s.rdnz, s.dsx_r, s.dsz_r);
    for (int i = 1; i <= 2; ++i)
        for (int j = 1; j <= 2; ++j) {
            // distances in radians, as the reference has them
            const float ds = sqrtf(sq(s.dsx_r - (float)(i - 1) * s.rdnx) + sq(s.dsz_r - (float)(j - 1) * s.rdnz));
            mv_T(m, isz - 1 + j, isx - 1 + i) = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
            mv_add(m, isz - 1 + j, isx - 1 + i);
        }
    int ended = 1;
    while (m.ntr > 0 && m.error == 0) {
        if (mv_get(m, isz, isx) == 0 && mv_get(m, isz + 1, isx) == 0 && mv_get(m, isz, isx + 1) == 0 &&
            mv_get(m, isz + 1, isx + 1) == 0) { ended = 0; break; }
        const int ix = hp_ix(m.heap[1]), iz = hp_iz(m.heap[1]);
        bool out = false;
        if (ix == 1 && s.open_xlo) out = true;
        if (ix == s.rnx && s.open_xhi) out = true;
        if (iz == 1 && s.open_zlo) out = true;
        if (iz == s.rnz && s.open_zhi) out = true;
        if (out) { mv_set(m, iz, ix, 0); break; }
        if (!mv_accept_root(m)) break;
    }
    w.flags[0] = ended;
    if (m.error) w.flags[1] = m.error;
    return ended;
}

// After the start-up march: encode the march state into the refined field for the fixed-point
// solve (alive -> pinned sign bit; everything else +inf unless the stage already ended, in which
// case close nodes keep their trial value).  Serial over the small window.
DSA_HD void refined_encode(const SourceDesc& s, SourceScratch& w, int ended)
{
    for (int lx = 0; lx < kRWin; ++lx)
        for (int lz = 0; lz < kRWin; ++lz) {
            const int ix = s.rwx0 + lx + 1, iz = s.rwz0 + lz + 1;
            if (ix < 1 || ix > s.rnx || iz < 1 || iz > s.rnz) continue;
            const int st = w.rst[lx * kRWin + lz];
            Rec& r = w.F_r[rec_index(s.nbz_r, iz - 1, ix - 1)];
            if (st == 0) { r.T = -r.T; }                           // tau = accept number of the march; -0.0f keeps the sign bit
            else if (st > 0 && ended) r.tau = r.T;                 // keep the trial value
            else { r.T = kInf; r.tau = kInf; }
        }
}

// ---------------------------------------------------------------------------------------------
// Hand-off, per refined node, after the fixed point on the box (or after an early end).
// rstar: acceptance rank of the first open-edge node; (ez, ex): that node (0 if none).
DSA_HD bool is_open_edge(const SourceDesc& s, int iz, int ix)
{
    return (ix == 1 && s.open_xlo) || (ix == s.rnx && s.open_xhi) || (iz == 1 && s.open_zlo) ||
           (iz == s.rnz && s.open_zhi);
}

// Acceptance order of the refined solve: by tau; among nodes accepted at the same clock value the
// causal one (T == tau) goes first, then the non-causal ones by value.  Packed so that unsigned
// comparison orders it (all values are >= 0).
DSA_HD uint64_t accept_rank(float t_raw, float tau_raw)
{
    const float t = t_value(t_raw), k = tau_value(tau_raw);
    union { float f; uint32_t u; } a, b;
    a.f = k; b.f = t;
    const uint32_t sec = (t < k) ? b.u + 1u : 0u;
    return ((uint64_t)a.u << 32) | sec;
}

// rstar: rank of the open-edge node that ended the refined stage (~0 if none); eid: its scan index
// (ix-1)*rnz + (iz-1), or -1.  A node whose rank TIES with that node's was accepted before it iff it comes
// earlier in scan order: the reference's march inserts the x-, x+, z-, z+ neighbours in that order, so of two
// mirror-image nodes with bit-equal times the one with the lower index sits higher in its tree and is popped
// first (seen at the symmetric partner of the terminating node in homogeneous media).
DSA_HD bool ref_alive(const SourceDesc& s, const SourceScratch& w, uint64_t rstar, int eid, int iz, int ix)
{
    const Rec r = w.F_r[rec_index(s.nbz_r, iz - 1, ix - 1)];
    if (t_pinned(r.T)) return true;
    const uint64_t rk = accept_rank(r.T, r.tau);
    return rk < rstar || (rk == rstar && (ix - 1) * s.rnz + (iz - 1) < eid);
}

// ref_alive on a record already in hand
// (round 6: `flip` -- the scan index of a node whose rank ties with the terminating node's, or -1 -- inverts that node's answer: the hand-off's tie probe,
// k_handoff, evaluates the snapshot both ways to measure what the reference's tree decided there and this rule can only guess)
template <bool FLIP = false>
DSA_HD bool ref_alive_rec(const SourceDesc& s, const Rec& r, uint64_t rstar, int eid, int iz, int ix, int flip = -1)
{
    if (t_pinned(r.T)) return true;
    const uint64_t rk = accept_rank(r.T, r.tau);
    const int id = (ix - 1) * s.rnz + (iz - 1);
    const bool alive = rk < rstar || (rk == rstar && id < eid);
    return (FLIP && id == flip) ? !alive : alive;
}

// classify node (iz, ix): returns status (-1, 0, 1) and the value to keep in *tout.  (Round 5: the eight stencil records of a node that is
// not alive are fetched together, before the first is looked at -- the hand-off kernel used to wait for them one after the other.)
template <bool FLIP = false>
DSA_HD int handoff_node(const GridDesc& g, const SourceDesc& s, const SourceScratch& w, int ended,
                        uint64_t rstar, int ez, int ex, int iz, int ix, float* tout, int flip = -1)
{
    const Rec own = w.F_r[rec_index(s.nbz_r, iz - 1, ix - 1)];
    const float raw = own.T;
    if (ended) {
        if (t_pinned(raw)) { *tout = t_value(raw); return 0; }
        if (t_value(raw) < kInf) { *tout = raw; return 1; }
        *tout = kInf; return -1;
    }
    const int eid = ex > 0 ? (ex - 1) * s.rnz + (ez - 1) : -1;
    if (ref_alive_rec<FLIP>(s, own, rstar, eid, iz, ix, flip)) { *tout = t_value(raw); return 0; }
    // not alive: close iff it touches an alive node; its value is the trial value from the alive
    // set (the edge node that ended the stage is alive but was never propagated)
    Stencil st;
    const int jx[2] = { ix - 1, ix + 1 }, jx2[2] = { ix - 2, ix + 2 };
    const int kz[2] = { iz - 1, iz + 1 }, kz2[2] = { iz - 2, iz + 2 };
    const Rec far = { kInf, kInf };
    Rec rj[2], rj2[2], rk[2], rk2[2];
    bool ej2[2], ek2[2];
    for (int d = 0; d < 2; ++d) {
        st.ej[d] = jx[d] >= 1 && jx[d] <= s.rnx;
        ej2[d] = jx2[d] >= 1 && jx2[d] <= s.rnx;
        st.ek[d] = kz[d] >= 1 && kz[d] <= s.rnz;
        ek2[d] = kz2[d] >= 1 && kz2[d] <= s.rnz;
        // (a neighbour outside the box reads the node's own record, which exists, and the value is dropped: no branch around a load)
        rj[d] = w.F_r[rec_index(s.nbz_r, iz - 1, (st.ej[d] ? jx[d] : ix) - 1)];
        rj2[d] = w.F_r[rec_index(s.nbz_r, iz - 1, (ej2[d] ? jx2[d] : ix) - 1)];
        rk[d] = w.F_r[rec_index(s.nbz_r, (st.ek[d] ? kz[d] : iz) - 1, ix - 1)];
        rk2[d] = w.F_r[rec_index(s.nbz_r, (ek2[d] ? kz2[d] : iz) - 1, ix - 1)];
    }
    for (int d = 0; d < 2; ++d) {
        if (!st.ej[d]) rj[d] = far;
        if (!ej2[d]) rj2[d] = far;
        if (!st.ek[d]) rk[d] = far;
        if (!ek2[d]) rk2[d] = far;
    }
    bool touch = false;
    for (int d = 0; d < 2; ++d) {
        st.aj[d] = st.ej[d] && ref_alive_rec<FLIP>(s, rj[d], rstar, eid, iz, jx[d], flip);
        st.tj[d] = st.aj[d] ? t_value(rj[d].T) : kInf;
        const bool o = ej2[d] && ref_alive_rec<FLIP>(s, rj2[d], rstar, eid, iz, jx2[d], flip);
        st.oj[d] = o;
        st.tj2[d] = o ? t_value(rj2[d].T) : kInf;
        st.ak[d] = st.ek[d] && ref_alive_rec<FLIP>(s, rk[d], rstar, eid, kz[d], ix, flip);
        st.tk[d] = st.ak[d] ? t_value(rk[d].T) : kInf;
        const bool p = ek2[d] && ref_alive_rec<FLIP>(s, rk2[d], rstar, eid, kz2[d], ix, flip);
        st.ok[d] = p;
        st.tk2[d] = p ? t_value(rk2[d].T) : kInf;
        touch = touch || st.aj[d] || st.ak[d];
    }
    if (!touch) { *tout = kInf; return -1; }
    // The reference's trial value of a node in the narrow band is the one fouds2 wrote at the LAST acceptance of one of its near neighbours (:431-485:
    // only the four near neighbours of an accepted node are evaluated again) -- from the nodes alive THEN.  An outer neighbour accepted after that
    // never reached this node's value.  In key order that cannot matter (an outer node accepted later than the near one in front of it is not smaller,
    // so the second-order leg is not taken either way); it does where the march ran out of key order: the source cell's corners, accepted by the
    // start-up march under their raised keys (a corner at 0.0092 s third, its neighbour at 0.0050 s sixth: the node beyond the corner keeps the
    // first-order value of accept 3).  So an outer node counts only if it was accepted no later than the last of the alive near neighbours
    // (accept_rank orders pinned nodes by their accept numbers, ahead of every other node).
    {
        uint64_t rlast = 0;
        for (int d = 0; d < 2; ++d) {
            const uint64_t a = st.aj[d] ? accept_rank(rj[d].T, rj[d].tau) : 0, c = st.ak[d] ? accept_rank(rk[d].T, rk[d].tau) : 0;
            rlast = a > rlast ? a : rlast; rlast = c > rlast ? c : rlast;
        }
        for (int d = 0; d < 2; ++d) {
            if (st.oj[d] && accept_rank(rj2[d].T, rj2[d].tau) > rlast) { st.oj[d] = false; st.tj2[d] = kInf; }
            if (st.ok[d] && accept_rank(rk2[d].T, rk2[d].tau) > rlast) { st.ok[d] = false; st.tk2[d] = kInf; }
        }
    }
    NodeGeom ng = { g.earth, w.risti_r[ix - 1], s.rdnx, s.rdnz };
    *tout = fouds2(st, w.slow_r[rec_index(s.nbz_r, iz - 1, ix - 1)], ng);
    if (iz == ez && ix == ex) return 0;
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Band march on the coarse grid: travel(urg=2) from the injected state until every node that
// started in the tree has been accepted.  W holds the (T, tau) records of the coarse march window only, (cwnz, cwnx)
// column-major (the hand-off put plain values there for status >= 0, +inf elsewhere); slow_c (tiled) / risti_c are the
// period's coarse tables.  Serial.  On return: alive nodes of the window are pinned (sign bit of T; tau = their accept
// number, 0 for the hand-off's alive nodes), all others +inf (T and tau).
// The pieces of the band march (the device kernel runs the loops over the window with a whole wavefront and only the tree with one lane):
DSA_HD MarchView band_march_view(const GridDesc& g, const SourceDesc& s, const SourceScratch& w, Rec* W, const float* slow_c, const float* risti_c)
{
    MarchView m;
    m.F = W; m.window = 1; m.slow = slow_c; m.nbz = g.nbz; m.risti = risti_c;
    m.status = w.cst; m.wz0 = s.cwz0; m.wx0 = s.cwx0; m.wnz = s.cwnz; m.wnx = s.cwnx;
    m.nnz = g.nnz; m.nnx = g.nnx; m.ri = g.earth; m.dnx = g.dnx; m.dnz = g.dnz;
    m.heap = w.heap; m.cap = kHeapCap; m.ntr = 0; m.error = 0; m.clock = 0.0f;
    return m;
}
// accept steps until the nodes that were in the tree at the start (cinit, `ninit` of them) are popped
DSA_HD void band_march_run(MarchView& m, const SourceDesc& s, SourceScratch& w, int ninit)
{
    while (m.ntr > 0 && ninit > 0 && m.error == 0) {
        const int ix = hp_ix(m.heap[1]), iz = hp_iz(m.heap[1]);
        const int q = (ix - 1 - s.cwx0) * s.cwnz + (iz - 1 - s.cwz0);
        if (w.cinit[q]) { w.cinit[q] = 0; --ninit; }
        if (!mv_accept_root(m)) break;
    }
    if (m.error) w.flags[1] = 16 + m.error;
}
// (round 6) The fixed point takes over from the band march on the assumption that from here on the reference accepts in the order of the keys -- true
// while its tree is a heap.  It need not be one when the last injected node has gone: an update that RAISED a key during the band march (updtree only
// moves an entry towards the root, CalSurfG.f90:899-920) leaves that entry above smaller ones, and the nodes beneath it are accepted late -- a node
// with time 0.2933 s after its neighbour with 0.2952 s in the case that showed it (profiles/r06_tie_diagnose_band_tree.log: 3.9e-4 s at the next
// node, no tie anywhere).  So the serial march goes on, accept by accept, until the tree is a heap again (at most kBandExtra accepts, and only
// while the root's neighbours lie inside the march's window); a tree that is still no heap then is reported (the unit is flagged for the march).
constexpr int kBandExtra = 96;
// slots s0, s0 + stride, ... of the tree: no entry smaller than its parent?
DSA_HD bool mv_heap_valid(MarchView& m, int s0, int stride)
{
    bool ok = true;
    for (int sl = s0; sl <= m.ntr; sl += stride) ok = ok && !(mv_key(m, sl) < mv_key(m, sl >> 1));      // (s0 >= 2)
    return ok;
}
// the root's four neighbours (those inside the grid) lie inside the march's window: an accept of the root cannot run into the window's edge
DSA_HD bool mv_root_inside(const MarchView& m)
{
    const int ix = hp_ix(m.heap[1]), iz = hp_iz(m.heap[1]);
    bool ok = true;
    if (ix - 1 >= 1) ok = ok && mv_inwin(m, iz, ix - 1);
    if (ix + 1 <= m.nnx) ok = ok && mv_inwin(m, iz, ix + 1);
    if (iz - 1 >= 1) ok = ok && mv_inwin(m, iz - 1, ix);
    if (iz + 1 <= m.nnz) ok = ok && mv_inwin(m, iz + 1, ix);
    return ok;
}
// the serial form (host tools / CPU checks; the device kernel checks the tree with a whole wavefront): returns 1 when the tree is left no heap
DSA_HD int band_march_settle(MarchView& m)
{
    for (int extra = 0; m.ntr > 1 && m.error == 0; ++extra) {
        if (mv_heap_valid(m, 2, 1)) return 0;
        if (extra >= kBandExtra || !mv_root_inside(m)) return 1;
        if (!mv_accept_root(m)) return 1;
    }
    return 0;
}

// window node q after the march: alive -> pinned (tau: accept number of the march, 0 for the hand-off's alive nodes), else unreached
DSA_HD void band_march_finish_node(const SourceScratch& w, Rec* W, int q)
{
    Rec& r = W[q];
    if (w.cst[q] == 0) { r.T = -t_value(r.T); }
    else { r.T = kInf; r.tau = kInf; }
}

DSA_HD void coarse_band_march(const GridDesc& g, const SourceDesc& s, SourceScratch& w, Rec* W,
                              const float* slow_c, const float* risti_c)
{
    MarchView m = band_march_view(g, s, w, W, slow_c, risti_c);
    int ninit = 0;
    // tree start order of the reference: ix outer, iz inner (:341-347)
    for (int lx = 0; lx < s.cwnx; ++lx)
        for (int lz = 0; lz < s.cwnz; ++lz) {
            const int q = lx * s.cwnz + lz;
            w.cinit[q] = 0;
            if (w.cst[q] == 0) W[q].tau = 0.0f;      // alive before the march
            if (w.cst[q] > 0) { w.cinit[q] = 1; ++ninit; mv_add(m, s.cwz0 + lz + 1, s.cwx0 + lx + 1); }
        }
    band_march_run(m, s, w, ninit);
    (void)band_march_settle(m);
    for (int q = 0; q < s.cwnx * s.cwnz; ++q) band_march_finish_node(w, W, q);
}

// the window's pinned nodes into a full-field record array (+inf everywhere else on entry): host tools / CPU checks
DSA_HD void export_window_records(const GridDesc& g, const SourceDesc& s, const Rec* W, Rec* F_c)
{
    for (int lx = 0; lx < s.cwnx; ++lx)
        for (int lz = 0; lz < s.cwnz; ++lz) {
            const Rec r = W[lx * s.cwnz + lz];
            if (t_pinned(r.T)) F_c[rec_index(g.nbz, s.cwz0 + lz, s.cwx0 + lx)] = r;
        }
}

// the window's pinned nodes into the compact coarse field (+inf everywhere on entry) and its exception table (empty on
// entry); returns false when the table overflows (cannot: the window has at most kCWinMax^2 nodes)
DSA_HD bool export_window_compact(const GridDesc& g, const SourceDesc& s, const Rec* W, float* Tc, unsigned long long* exc, int log2cap)
{
    bool ok = true;
    for (int lx = 0; lx < s.cwnx; ++lx)
        for (int lz = 0; lz < s.cwnz; ++lz) {
            const Rec r = W[lx * s.cwnz + lz];
            if (!t_pinned(r.T)) continue;
            const int id = rec_index(g.nbz, s.cwz0 + lz, s.cwx0 + lx);
            ok = exc_insert_serial(exc, log2cap, id | kExcPinned, r.tau) && ok;
            Tc[id] = r.T;                                      // -T: the sign bit marks the exceptional node
        }
    return ok;
}

}  // namespace dsa
