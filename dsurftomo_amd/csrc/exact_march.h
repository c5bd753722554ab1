// Exact mode: the reference's Fast Marching itself, one wavefront per (period, source) unit.
//
// The fixed-point solve (fim_kernel.hip) lands on the reference's travel times everywhere except downstream of exact time
// ties between neighbouring narrow-band nodes, where the reference's answer depends on which of the two its binary tree
// happens to hold nearer the root (reference CalSurfG.f90:417-485 with the tree of :768-921).  That cannot be
// derived locally, so for the units the tie detector flags (engine option `exact_ties`) the march is replayed literally:
// `travel(urg=1)` on the refined box, the hand-off of :1287-1349, `travel(urg=2)` on the propagation grid -- the same tree,
// the same insertion order, the same comparisons -- and the result is bit-identical to the reference's.
//
// How a serial algorithm is laid on a wavefront:
//   * the tree lives in LDS (slots 1..lcap; deeper slots, if a front ever needs them, in global memory), each entry carrying
//     its key and the node's record index, so a sift step is one LDS access and no field access;
//   * per accept step, lanes 0..3 each own one neighbour of the accepted node: they fetch its nine-point neighbourhood
//     (ten independent loads per lane, one memory round trip for the step) BEFORE the root is sifted down, and evaluate
//     the stencil afterwards, so the tree work hides the memory latency;
//   * a neighbour's tree slot fetched before the sift is stale when the step itself moved that entry; the slot is checked
//     against the tree (one LDS read) and, when stale, looked up in the step's log of (node, slot) assignments in LDS by all
//     lanes at once;
//   * everything that is sequential (tree, statuses) is wave-uniform: values read from LDS are made scalar (readfirstlane),
//     so index arithmetic and branches run on the scalar unit; stores are issued by lane 0.
// Written __host__ __device__ so that tests/hostcheck.cpp can run the same logic on a CPU against the oracle.
#pragma once

#include "source_stage.h"

namespace dsa {

struct XEntry { float key; int id; };       // id = record index of the node in the tiled field (eikonal_core.h rec_index)
struct XRec { float T; int st; };           // st: -1 far, 0 alive, > 0 slot in the tree (reference nsts, CalSurfG.f90:227)
struct XLog { int id; int slot; };

#if defined(__HIP_DEVICE_COMPILE__)
#define DSA_LDS __attribute__((address_space(3)))
#else
#define DSA_LDS
#endif

constexpr int kXLogCap = 128;               // (node, slot) assignments one accept step can make (5 sifts of <= 25 levels)
constexpr int kXStage = kRefTiles * kRefTiles;   // coarse nodes under the refined box (17 x 17)

struct XMarch {
    XRec* F;              // tiled records of the grid being marched
    const float* slow;    // tiled slowness
    const float* risti;   // per ix (0-based)
    int nbz, nnx, nnz;
    unsigned nbz_inv;     // ceil(2^32 / nbz): tile -> (bx, bz) without a division
    float ri, dnx, dnz;
    DSA_LDS XEntry* hl;   // tree slots 1..lcap at hl[1..lcap]
    int lcap;
    XEntry* hg;           // tree slots lcap+1 .. lcap+gcap at hg[0..gcap-1]
    int gcap;
    int ntr;
    int error;            // 1: tree capacity, 2: log capacity
    DSA_LDS XLog* log;    // kXLogCap entries
    int nlog;
    unsigned pops;
};

DSA_HD int x_lane()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(threadIdx.x & 63u);
#else
    return 0;
#endif
}
// a value every lane holds alike, told to the compiler (scalar register)
DSA_HD int x_uni(int v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
DSA_HD float x_unif(float v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
#else
    return v;
#endif
}
DSA_HD void x_set_grid(XMarch& m, int nbz, int nnx, int nnz)
{
    m.nbz = nbz; m.nnx = nnx; m.nnz = nnz;
    m.nbz_inv = nbz > 1 ? 0xffffffffu / (unsigned)nbz + 1u : 0u;
}
// 0-based coordinates of record id
DSA_HD void x_coords(const XMarch& m, int id, int* iz0, int* ix0)
{
    const unsigned tile = (unsigned)id >> 6;
    const unsigned bx = m.nbz > 1 ? (unsigned)(((unsigned long long)tile * m.nbz_inv) >> 32) : tile;
    const unsigned bz = tile - bx * (unsigned)m.nbz;
    *ix0 = (int)(bx << kTileShift) + rec_ix_in_tile(id);
    *iz0 = (int)(bz << kTileShift) + rec_iz_in_tile(id);
}

DSA_HD XEntry xh_get(const XMarch& m, int s)
{
    const XEntry e = s <= m.lcap ? m.hl[s] : m.hg[s - m.lcap - 1];
    return XEntry{ x_unif(e.key), x_uni(e.id) };
}
// entry into slot s: the tree, the node's status, the step's log
DSA_HD void xh_put(XMarch& m, int s, XEntry e)
{
    m.nlog = x_uni(m.nlog);
    if (m.nlog >= kXLogCap) { m.error = 2; return; }
    if (x_lane() == 0) {
        if (s <= m.lcap) m.hl[s] = e; else m.hg[s - m.lcap - 1] = e;
        m.F[e.id].st = s;
        m.log[m.nlog] = XLog{ e.id, s };
    }
    m.nlog += 1;
}
// slot of a node that was in the tree when its status was fetched at the start of the step: unless the step itself moved it
DSA_HD int x_current_slot(const XMarch& m, int id, int fetched)
{
    if (fetched <= m.ntr && xh_get(m, fetched).id == id) return fetched;
    int best = -1;
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = x_lane();
    for (int base = 0; base < m.nlog; base += 64) {
        const int i = base + lane;
        const bool hit = i < m.nlog && m.log[i].id == id;
        const unsigned long long b = __ballot(hit);
        if (b) best = base + 63 - __clzll((long long)b);
    }
    return x_uni(best >= 0 ? m.log[best].slot : fetched);
#else
    for (int i = 0; i < m.nlog; ++i) if (m.log[i].id == id) best = i;
    return best >= 0 ? m.log[best].slot : fetched;
#endif
}

// reference updtree / the tail of addtree (CalSurfG.f90:768-790, :906-920): towards the root while strictly smaller
DSA_HD void x_sift_up(XMarch& m, XEntry e, int tpc)
{
    tpc = x_uni(tpc);
    int tpp = tpc >> 1;
    while (tpp > 0) {
        const XEntry p = xh_get(m, tpp);
        if (e.key < p.key) { xh_put(m, tpc, p); tpc = tpp; tpp = tpc >> 1; }
        else break;
    }
    xh_put(m, tpc, e);
}
DSA_HD void x_add(XMarch& m, int id, float key)
{
    if (m.ntr + 1 > m.lcap + m.gcap) { m.error = 1; return; }
    m.ntr += 1;
    x_sift_up(m, XEntry{ key, id }, m.ntr);
}
// reference downtree (CalSurfG.f90:800-858): the last entry replaces the root and sinks; of two children with equal keys the
// left one is taken (`>`), a child moves up only when strictly smaller
DSA_HD void x_pop_root(XMarch& m)
{
    m.ntr = x_uni(m.ntr);
    if (m.ntr == 1) { m.ntr = 0; return; }
    const XEntry e = xh_get(m, m.ntr);
    m.ntr -= 1;
    int tpp = 1, tpc = 2;
    while (tpc < m.ntr) {
        XEntry a, b;
        if (tpc + 1 <= m.lcap) {                                              // the two children sit side by side (one 16-byte read)
            const XEntry a0 = m.hl[tpc], b0 = m.hl[tpc + 1];
            a = XEntry{ x_unif(a0.key), x_uni(a0.id) }; b = XEntry{ x_unif(b0.key), x_uni(b0.id) };
        } else { a = xh_get(m, tpc); b = xh_get(m, tpc + 1); }
        if (a.key > b.key) { a = b; tpc += 1; }
        if (a.key < e.key) { xh_put(m, tpp, a); tpp = tpc; tpc = 2 * tpp; }
        else tpc = m.ntr + 1;
    }
    if (tpc == m.ntr) {
        const XEntry a = xh_get(m, tpc);
        if (a.key < e.key) { xh_put(m, tpp, a); tpp = tpc; }
    }
    xh_put(m, tpp, e);
}

// the nine-point neighbourhood of a neighbour (record nid, 1-based coordinates (nz, nx)) of the accepted node, as fetched
// (before the step changes anything)
struct XRaw { int in; int nid[8]; XRec own; XRec a[8]; float slown; float risti; };
DSA_HD XRaw x_fetch(const XMarch& m, int id, int nz, int nx)
{
    XRaw r;
    r.in = nx >= 1 && nx <= m.nnx && nz >= 1 && nz <= m.nnz;
    r.own = XRec{ 0.0f, 0 }; r.slown = 1.0f; r.risti = 1.0f;
    for (int q = 0; q < 8; ++q) { r.a[q] = XRec{ kInf, -1 }; r.nid[q] = -1; }
    if (!r.in) return r;
    rec_stencil(m.nbz, id, r.nid);
    const bool inq[8] = { nx > 1, nx < m.nnx, nz > 1, nz < m.nnz, nx > 2, nx + 1 < m.nnx, nz > 2, nz + 1 < m.nnz };
    r.own = m.F[id];
    for (int q = 0; q < 8; ++q) { if (inq[q]) r.a[q] = m.F[r.nid[q]]; else r.nid[q] = -1; }
    r.slown = m.slow[id];
    r.risti = m.risti[nx - 1];
    return r;
}
// its trial value from the alive set, the node being accepted (root) included; reference fouds2 (CalSurfG.f90:587-759)
DSA_HD float x_trial(const XMarch& m, const XRaw& r, XEntry root)
{
    bool al[8];
    float t[8];
    for (int q = 0; q < 8; ++q) {
        const bool is_root = r.nid[q] == root.id;
        al[q] = r.nid[q] >= 0 && (r.a[q].st == 0 || is_root);
        t[q] = al[q] ? (is_root ? root.key : r.a[q].T) : kInf;
    }
    Stencil s;
    for (int d = 0; d < 2; ++d) {
        s.ej[d] = r.nid[d] >= 0;     s.aj[d] = al[d];     s.tj[d] = t[d];
        s.oj[d] = al[4 + d];         s.tj2[d] = t[4 + d];
        s.ek[d] = r.nid[2 + d] >= 0; s.ak[d] = al[2 + d]; s.tk[d] = t[2 + d];
        s.ok[d] = al[6 + d];         s.tk2[d] = t[6 + d];
    }
    const NodeGeom g = { m.ri, r.risti, m.dnx, m.dnz };
    return fouds2(s, r.slown, g);
}

// One accept step of reference travel (CalSurfG.f90:417-485): the root becomes alive, leaves the tree, and its four neighbours
// x-, x+, z-, z+ (in that order) get a new trial value and enter the tree / move in it.
DSA_HD void x_accept_root(XMarch& m, XEntry root, int iz0, int ix0)
{
    const int iz = iz0 + 1, ix = ix0 + 1;
    int rid[8];
    rec_stencil(m.nbz, root.id, rid);
    const int nzq[4] = { iz, iz, iz - 1, iz + 1 }, nxq[4] = { ix - 1, ix + 1, ix, ix };
    int nb_in[4], nb_st[4];
    float nb_trial[4];
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = x_lane();
    const int ql = lane & 3;
    const int mz = ql == 0 ? nzq[0] : ql == 1 ? nzq[1] : ql == 2 ? nzq[2] : nzq[3];
    const int mx = ql == 0 ? nxq[0] : ql == 1 ? nxq[1] : ql == 2 ? nxq[2] : nxq[3];
    const int mid = ql == 0 ? rid[0] : ql == 1 ? rid[1] : ql == 2 ? rid[2] : rid[3];
    XRaw raw;
    raw.in = 0;
    if (lane < 4) raw = x_fetch(m, mid, mz, mx);               // ten loads per lane in flight
#else
    XRaw raws[4];
    for (int q = 0; q < 4; ++q) raws[q] = x_fetch(m, rid[q], nzq[q], nxq[q]);
#endif
    m.nlog = 0;
    if (x_lane() == 0) m.F[root.id].st = 0;
    x_pop_root(m);
#if defined(__HIP_DEVICE_COMPILE__)
    float trial = kInf;
    int st = 0, in = 0;
    if (lane < 4) {
        in = raw.in; st = raw.own.st;
        if (in && st != 0) trial = x_trial(m, raw, root);
    }
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = __builtin_amdgcn_readlane(in, q);
        nb_st[q] = __builtin_amdgcn_readlane(st, q);
        nb_trial[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(trial), q));
    }
#else
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = raws[q].in; nb_st[q] = raws[q].own.st; nb_trial[q] = kInf;
        if (nb_in[q] && nb_st[q] != 0) nb_trial[q] = x_trial(m, raws[q], root);
    }
#endif
    for (int q = 0; q < 4; ++q) {
        if (!nb_in[q] || nb_st[q] == 0) continue;
        if (x_lane() == 0) m.F[rid[q]].T = nb_trial[q];               // fouds2 overwrites unconditionally (:758)
        if (nb_st[q] < 0) x_add(m, rid[q], nb_trial[q]);
        else x_sift_up(m, XEntry{ nb_trial[q], rid[q] }, x_current_slot(m, rid[q], nb_st[q]));
    }
    m.pops += 1u;
}

// the march until the tree is empty; REFINED: reference's exit of the refined stage -- the root lies on an edge of the box that is
// not an edge of the model by the literal test of :396-407 -- marks that node alive and stops
template <bool REFINED>
DSA_HD void x_march(XMarch& m, const SourceDesc& sd)
{
    while (m.ntr > 0 && m.error == 0) {
        const XEntry root = xh_get(m, 1);
        int iz0, ix0;
        x_coords(m, root.id, &iz0, &ix0);
        if (REFINED && is_open_edge(sd, iz0 + 1, ix0 + 1)) {
            if (x_lane() == 0) m.F[root.id].st = 0;
            break;
        }
        x_accept_root(m, root, iz0, ix0);
    }
}

// start of the refined stage: the four corners of the source cell, values with distances in radians (:360-375)
DSA_HD void x_refined_start(XMarch& m, const SourceDesc& s, const float* vcorner)
{
    float vss[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) vss[i][j] = vcorner[i * 2 + j];
    const float vsrc = bilinear4(vss, s.rdnx, s.rdnz, s.dsx_r, s.dsz_r);
    for (int i = 1; i <= 2; ++i)
        for (int j = 1; j <= 2; ++j) {
            const float ds = sqrtf(sq(s.dsx_r - (float)(i - 1) * s.rdnx) + sq(s.dsz_r - (float)(j - 1) * s.rdnz));
            const float t = x_unif(2.0f * ds / (vss[i - 1][j - 1] + vsrc));
            const int id = rec_index(m.nbz, s.isz_r - 2 + j, s.isx_r - 2 + i);
            if (x_lane() == 0) m.F[id].T = t;
            x_add(m, id, t);
        }
}

}  // namespace dsa
