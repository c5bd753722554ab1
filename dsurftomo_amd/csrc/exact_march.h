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
//     its key, so a sift step is one LDS access and no field access;
//   * per accept step, lanes 0..3 each own one neighbour of the accepted node: they fetch its nine-point neighbourhood
//     (ten independent loads per lane, one memory round trip for the step) BEFORE the root is sifted down, and evaluate
//     the stencil afterwards, so the tree work hides the memory latency;
//   * the neighbour statuses fetched before the sift are stale for tree entries the step itself moves; a log of the
//     step's (node, slot) assignments in LDS corrects them (looked up by all lanes at once);
//   * everything that is sequential (tree, statuses) is executed uniformly by the wave, stores by lane 0.
// Written __host__ __device__ so that tests/hostcheck.cpp can run the same logic on a CPU against the oracle.
#pragma once

#include "source_stage.h"

namespace dsa {

struct XEntry { float key; int node; };     // node = (iz << 16) | ix, 1-based
struct XRec { float T; int st; };           // st: -1 far, 0 alive, > 0 slot in the tree (reference nsts, CalSurfG.f90:227)

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
    float ri, dnx, dnz;
    DSA_LDS XEntry* hl;   // tree slots 1..lcap at hl[1..lcap]
    int lcap;
    XEntry* hg;           // tree slots lcap+1 .. lcap+gcap at hg[0..gcap-1]
    int gcap;
    int ntr;
    int error;            // 1: tree capacity, 2: log capacity
    DSA_LDS int* log;     // 2 * kXLogCap ints
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
DSA_HD int x_iz(int node) { return node >> 16; }
DSA_HD int x_ix(int node) { return node & 0xffff; }
DSA_HD int x_id(const XMarch& m, int node) { return rec_index(m.nbz, x_iz(node) - 1, x_ix(node) - 1); }

DSA_HD XEntry xh_get(const XMarch& m, int s) { return s <= m.lcap ? m.hl[s] : m.hg[s - m.lcap - 1]; }
// entry into slot s: the tree, the node's status, the step's log
DSA_HD void xh_put(XMarch& m, int s, XEntry e)
{
    if (x_lane() == 0) {
        if (s <= m.lcap) m.hl[s] = e; else m.hg[s - m.lcap - 1] = e;
        m.F[x_id(m, e.node)].st = s;
        if (m.nlog < kXLogCap) { m.log[2 * m.nlog] = e.node; m.log[2 * m.nlog + 1] = s; }
    }
    if (m.nlog < kXLogCap) m.nlog += 1; else m.error = 2;
}
// slot of a node that was in the tree when its status was fetched at the start of the step: the step's own moves come first
DSA_HD int x_current_slot(const XMarch& m, int node, int fetched)
{
    int best = -1;
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = x_lane();
    for (int base = 0; base < m.nlog; base += 64) {
        const int i = base + lane;
        const bool hit = i < m.nlog && m.log[2 * i] == node;
        const unsigned long long b = __ballot(hit);
        if (b) best = base + 63 - __clzll((long long)b);
    }
#else
    for (int i = 0; i < m.nlog; ++i) if (m.log[2 * i] == node) best = i;
#endif
    return best >= 0 ? m.log[2 * best + 1] : fetched;
}

// reference updtree / the tail of addtree (CalSurfG.f90:768-790, :906-920): towards the root while strictly smaller
DSA_HD void x_sift_up(XMarch& m, XEntry e, int tpc)
{
    int tpp = tpc >> 1;
    while (tpp > 0) {
        const XEntry p = xh_get(m, tpp);
        if (e.key < p.key) { xh_put(m, tpc, p); tpc = tpp; tpp = tpc >> 1; }
        else break;
    }
    xh_put(m, tpc, e);
}
DSA_HD void x_add(XMarch& m, int node, float key)
{
    if (m.ntr + 1 > m.lcap + m.gcap) { m.error = 1; return; }
    m.ntr += 1;
    x_sift_up(m, XEntry{ key, node }, m.ntr);
}
// reference downtree (CalSurfG.f90:800-858): the last entry replaces the root and sinks; of two children with equal keys the
// left one is taken (`>`), a child moves up only when strictly smaller
DSA_HD void x_pop_root(XMarch& m)
{
    if (m.ntr == 1) { m.ntr = 0; return; }
    const XEntry e = xh_get(m, m.ntr);
    m.ntr -= 1;
    int tpp = 1, tpc = 2;
    while (tpc < m.ntr) {
        XEntry a, b;
        if (tpc + 1 <= m.lcap) { a = m.hl[tpc]; b = m.hl[tpc + 1]; }          // the two children sit side by side (one 16-byte read)
        else { a = xh_get(m, tpc); b = xh_get(m, tpc + 1); }
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

// the nine-point neighbourhood of neighbour (nz, nx) of the accepted node, as fetched (before the step changes anything)
struct XRaw { int in; XRec own; XRec a[8]; float slown; float risti; };
DSA_HD XRaw x_fetch(const XMarch& m, int nz, int nx)
{
    XRaw r;
    r.in = nx >= 1 && nx <= m.nnx && nz >= 1 && nz <= m.nnz;
    r.own = XRec{ 0.0f, 0 }; r.slown = 1.0f; r.risti = 1.0f;
    for (int q = 0; q < 8; ++q) r.a[q] = XRec{ kInf, -1 };
    if (!r.in) return r;
    const int id = rec_index(m.nbz, nz - 1, nx - 1);
    int nid[8];
    rec_stencil(m.nbz, id, nid);
    const bool inq[8] = { nx > 1, nx < m.nnx, nz > 1, nz < m.nnz, nx > 2, nx + 1 < m.nnx, nz > 2, nz + 1 < m.nnz };
    r.own = m.F[id];
    for (int q = 0; q < 8; ++q) if (inq[q]) r.a[q] = m.F[nid[q]];
    r.slown = m.slow[id];
    r.risti = m.risti[nx - 1];
    return r;
}
// its trial value from the alive set, the node being accepted (root, value troot) included; reference fouds2 (CalSurfG.f90:587-759)
DSA_HD float x_trial(const XMarch& m, const XRaw& r, int nz, int nx, int root, float troot)
{
    const int nn[8] = { (nz << 16) | (nx - 1), (nz << 16) | (nx + 1), ((nz - 1) << 16) | nx, ((nz + 1) << 16) | nx,
                        (nz << 16) | (nx - 2), (nz << 16) | (nx + 2), ((nz - 2) << 16) | nx, ((nz + 2) << 16) | nx };
    const bool inq[8] = { nx > 1, nx < m.nnx, nz > 1, nz < m.nnz, nx > 2, nx + 1 < m.nnx, nz > 2, nz + 1 < m.nnz };
    bool al[8];
    float t[8];
    for (int q = 0; q < 8; ++q) {
        const bool is_root = inq[q] && nn[q] == root;
        al[q] = inq[q] && (r.a[q].st == 0 || is_root);
        t[q] = al[q] ? (is_root ? troot : r.a[q].T) : kInf;
    }
    Stencil s;
    for (int d = 0; d < 2; ++d) {
        s.ej[d] = inq[d];     s.aj[d] = al[d];     s.tj[d] = t[d];
        s.oj[d] = al[4 + d];  s.tj2[d] = t[4 + d];
        s.ek[d] = inq[2 + d]; s.ak[d] = al[2 + d]; s.tk[d] = t[2 + d];
        s.ok[d] = al[6 + d];  s.tk2[d] = t[6 + d];
    }
    const NodeGeom g = { m.ri, r.risti, m.dnx, m.dnz };
    return fouds2(s, r.slown, g);
}

// One accept step of reference travel (CalSurfG.f90:417-485): the root becomes alive, leaves the tree, and its four neighbours
// x-, x+, z-, z+ (in that order) get a new trial value and enter the tree / move in it.
DSA_HD void x_accept_root(XMarch& m, XEntry root)
{
    const int iz = x_iz(root.node), ix = x_ix(root.node);
    const int nzq[4] = { iz, iz, iz - 1, iz + 1 }, nxq[4] = { ix - 1, ix + 1, ix, ix };
    int nb_in[4], nb_st[4];
    float nb_trial[4];
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = x_lane();
    const int ql = lane & 3;
    const int mz = ql == 0 ? nzq[0] : ql == 1 ? nzq[1] : ql == 2 ? nzq[2] : nzq[3];
    const int mx = ql == 0 ? nxq[0] : ql == 1 ? nxq[1] : ql == 2 ? nxq[2] : nxq[3];
    XRaw raw;
    raw.in = 0;
    if (lane < 4) raw = x_fetch(m, mz, mx);                    // ten loads per lane in flight
#else
    XRaw raws[4];
    for (int q = 0; q < 4; ++q) raws[q] = x_fetch(m, nzq[q], nxq[q]);
#endif
    m.nlog = 0;
    if (x_lane() == 0) m.F[x_id(m, root.node)].st = 0;
    x_pop_root(m);
#if defined(__HIP_DEVICE_COMPILE__)
    float trial = kInf;
    int st = 0, in = 0;
    if (lane < 4) {
        in = raw.in; st = raw.own.st;
        if (in && st != 0) trial = x_trial(m, raw, mz, mx, root.node, root.key);
    }
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = __builtin_amdgcn_readlane(in, q);
        nb_st[q] = __builtin_amdgcn_readlane(st, q);
        nb_trial[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(trial), q));
    }
#else
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = raws[q].in; nb_st[q] = raws[q].own.st; nb_trial[q] = kInf;
        if (nb_in[q] && nb_st[q] != 0) nb_trial[q] = x_trial(m, raws[q], nzq[q], nxq[q], root.node, root.key);
    }
#endif
    for (int q = 0; q < 4; ++q) {
        if (!nb_in[q] || nb_st[q] == 0) continue;
        const int node = (nzq[q] << 16) | nxq[q];
        if (x_lane() == 0) m.F[x_id(m, node)].T = nb_trial[q];       // fouds2 overwrites unconditionally (:758)
        if (nb_st[q] < 0) x_add(m, node, nb_trial[q]);
        else x_sift_up(m, XEntry{ nb_trial[q], node }, x_current_slot(m, node, nb_st[q]));
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
        if (REFINED && is_open_edge(sd, x_iz(root.node), x_ix(root.node))) {
            if (x_lane() == 0) m.F[x_id(m, root.node)].st = 0;
            break;
        }
        x_accept_root(m, root);
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
            const float t = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
            const int node = ((s.isz_r - 1 + j) << 16) | (s.isx_r - 1 + i);
            if (x_lane() == 0) m.F[x_id(m, node)].T = t;
            x_add(m, node, t);
        }
}

}  // namespace dsa
