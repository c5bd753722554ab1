// Exact mode: the reference's Fast Marching itself (engine option `exact_ties`).
//
// The fixed-point solve (fim_kernel.hip) lands on the reference's travel times everywhere except downstream of exact time
// ties between neighbouring narrow-band nodes, where the reference's answer depends on which of the two its binary tree
// happens to hold nearer the root (reference CalSurfG.f90:417-485 with the tree of :768-921).  That cannot be
// derived locally, so the march is replayed literally: `travel(urg=1)` on the refined box, the hand-off of :1287-1349,
// `travel(urg=2)` on the propagation grid -- the same tree, the same insertion order, the same comparisons -- and the result is
// bit-identical to the reference's.
//
// This header holds what the device kernels (exact_kernel.hip: four units per wavefront, a group of sixteen lanes each) and the CPU
// model of the march (tests/hostcheck.cpp runs it against the oracle) share: the record / tree entry types, the quadrant form of the
// stencil -- sixteen (neighbour, quadrant) pairs per accept step, one per lane on the device -- and, for the CPU model only, a plain
// serial march `XMarch` with the tree split into a "near" and a "far" part like the device's LDS / global split.
#pragma once

#include "source_stage.h"

namespace dsa {

struct XEntry { float key; int id; };       // id = record index of the node in the tiled field (eikonal_core.h rec_index)
struct XRec { float T; int st; };           // st: -1 far, 0 alive, > 0 slot in the tree (reference nsts, CalSurfG.f90:227)

#if defined(__HIP_DEVICE_COMPILE__)
#define DSA_LDS __attribute__((address_space(3)))
#else
#define DSA_LDS
#endif

constexpr int kXStage = kRefTiles * kRefTiles;   // coarse nodes under the refined box (17 x 17)

// CPU model of one unit's march (the device keeps the same state per group of sixteen lanes: exact_kernel.hip XG)
struct XMarch {
    XRec* F;              // tiled records of the grid being marched
    const float* slow;    // tiled slowness
    const float* risti;   // per ix (0-based)
    int nbz, nnx, nnz;
    unsigned nbz_inv;     // ceil(2^32 / nbz): tile -> (bx, bz) without a division
    float ri, dnx, dnz;
    XEntry* hl;           // tree slots 1..lcap at hl[1..lcap] (the device: LDS)
    int lcap;
    XEntry* hg;           // tree slots lcap+1 .. lcap+gcap at hg[0..gcap-1] (the device: global memory)
    int gcap;
    int ntr;
    int error;            // 1: tree capacity
    unsigned pops;
};

template <class M>
DSA_HD void x_set_grid(M& m, int nbz, int nnx, int nnz)
{
    m.nbz = nbz; m.nnx = nnx; m.nnz = nnz;
    m.nbz_inv = nbz > 1 ? 0xffffffffu / (unsigned)nbz + 1u : 0u;
}
// 0-based coordinates of record id (M: any march state with nbz, nbz_inv)
template <class M>
DSA_HD void x_coords(const M& m, int id, int* iz0, int* ix0)
{
    const unsigned tile = (unsigned)id >> 6;
    const unsigned bx = m.nbz > 1 ? (unsigned)(((unsigned long long)tile * m.nbz_inv) >> 32) : tile;
    const unsigned bz = tile - bx * (unsigned)m.nbz;
    *ix0 = (int)(bx << kTileShift) + rec_ix_in_tile(id);
    *iz0 = (int)(bz << kTileShift) + rec_iz_in_tile(id);
}

DSA_HD XEntry xh_get(const XMarch& m, int s) { return s <= m.lcap ? m.hl[s] : m.hg[s - m.lcap - 1]; }
// entry into slot s: the tree and the node's status
DSA_HD void xh_put(XMarch& m, int s, XEntry e)
{
    if (s <= m.lcap) m.hl[s] = e; else m.hg[s - m.lcap - 1] = e;
    m.F[e.id].st = s;
}
// reference updtree / the tail of addtree (CalSurfG.f90:768-790, :906-920): towards the root while strictly smaller
DSA_HD void x_sift_up(XMarch& m, XEntry e, int tpc)
{
    int tpp = tpc >> 1;
    while (tpp > 0) {
        const XEntry p = xh_get(m, tpp);
        if (!(e.key < p.key)) break;
        xh_put(m, tpc, p);
        tpc = tpp; tpp = tpc >> 1;
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
    if (m.ntr == 1) { m.ntr = 0; return; }
    const XEntry e = xh_get(m, m.ntr);
    m.ntr -= 1;
    int tpp = 1, tpc = 2;
    while (tpc < m.ntr) {
        XEntry a = xh_get(m, tpc);
        const XEntry b = xh_get(m, tpc + 1);
        if (a.key > b.key) { a = b; tpc += 1; }
        if (!(a.key < e.key)) { tpc = m.ntr + 1; break; }
        xh_put(m, tpp, a); tpp = tpc; tpc = 2 * tpp;
    }
    if (tpc == m.ntr) {
        const XEntry a = xh_get(m, tpc);
        if (a.key < e.key) { xh_put(m, tpp, a); tpp = tpc; }
    }
    xh_put(m, tpp, e);
}

// ---- trial values of the four neighbours, laid on sixteen lanes -----------------------------------------------------------
// Lane 4 q + 2 j + k owns one quadrant (j: the x- / x+ side, k: the z- / z+ side) of neighbour q of the accepted node.  It fetches
// the neighbour's own record and slowness and the four stencil records of its quadrant (seven loads instead of eleven, and the
// sixteen lanes share their instructions), evaluates the candidates reference fouds2 (CalSurfG.f90:587-759) takes from that quadrant
// -- the one-sided step from j when some z neighbour inside the grid is not alive, the one-sided step from k likewise, the two-sided
// quadratic when both are alive, each by fouds2's own expression (eikonal_core.h) -- and the minimum over the four lanes of a
// neighbour is its trial value (the minimum does not depend on the order).  tests/test_hostcheck.py compares the march built on this
// with the oracle bit for bit; tests/hostcheck.cpp also compares x_trial_of_quads with fouds2 on random neighbourhoods.
struct XQuad {
    int in;                    // the neighbour lies inside the grid
    int idj, idj2, idk, idk2;  // record indices of the quadrant's four stencil nodes (-1: outside the grid)
    XRec own, rj, rj2, rk, rk2;
    float slown, risti;
};
// OWN = false: the neighbour's own record is not fetched (the caller reads its status later, behind the step's tree moves)
template <bool OWN = true, class M>
DSA_HD XQuad x_fetch_quad(const M& m, int id, int nz, int nx, int j, int k)
{
    XQuad r;
    r.in = nx >= 1 && nx <= m.nnx && nz >= 1 && nz <= m.nnz;
    r.own = XRec{ 0.0f, 0 }; r.slown = 1.0f; r.risti = 1.0f;
    r.rj = r.rj2 = r.rk = r.rk2 = XRec{ kInf, -1 };
    r.idj = r.idj2 = r.idk = r.idk2 = -1;
    if (!r.in) return r;
    int nid[8];
    rec_stencil(m.nbz, id, nid);
    const bool inj = j ? nx < m.nnx : nx > 1, inj2 = j ? nx + 1 < m.nnx : nx > 2;
    const bool ink = k ? nz < m.nnz : nz > 1, ink2 = k ? nz + 1 < m.nnz : nz > 2;
    r.idj = inj ? (j ? nid[1] : nid[0]) : -1;   r.idj2 = inj2 ? (j ? nid[5] : nid[4]) : -1;
    r.idk = ink ? (k ? nid[3] : nid[2]) : -1;   r.idk2 = ink2 ? (k ? nid[7] : nid[6]) : -1;
    if (OWN) r.own = m.F[id];
    if (inj) r.rj = m.F[r.idj];
    if (inj2) r.rj2 = m.F[r.idj2];
    if (ink) r.rk = m.F[r.idk];
    if (ink2) r.rk2 = m.F[r.idk2];
    r.slown = m.slow[id];
    r.risti = m.risti[nx - 1];
    return r;
}
// what a quadrant knows after the fetch: who is alive (the node being accepted counts, with its key as value) and the values
struct XQuadState { bool ej, aj, oj, ek, ak, ok; float tj, tj2, tk, tk2; };
DSA_HD XQuadState x_quad_state(const XQuad& r, XEntry root)
{
    XQuadState q;
    const bool rj = r.idj == root.id, rj2 = r.idj2 == root.id, rk = r.idk == root.id, rk2 = r.idk2 == root.id;
    q.ej = r.idj >= 0; q.ek = r.idk >= 0;
    q.aj = q.ej && (r.rj.st == 0 || rj);           q.tj = q.aj ? (rj ? root.key : r.rj.T) : kInf;
    q.oj = r.idj2 >= 0 && (r.rj2.st == 0 || rj2);  q.tj2 = q.oj ? (rj2 ? root.key : r.rj2.T) : kInf;
    q.ak = q.ek && (r.rk.st == 0 || rk);           q.tk = q.ak ? (rk ? root.key : r.rk.T) : kInf;
    q.ok = r.idk2 >= 0 && (r.rk2.st == 0 || rk2);  q.tk2 = q.ok ? (rk2 ? root.key : r.rk2.T) : kInf;
    return q;
}
// the candidates of one quadrant; k_dead / j_dead: some z / x neighbour of the node inside the grid is not alive (both sides looked at)
DSA_HD float x_quad_candidates(const XQuadState& s, bool k_dead, bool j_dead, float slown, const NodeGeom& g)
{
    const float ri = g.ri, risti = g.risti, dnx = g.dnx, dnz = g.dnz;
    const float s2 = sq(slown);
    const bool swj = s.ej && s.aj && s.oj && (s.tj > s.tj2);
    const bool swk = s.ek && s.ak && s.ok && (s.tk > s.tk2);
    float best = kInf;
    if (k_dead && s.ej && s.aj) {
        float trav;
        if (swj) { const float u = 2.0f * ri * dnx; trav = div3((4.0f * s.tj - s.tj2) + sqrt_pos(sq(u) * s2)); }
        else trav = s.tj + sqrt_pos(s2 * sq(ri) * sq(dnx));
        best = (trav < best) ? trav : best;
    }
    if (j_dead && s.ek && s.ak) {
        float trav;
        if (swk) { const float u = 2.0f * risti * dnz; trav = div3((4.0f * s.tk - s.tk2) + sqrt_pos(sq(u) * s2)); }
        else trav = s.tk + sqrt_pos(s2 * sq(risti) * sq(dnz));
        best = (trav < best) ? trav : best;
    }
    if (s.ej && s.aj && s.ek && s.ak) {
        float a, b, c, tref;
        bool third = false;
        if (swj) {
            if (swk) {
                const float u = 2.0f * ri * dnx;
                const float v = 2.0f * risti * dnz;
                float em = 4.0f * s.tj - s.tj2 - 4.0f * s.tk;
                em = em + s.tk2;
                a = sq(v) + sq(u);
                b = 2.0f * em * sq(u);
                c = sq(u) * (sq(em) - s2 * sq(v));
                tref = 4.0f * s.tj - s.tj2;
                third = true;
            } else {
                const float u = risti * dnz;
                const float v = 2.0f * ri * dnx;
                const float em = 3.0f * s.tk - 4.0f * s.tj + s.tj2;
                a = sq(v) + 9.0f * sq(u);
                b = 6.0f * em * sq(u);
                c = sq(u) * (sq(em) - s2 * sq(v));
                tref = s.tk;
            }
        } else {
            if (swk) {
                const float u = ri * dnx;
                const float v = 2.0f * risti * dnz;
                const float em = 3.0f * s.tj - 4.0f * s.tk + s.tk2;
                a = sq(v) + 9.0f * sq(u);
                b = 6.0f * em * sq(u);
                c = sq(u) * (sq(em) - sq(v) * s2);
                tref = s.tj;
            } else {
                const float u = ri * dnx;
                const float v = risti * dnz;
                const float em = s.tk - s.tj;
                a = sq(u) + sq(v);
                b = -(2.0f * sq(u) * em);
                c = sq(u) * (sq(em) - sq(v) * s2);
                tref = s.tj;
            }
        }
        float rd1 = sq(b) - 4.0f * a * c;
        if (rd1 < 0.0f) rd1 = 0.0f;
        const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
        float trav = tref + tdsh;
        if (third) trav = div3(trav);
        best = (trav < best) ? trav : best;
    }
    return best;
}
// What the device evaluates per lane (round 4): the same candidates, cut for a lane's instruction count -- no branches, and every lane
// does ONE one-sided candidate instead of two: of a node's four one-sided steps (from x-, x+, z-, z+) lane (j, k) takes the x step of its
// side j when j == k and the z step of its side k otherwise -- (0,0): x-, (1,1): x+, (0,1): z+, (1,0): z-, each from the lane's own
// stencil values -- so the four lanes of a node cover all four, and the minimum over the lanes is fouds2's minimum.  The two-sided
// candidate is the one formula with selected operands of eikonal_core.h's solve_node (same table, same roundings; checked there against
// the literal fouds2 on 2e7 neighbourhoods, and here by tests/hostcheck.cpp: hc_quads_compare).
// A = (ri dnx)^2, B = (risti dnz)^2.
DSA_HD float x_quad_lane(const XQuadState& s, bool k_dead, bool j_dead, int j, int k, float slown, const NodeGeom& g)
{
    const float s2 = sq(slown);
    const bool aj = s.ej && s.aj, ak = s.ek && s.ak;
    const bool sj = aj && s.oj && (s.tj > s.tj2);
    const bool sk = ak && s.ok && (s.tk > s.tk2);
    // the lane's one-sided step
    const bool dirx = j == k;
    const bool a1 = dirx ? aj : ak, sw1 = dirx ? sj : sk, dead = dirx ? k_dead : j_dead;
    const float t1 = dirx ? s.tj : s.tk, t12 = dirx ? s.tj2 : s.tk2;
    const float R = dirx ? g.ri : g.risti, d = dirx ? g.dnx : g.dnz;
    const float u2 = 2.0f * R * d;
    const float arg = sw1 ? sq(u2) * s2 : s2 * sq(R) * sq(d);
    const float root1 = sqrt_pos(arg);
    const float one = sw1 ? div3((4.0f * t1 - t12) + root1) : t1 + root1;
    float best = (dead && a1) ? one : kInf;
    // the quadrant's two-sided step
    {
        const float A = sq(g.ri * g.dnx), B = sq(g.risti * g.dnz);
        const float s2A = A * s2, s2B = B * s2;
        const float a00 = A + B, a11 = 4.0f * a00, a10 = 4.0f * A + 9.0f * B, a01 = 4.0f * B + 9.0f * A;
        const float tj = s.tj, tj2 = s.tj2, tk = s.tk, tk2 = s.tk2;
        const float Pj = fmaf(4.0f, tj, -tj2);                                  // 4 tj - tj2 (4 tj is exact)
        const bool both = sj && sk, onesw = sj != sk;
        const float p = sj ? (sk ? Pj : 3.0f * tk) : (sk ? 3.0f * tj : tk);
        const float q_ = (sj && !sk) ? 4.0f * tj : (sk ? 4.0f * tk : tj);
        const float r = sk ? tk2 : (sj ? tj2 : 0.0f);
        const float U = (sj && !sk) ? B : A;
        const float S = (sj && !sk) ? 4.0f * s2A : (sk ? 4.0f * s2B : s2B);
        const float a = sj ? (sk ? a11 : a10) : (sk ? a01 : a00);
        const float tref = sj ? (sk ? Pj : tk) : tj;
        const float em = (p - q_) + r;
        const float b = (both ? 8.0f : (onesw ? 1.0f : -2.0f)) * (((onesw ? 6.0f : 1.0f) * em) * U);
        const float cc = (both ? 4.0f : 1.0f) * (U * (sq(em) - S));
        float rd1 = sq(b) - 4.0f * a * cc;
        if (rd1 < 0.0f) rd1 = 0.0f;
        const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
        float trav = tref + tdsh;
        if (both) trav = div3(trav);
        best = (aj && ak && trav < best) ? trav : best;
    }
    return best;
}
// the trial value of a node from the states of its four quadrants (index 2 j + k); what the sixteen lanes compute, in one place
// for the host (and for the comparison with fouds2 in tests/hostcheck.cpp)
DSA_HD float x_trial_of_quads(const XQuadState* q4, float slown, const NodeGeom& g)
{
    const bool k_dead = (q4[0].ek && !q4[0].ak) || (q4[1].ek && !q4[1].ak);
    const bool j_dead = (q4[0].ej && !q4[0].aj) || (q4[2].ej && !q4[2].aj);
    float best = kInf;
    for (int i = 0; i < 4; ++i) {
        const float c = x_quad_lane(q4[i], k_dead, j_dead, i >> 1, i & 1, slown, g);
        best = (c < best) ? c : best;
    }
    return best;
}
// (the literal per-quadrant form, x_quad_candidates above, stays as the second witness: hc_quads_compare checks both against fouds2)
DSA_HD float x_trial_of_quads_literal(const XQuadState* q4, float slown, const NodeGeom& g)
{
    const bool k_dead = (q4[0].ek && !q4[0].ak) || (q4[1].ek && !q4[1].ak);
    const bool j_dead = (q4[0].ej && !q4[0].aj) || (q4[2].ej && !q4[2].aj);
    float best = kInf;
    for (int i = 0; i < 4; ++i) {
        const float c = x_quad_candidates(q4[i], k_dead, j_dead, slown, g);
        best = (c < best) ? c : best;
    }
    return best;
}

// One accept step of reference travel (CalSurfG.f90:417-485), CPU model in the device's order of events: the stencil records of the four
// neighbours' quadrants are fetched, the root becomes alive and leaves the tree, the neighbours' statuses are read BEHIND the tree moves
// of that removal, and the four neighbours x-, x+, z-, z+ (in that order) get a new trial value and enter the tree / move up in it.  A
// later neighbour whose entry an earlier one's move pushed down a level is no longer at the slot its status said: the slot is checked
// against the tree and the status read again (the device: the sixteenth lane of the group, exact_kernel.hip xg_sift_up).
DSA_HD void x_accept_root(XMarch& m, XEntry root, int iz0, int ix0)
{
    const int iz = iz0 + 1, ix = ix0 + 1;
    int rid[8];
    rec_stencil(m.nbz, root.id, rid);
    const int nzq[4] = { iz, iz, iz - 1, iz + 1 }, nxq[4] = { ix - 1, ix + 1, ix, ix };
    XQuad raws[16];
    for (int l = 0; l < 16; ++l) raws[l] = x_fetch_quad<false>(m, rid[l >> 2], nzq[l >> 2], nxq[l >> 2], (l >> 1) & 1, l & 1);
    m.F[root.id].st = 0;
    x_pop_root(m);
    int nb_in[4], nb_st[4];
    float nb_trial[4];
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = raws[4 * q].in; nb_st[q] = nb_in[q] ? m.F[rid[q]].st : 0; nb_trial[q] = kInf;
        if (nb_in[q] && nb_st[q] != 0) {
            XQuadState s4[4];
            for (int i = 0; i < 4; ++i) { raws[4 * q + i].own.st = nb_st[q]; s4[i] = x_quad_state(raws[4 * q + i], root); }
            const NodeGeom g = { m.ri, raws[4 * q].risti, m.dnx, m.dnz };
            nb_trial[q] = x_trial_of_quads(s4, raws[4 * q].slown, g);
        }
    }
    for (int q = 0; q < 4; ++q) {
        if (!nb_in[q] || nb_st[q] == 0) continue;
        m.F[rid[q]].T = nb_trial[q];                                           // fouds2 overwrites unconditionally (:758)
        if (nb_st[q] < 0) x_add(m, rid[q], nb_trial[q]);
        else {
            int slot = nb_st[q];
            if (xh_get(m, slot).id != rid[q]) slot = m.F[rid[q]].st;
            x_sift_up(m, XEntry{ nb_trial[q], rid[q] }, slot);
        }
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
            m.F[root.id].st = 0;
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
            const float t = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
            const int id = rec_index(m.nbz, s.isz_r - 2 + j, s.isx_r - 2 + i);
            m.F[id].T = t;
            x_add(m, id, t);
        }
}

}  // namespace dsa
