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
//   * per accept step, sixteen lanes own the four quadrants of the four neighbours of the accepted node: they fetch what their
//     quadrant of the stencil needs (seven independent loads per lane, one memory round trip for the step) BEFORE the root is
//     sifted down, and evaluate the stencil afterwards, so the tree work hides the memory latency;
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

constexpr int kXLogCap = 192;               // (node, slot) assignments one accept step can make: five sifts of <= 27 levels (trees below 2^26 entries) = 135
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
// entry into slot s: the tree, the node's status, the step's log.  LDS_ONLY: the caller knows s <= lcap
// store one word of a record whose index every lane holds alike (lane 0 calls it): a vector offset on the field's scalar base -- one VALU
// shift instead of four SALU instructions of 64-bit address arithmetic (the step is bound by the scalar unit: ~1000 scalar against ~500
// vector instructions per accept)
DSA_HD void x_store_word(XMarch& m, int id, int word, int bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned off;
    asm volatile("v_lshlrev_b32 %0, 3, %1" : "=v"(off) : "s"(id));
    typedef __attribute__((address_space(1))) char GChar;
    *(__attribute__((address_space(1))) int*)((GChar*)m.F + off + 4 * word) = bits;
#else
    reinterpret_cast<int*>(&m.F[id])[word] = bits;
#endif
}
DSA_HD int x_float_bits(float f) { union { float f; int i; } u; u.f = f; return u.i; }
// (the log cannot overflow inside a step, see kXLogCap; x_accept_root checks the count once, behind the step)
template <bool LDS_ONLY = false>
DSA_HD void xh_put(XMarch& m, int s, XEntry e)
{
    m.nlog = x_uni(m.nlog);
    if (x_lane() == 0) {
        if (LDS_ONLY || s <= m.lcap) m.hl[s] = e; else m.hg[s - m.lcap - 1] = e;
        x_store_word(m, e.id, 1, s);
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
    // the levels that lie in LDS altogether (children tpc, tpc + 1 <= lcap, hence the parent too): no look at where a slot lives
    const int lim = m.ntr < m.lcap ? m.ntr : m.lcap;
    while (tpc < lim) {
        const XEntry a0 = m.hl[tpc], b0 = m.hl[tpc + 1];              // the two children sit side by side (one 16-byte read)
        XEntry a = XEntry{ x_unif(a0.key), x_uni(a0.id) };
        const XEntry b = XEntry{ x_unif(b0.key), x_uni(b0.id) };
        if (a.key > b.key) { a = b; tpc += 1; }
        if (a.key < e.key) { xh_put<true>(m, tpp, a); tpp = tpc; tpc = 2 * tpp; }
        else { tpc = m.ntr + 1; break; }
    }
    while (tpc < m.ntr) {
        XEntry a, b;
        if (tpc + 1 <= m.lcap) {
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
DSA_HD XQuad x_fetch_quad(const XMarch& m, int id, int nz, int nx, int j, int k)
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
    r.own = m.F[id];
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
// the trial value of a node from the states of its four quadrants (index 2 j + k); what the sixteen lanes compute, in one place
// for the host (and for the comparison with fouds2 in tests/hostcheck.cpp)
DSA_HD float x_trial_of_quads(const XQuadState* q4, float slown, const NodeGeom& g)
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

#if defined(__HIP_DEVICE_COMPILE__)
// exchange inside a group of four lanes: quad_perm [1,0,3,2] (the other k) and [2,3,0,1] (the other j)
DSA_HD int x_dpp_other_k(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }
DSA_HD int x_dpp_other_j(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true); }
DSA_HD float x_dpp_min4(float v)
{
    float o = __int_as_float(x_dpp_other_k(__float_as_int(v)));
    v = (o < v) ? o : v;
    o = __int_as_float(x_dpp_other_j(__float_as_int(v)));
    return (o < v) ? o : v;
}
#endif

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
    const int ql = (lane >> 2) & 3, jl = (lane >> 1) & 1, kl = lane & 1;
    const int mz = ql == 0 ? nzq[0] : ql == 1 ? nzq[1] : ql == 2 ? nzq[2] : nzq[3];
    const int mx = ql == 0 ? nxq[0] : ql == 1 ? nxq[1] : ql == 2 ? nxq[2] : nxq[3];
    const int mid = ql == 0 ? rid[0] : ql == 1 ? rid[1] : ql == 2 ? rid[2] : rid[3];
    XQuad raw;
    raw.in = 0;
    if (lane < 16) raw = x_fetch_quad(m, mid, mz, mx, jl, kl);   // seven loads per lane in flight
#else
    XQuad raws[16];
    for (int l = 0; l < 16; ++l) raws[l] = x_fetch_quad(m, rid[l >> 2], nzq[l >> 2], nxq[l >> 2], (l >> 1) & 1, l & 1);
#endif
    m.nlog = 0;
    if (x_lane() == 0) x_store_word(m, root.id, 1, 0);
    x_pop_root(m);
#if defined(__HIP_DEVICE_COMPILE__)
    float trial = kInf;
    int st = 0, in = 0;
    if (lane < 16) {
        in = raw.in; st = raw.own.st;
        const XQuadState s = x_quad_state(raw, root);
        // the other k of my j, the other j of my k: who is inside the grid and not alive
        const int dk = (s.ek && !s.ak) ? 1 : 0, dj = (s.ej && !s.aj) ? 1 : 0;
        const bool k_dead = dk || x_dpp_other_k(dk), j_dead = dj || x_dpp_other_j(dj);
        const NodeGeom g = { m.ri, raw.risti, m.dnx, m.dnz };
        float c = kInf;
        if (in && st != 0) c = x_quad_candidates(s, k_dead, j_dead, raw.slown, g);
        trial = x_dpp_min4(c);
    }
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = __builtin_amdgcn_readlane(in, 4 * q);
        nb_st[q] = __builtin_amdgcn_readlane(st, 4 * q);
        nb_trial[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(trial), 4 * q));
    }
#else
    for (int q = 0; q < 4; ++q) {
        nb_in[q] = raws[4 * q].in; nb_st[q] = raws[4 * q].own.st; nb_trial[q] = kInf;
        if (nb_in[q] && nb_st[q] != 0) {
            XQuadState s4[4];
            for (int i = 0; i < 4; ++i) s4[i] = x_quad_state(raws[4 * q + i], root);
            const NodeGeom g = { m.ri, raws[4 * q].risti, m.dnx, m.dnz };
            nb_trial[q] = x_trial_of_quads(s4, raws[4 * q].slown, g);
        }
    }
#endif
    for (int q = 0; q < 4; ++q) {
        if (!nb_in[q] || nb_st[q] == 0) continue;
        if (x_lane() == 0) x_store_word(m, rid[q], 0, x_float_bits(nb_trial[q]));       // fouds2 overwrites unconditionally (:758)
        if (nb_st[q] < 0) x_add(m, rid[q], nb_trial[q]);
        else x_sift_up(m, XEntry{ nb_trial[q], rid[q] }, x_current_slot(m, rid[q], nb_st[q]));
    }
    if (m.nlog > kXLogCap - 48) m.error = 2;          // (cannot happen below 2^26 tree entries; the guard keeps the log inside its array)
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
