// Per-node eikonal arithmetic shared by every kernel of the CalSurfG path.
//
// Everything in here is a pure function on values (no global state), written once for device
// code; the same header compiles for the host so that tests can drive the serial/per-node logic
// on a CPU and compare it with the oracle (tests/hostcheck.cpp).  The product never runs it on
// the host.
//
// Arithmetic contract: fp32, one rounding per operation, NO fused multiply-add (build with
// -ffp-contract=off), IEEE divide and sqrt.  Expression order follows the reference's upwind
// stencil `fouds2` (reference CalSurfG.f90:612-758) term by term, because the travel-time field
// has to land on the reference's Fast-Marching answer to <= 1e-4 s and one ulp at T ~ 100 s is
// already 8e-6 s.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DSA_HD __host__ __device__ __forceinline__
#define DSA_HDM __host__ __device__ __forceinline__
#else
#define DSA_HD static inline
#define DSA_HDM inline
#endif

namespace dsa {

constexpr float kInf = __builtin_inff();

// Travel times are non-negative, so the sign bit of a stored T is free: a set sign bit marks a
// node that is "alive from the start" (accepted by the serial prologue, never recomputed).
DSA_HD bool t_pinned(float t) { return __builtin_signbit(t); }
DSA_HD float t_value(float t) { return __builtin_fabsf(t); }

// Field layout in HBM.  A node's state is one 8-byte record (T, tau) and records are stored in
// tiles of 8x8 nodes (512 B = four 128-B lines), tiles in z-fastest order; the slowness field uses
// the same tiling with 4-byte entries.  Why: the update reads T and tau at x-2..x+2 and z-2..z+2.
// With plain z-fastest rows the x neighbours are a whole row (4 KB at N = 1025) apart, so one
// evaluation touches ten separate cache lines for 80 useful bytes, and with one workgroup per CU
// sharing a 4 MB L2 32 ways the measured HBM traffic was ~360x the algorithmic bytes
// (profiles/r01_pmc_k_fim_rowmajor.txt).  In a tile the whole stencil sits in 1-3 lines, and the
// nodes of a front segment crossing the tile share them.
struct Rec { float T, tau; };
constexpr int kTileShift = 3, kTile = 8, kTileRecs = 64;
DSA_HD int tiles_of(int n) { return (n + kTile - 1) >> kTileShift; }
// record index of node (iz0, ix0), 0-based; nbz = tiles_of(nnz)
// inside a tile: eight rows (ix) of eight records (iz), one row = 64 B
DSA_HD int rec_in_tile(int iz0, int ix0) { return ((ix0 & 7) << 3) + (iz0 & 7); }
DSA_HD int rec_ix_in_tile(int r) { return (r >> 3) & 7; }
DSA_HD int rec_iz_in_tile(int r) { return r & 7; }
DSA_HD int rec_index(int nbz, int iz0, int ix0)
{
    return (((ix0 >> kTileShift) * nbz + (iz0 >> kTileShift)) << 6) + rec_in_tile(iz0, ix0);
}
DSA_HD void rec_coords(int nbz, int id, int* iz0, int* ix0)
{
    const int tile = id >> 6, bx = tile / nbz, bz = tile - bx * nbz;
    *ix0 = (bx << kTileShift) + rec_ix_in_tile(id);
    *iz0 = (bz << kTileShift) + rec_iz_in_tile(id);
}

// Record indices of the eight stencil nodes of record `id` (x-, x+, z-, z+; near ones in [0..3], outer ones in
// [4..7]) without going through coordinates: inside a tile a step in x is 8 records and a step in z is 1; leaving the
// tile adds the distance to the neighbouring tile (nbz tiles further in x, one tile further in z).  Indices of
// nodes outside the grid are meaningless (callers test the coordinates before using them).
DSA_HD void rec_stencil(int nbz, int id, int* nid)
{
    const int rx = rec_ix_in_tile(id), rz = rec_iz_in_tile(id);
    const int dx = (nbz << 6) - 64, dz = 56;
    nid[0] = id - 8 - (rx == 0 ? dx : 0);  nid[4] = id - 16 - (rx < 2 ? dx : 0);
    nid[1] = id + 8 + (rx == 7 ? dx : 0);  nid[5] = id + 16 + (rx > 5 ? dx : 0);
    nid[2] = id - 1 - (rz == 0 ? dz : 0);  nid[6] = id - 2 - (rz < 2 ? dz : 0);
    nid[3] = id + 1 + (rz == 7 ? dz : 0);  nid[7] = id + 2 + (rz > 5 ? dz : 0);
}

// ---------------------------------------------------------------------------------------------
// Compact field of the COARSE solve.  tau differs from T at ~0.02 % of the nodes of a field (the ~150 nodes pinned by the
// serial prologue, which carry their accept number, and the non-causal nodes along colliding fronts), so the coarse solve
// stores ONE float per node, tiled like the records: v = +inf unreached; v >= 0: T = tau = v; sign bit set: an exceptional
// node, T = |v|, whose tau (and whether it is pinned) sits in a per-field open-addressing table keyed by the record index.
// Half the bytes, half the cache lines per tile, half the footprint of the thousand fronts a launch keeps in flight.
// Table entry: low word = key (record index, kExcPinned for a pinned node; -1 = empty), high word = tau bits.  A writer
// stores the entry first and the field value second; entries are never removed (a node that turns causal again simply
// stops being looked up), so the capacity bounds the nodes that were ever exceptional: overflow is reported, not ignored.
constexpr int kExcPinned = 0x40000000;
constexpr unsigned long long kExcEmpty = 0x00000000ffffffffull;
DSA_HD unsigned long long exc_pack(int key, float tau)
{
    union { float f; uint32_t u; } a; a.f = tau;
    return ((unsigned long long)a.u << 32) | (uint32_t)key;
}
DSA_HD int exc_key(unsigned long long e) { return (int)(uint32_t)e; }
DSA_HD float exc_tau(unsigned long long e) { union { float f; uint32_t u; } a; a.u = (uint32_t)(e >> 32); return a.f; }
DSA_HD unsigned exc_hash(int id, int log2cap) { return ((uint32_t)id * 2654435761u) >> (32 - log2cap); }
DSA_HD int exc_log2cap_of(int nnx, int nnz) { int l = 10; while ((1 << l) < 4 * (nnx + nnz) + 1024) ++l; return l; }
// serial insert (prologue, one lane per field; host tools): returns false when the table is full
DSA_HD bool exc_insert_serial(unsigned long long* tab, int log2cap, int key, float tau)
{
    const unsigned mask = (1u << log2cap) - 1u;
    unsigned h = exc_hash(key & 0x3fffffff, log2cap);
    for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
        const int k = exc_key(tab[h]);
        if (k == -1 || (k & 0x3fffffff) == (key & 0x3fffffff)) { tab[h] = exc_pack(key, tau); return true; }
    }
    return false;
}
// lookup: tau of node `id`, *pinned set; +inf when absent (cannot happen for a flagged node: the entry is written first)
DSA_HD float exc_find(const unsigned long long* tab, int log2cap, int id, bool* pinned)
{
    const unsigned mask = (1u << log2cap) - 1u;
    unsigned h = exc_hash(id, log2cap);
    for (unsigned n = 0; n <= mask; ++n, h = (h + 1u) & mask) {
        const unsigned long long e = tab[h];
        const int k = exc_key(e);
        if (k == -1) break;
        if ((k & 0x3fffffff) == id) { *pinned = (k & kExcPinned) != 0; return exc_tau(e); }
    }
    *pinned = false;
    return kInf;
}

// Geometry of one node column (depends on ix only); reference CalSurfG.f90:613-615.
struct NodeGeom {
    float ri;     // earth radius
    float risti;  // ri * sin(theta_ix), supplied by a host-computed table (device sinf != libm sinf)
    float dnx;    // node spacing in colatitude (radians)
    float dnz;    // node spacing in longitude (radians)
};

// The 9-point upwind neighbourhood of a node. Index 0 = lower index (ix-1 / iz-1), 1 = higher.
struct Stencil {
    float tj[2], tj2[2];  // x neighbours (iz, ix-+1) and their outer neighbours (iz, ix-+2)
    float tk[2], tk2[2];  // z neighbours (iz-+1, ix) and outer (iz-+2, ix)
    bool ej[2], ek[2];    // neighbour lies inside the grid
    bool aj[2], ak[2];    // neighbour is alive
    bool oj[2], ok[2];    // outer neighbour is alive (and inside the grid)
};

DSA_HD float sq(float x) { return x * x; }

// x / 3.0f, correctly rounded, in three instructions instead of the eleven of an IEEE division: q = RN(x c) with c = RN(1/3), the
// exact residual r = x - 3 q by one FMA, q + r c by another.  Checked against x / 3.0f over ALL finite floats (2^32 patterns:
// bit-identical except x = -0, which gives +0; the arguments below are sums with a positive square-root term).  The FMAs are
// explicit: -ffp-contract=off, the numerical contract of this file, only forbids the compiler to fuse on its own.
// sqrtf for an argument known to be a normal positive number (the one-sided steps below: slowness^2 times grid constants, far
// from the denormal range and from zero): the hardware's v_sqrt_f32 (1 ulp) plus the compiler's own +-1 ulp correction with two
// exact FMA residuals, WITHOUT the range scaling and the zero / infinity class test it wraps around them (9 instructions
// instead of 16).  Correctly rounded, i.e. bit-identical to sqrtf, over the whole range 1e-30 .. 1e30 (tools/micro/exact_math_check.hip
// compares all 1.7e9 patterns on the device).  The host build (CPU checks) uses libm's correctly rounded sqrtf.
DSA_HD float sqrt_pos(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float r1 = fmaf(-sm, s, x);
    float out = (0.0f >= r1) ? sm : s;
    const float r2 = fmaf(-sp, s, x);
    out = (0.0f < r2) ? sp : out;
    return out;
#else
    return sqrtf(x);
#endif
}

DSA_HD float div3(float x)
{
    const float c = 0x1.555556p-2f;
    const float q = x * c;
    const float r = fmaf(-3.0f, q, x);
    return fmaf(r, c, q);
}

// One evaluation of the mixed first/second-order upwind update; returns +inf when no quadrant
// has an alive neighbour.
//
// The reference (CalSurfG.f90:616-757) loops over the four (j, k) quadrants, picks one of eight
// coefficient sets per quadrant, solves a quadratic and keeps the running minimum; a quadrant whose
// j or k neighbour lies outside the grid is skipped entirely (not "not alive").  This is the same
// arithmetic, reorganised so that lanes of a wave do not walk through every variant:
//   * the minimum is order independent, so candidates are enumerated per neighbour;
//   * the one-sided variants have a = 1, b = 0, for which (-b + sqrt(b*b - 4ac)) / (2a) is exactly
//     sqrt(-c) (scaling by 4 and by 1/2 is exact, sqrt is correctly rounded), and dividing by
//     tdiv = 1 is the identity -- so they cost one sqrt and no division, bit for bit;
//   * a one-sided candidate from neighbour j exists iff some k inside the grid is not alive (that
//     is the quadrant the reference would have produced it in), and vice versa;
//   * only quadrants with both neighbours alive need the full quadratic.
// tests/test_hostcheck.py pins this bitwise against the oracle's literal restatement.
DSA_HD float fouds2(const Stencil& s, float slown, const NodeGeom& g)
{
    const float ri = g.ri, risti = g.risti, dnx = g.dnx, dnz = g.dnz;
    const float s2 = sq(slown);
    float best = kInf;
    bool swj[2], swk[2];
    for (int d = 0; d < 2; ++d) {
        swj[d] = s.ej[d] && s.aj[d] && s.oj[d] && (s.tj[d] > s.tj2[d]);
        swk[d] = s.ek[d] && s.ak[d] && s.ok[d] && (s.tk[d] > s.tk2[d]);
    }
    const bool k_dead = (s.ek[0] && !s.ak[0]) || (s.ek[1] && !s.ak[1]);   // some k quadrant without k
    const bool j_dead = (s.ej[0] && !s.aj[0]) || (s.ej[1] && !s.aj[1]);
    const bool any_k = s.ek[0] || s.ek[1], any_j = s.ej[0] || s.ej[1];

    // one-sided candidates
    if (k_dead)
        for (int j = 0; j < 2; ++j) {
            if (!(s.ej[j] && s.aj[j])) continue;
            float trav;
            if (swj[j]) {
                const float u = 2.0f * ri * dnx;
                trav = div3((4.0f * s.tj[j] - s.tj2[j]) + sqrt_pos(sq(u) * s2));
            } else {
                trav = s.tj[j] + sqrt_pos(s2 * sq(ri) * sq(dnx));
            }
            best = (trav < best) ? trav : best;
        }
    if (j_dead)
        for (int k = 0; k < 2; ++k) {
            if (!(s.ek[k] && s.ak[k])) continue;
            float trav;
            if (swk[k]) {
                const float u = 2.0f * risti * dnz;
                trav = div3((4.0f * s.tk[k] - s.tk2[k]) + sqrt_pos(sq(u) * s2));
            } else {
                trav = s.tk[k] + sqrt_pos(s2 * sq(risti) * sq(dnz));
            }
            best = (trav < best) ? trav : best;
        }
    (void)any_k; (void)any_j;

    // two-sided candidates
    for (int j = 0; j < 2; ++j) {
        if (!(s.ej[j] && s.aj[j])) continue;
        for (int k = 0; k < 2; ++k) {
            if (!(s.ek[k] && s.ak[k])) continue;
            float a, b, c, tref;
            bool third = false;
            if (swj[j]) {
                if (swk[k]) {
                    const float u = 2.0f * ri * dnx;
                    const float v = 2.0f * risti * dnz;
                    float em = 4.0f * s.tj[j] - s.tj2[j] - 4.0f * s.tk[k];
                    em = em + s.tk2[k];
                    a = sq(v) + sq(u);
                    b = 2.0f * em * sq(u);
                    c = sq(u) * (sq(em) - s2 * sq(v));
                    tref = 4.0f * s.tj[j] - s.tj2[j];
                    third = true;
                } else {
                    const float u = risti * dnz;
                    const float v = 2.0f * ri * dnx;
                    const float em = 3.0f * s.tk[k] - 4.0f * s.tj[j] + s.tj2[j];
                    a = sq(v) + 9.0f * sq(u);
                    b = 6.0f * em * sq(u);
                    c = sq(u) * (sq(em) - s2 * sq(v));
                    tref = s.tk[k];
                }
            } else {
                if (swk[k]) {
                    const float u = ri * dnx;
                    const float v = 2.0f * risti * dnz;
                    const float em = 3.0f * s.tj[j] - 4.0f * s.tk[k] + s.tk2[k];
                    a = sq(v) + 9.0f * sq(u);
                    b = 6.0f * em * sq(u);
                    c = sq(u) * (sq(em) - sq(v) * s2);
                    tref = s.tj[j];
                } else {
                    const float u = ri * dnx;
                    const float v = risti * dnz;
                    const float em = s.tk[k] - s.tj[j];
                    a = sq(u) + sq(v);
                    b = -(2.0f * sq(u) * em);
                    c = sq(u) * (sq(em) - sq(v) * s2);
                    tref = s.tj[j];
                }
            }
            float rd1 = sq(b) - 4.0f * a * c;
            if (rd1 < 0.0f) rd1 = 0.0f;
            const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
            float trav = tref + tdsh;
            if (third) trav = div3(trav);
            best = (trav < best) ? trav : best;
        }
    }
    return best;
}

// Per-node state of the fixed-point solve: the travel time T and the acceptance time tau.
//
// Fast Marching accepts nodes in the order of a clock that never runs backwards, but the value a
// node is accepted with can be *smaller* than the clock: when a neighbour Y is accepted at time
// t and the update of X from Y gives c < t (two fronts meeting head-on with second-order
// stencils), X is accepted right away, at clock time t, with value c.  Its other neighbours must
// then treat X as "accepted at t", not "accepted at c".  So every node carries
//     T   = its travel time (what the stencil uses),
//     tau = max(T, tau of the last neighbour it used) = when it was accepted (what orders things).
// With T alone the update has no fixed point at such nodes and the iteration cycles forever; with
// (T, tau) the reference's field is a fixed point everywhere except at exact time ties.
//
// Storage: T with sign bit = pinned (accepted by the serial march, never recomputed); tau with
// sign bit = "queued" (the node is in an active list).  +inf = not reached.
constexpr uint32_t kQueuedBit = 0x80000000u;
DSA_HD float tau_value(float t) { return __builtin_fabsf(t); }

// Raw neighbourhood of a node. Order: [0]=x-, [1]=x+, [2]=z-, [3]=z+; `*_outer` two steps away.
struct Hood {
    float near_[4], near_tau[4];
    float outer[4], outer_tau[4];
    bool in[4];       // near neighbour inside the grid
    bool in_outer[4]; // outer neighbour inside the grid
};

// Local solver: the (T, tau) Fast Marching would have accepted at this node, as a pure function
// of the neighbours' states.  FMM recomputes a trial value each time a neighbour is accepted and
// freezes it when the node itself is popped, i.e. when its trial value is no later than the next
// neighbour's acceptance.  So: pinned neighbours are alive from the start; the others are taken in
// order of increasing tau, and the walk stops at the first trial value c with c <= tau(next).
// An outer node counts as alive when it is pinned or was accepted before the neighbour most
// recently added ("now").  Ties stop the walk (c <= tau): the reference's own tie order depends on
// its heap layout and cannot be derived locally ("ties").
// Everything below is indexed with compile-time constants only: runtime-indexed local arrays would
// live in scratch memory, and this function is the inner loop of the solve kernel.
//
// Round 2 form.  The walk evaluates the stencil once per neighbour it takes in, and 64 lanes never agree on which of
// `fouds2`'s variants they are in, so a wave used to execute all of them, five times over (450 VALU instructions per trip, 45 %
// of the kernel's).  Here the stencil is restated for the walk (`fouds2` above stays the literal form; the serial marches use it):
//   * one copy of the evaluation inside a loop over the walk's steps (lanes leave the loop when their walk stops);
//   * what depends on the node only -- the squares of the grid steps, their products with the squared slowness, the four
//     one-sided increments with their square roots -- is computed once, not per evaluation;
//   * one-sided candidates are selects, no branches;
//   * the two-sided candidate of a quadrant is ONE formula whose operands are selected by (second order in x, second order in z):
//         em = (p - q) + r,  b = kb2 ((kb1 em) U),  c = kc (U (em^2 - S)),  t = tref + (-b + sqrt(max(b^2 - (4a) c, 0))) / (2a)
//     with p, q, r, U, S, a, tref and the power-of-two factors from the table in the code; every product and sum is the one
//     `fouds2` rounds (the factors 2, 4, 8 it applies to u, v before squaring are exact and are applied after, 4 t - t2 is one FMA
//     because 4 t is exact), and a lane visits only the quadrants in which both its neighbours are alive (one, as a rule).
// tests/test_hostcheck.py compares this with the step-by-step form around `fouds2` (tests/solve_node_walk_ref.h) on 2e7 random
// neighbourhoods, bit for bit, and the fields it produces with the oracle's.
//
// Round 3.  (1) The first step of a walk that starts without pinned neighbours takes in ONE neighbour: no quadrant has both
// sides alive, and of the four one-sided candidates only that neighbour's exists -- it is evaluated by its own formula in front of
// the loop (the same operations on the same operands, so the same bits), and the loop runs one trip less for the whole wave.
// (2) TIE = true (the tie detector of the engine option `exact_ties`): when the walk stops at an exact tie, c == tau(next) bit for
// bit, Fast Marching would accept one of the two nodes first and re-evaluate the other against it, and which one is decided by the
// layout of the reference's heap.  The detector takes the tied neighbour in for one more trip of the same body and
// reports |c' - c| in *tie_out (>= 0; -1 when the walk did not end on a tie); the returned (T, tau) are those of the walk that
// stopped at the tie, as before.
// (DSA_LEDGER builds of fim_kernel.hip count the trips of the solver's parts: tools/isa_ledger.py)
#ifndef DSA_LEDGER_PARAM
#define DSA_LEDGER_PARAM
#define DSA_LEDGER_PASS
#define DSA_LEDGER_COUNT(k, name) do { } while (0)
#endif
// AMBONLY (with TIE = false; round 6): the plain walk that only NOTES a raised-key ambiguity (below) -- *tie_out = 0 when it met one, -1 otherwise --
// without the detector's probes: what the bundle kernel's slow pass marks its census candidates by (the full detector there cost 2.3 % of the kernel).
template <bool TIE, bool AMBONLY = false>
DSA_HD float solve_node_t(const Hood& h, float slown, const NodeGeom& g, float* tau_out, float* tie_out DSA_LEDGER_PARAM)
{
    constexpr bool KEYS = TIE || AMBONLY;          // (the running minimum of the node's keys is kept)
    DSA_LEDGER_COUNT(10, "solve_prologue");
    float tn[4], t2[4], ko[4], key[4];
    int idx[4] = { 0, 1, 2, 3 };
    unsigned alive = 0u, inside = 0u;   // bit q: near neighbour q is alive / inside the grid
    float tnow = -kInf;                 // clock of the most recent neighbour acceptance taken into account
    for (int q = 0; q < 4; ++q) {
        const bool in = h.in[q];
        const float raw = in ? h.near_[q] : kInf;
        tn[q] = t_value(raw);
        const bool pin = in && t_pinned(raw);
        const float k = in ? tau_value(h.near_tau[q]) : kInf;
        if (in) inside |= 1u << q;
        if (pin) { alive |= 1u << q; tnow = k > tnow ? k : tnow; }
        key[q] = (in && !pin) ? k : kInf;          // +inf: not a candidate of the walk
        t2[q] = h.in_outer[q] ? t_value(h.outer[q]) : kInf;
        ko[q] = h.in_outer[q] ? tau_value(h.outer_tau[q]) : kInf;
    }
    // sort the candidates by (acceptance time, index): 5-comparator network, same order as a stable sort
#define DSA_CE(a, b)                                                                         \
    do {                                                                                     \
        const bool sw = key[a] > key[b] || (key[a] == key[b] && idx[a] > idx[b]);            \
        const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];                    \
        const int ia = sw ? idx[b] : idx[a], ib = sw ? idx[a] : idx[b];                      \
        key[a] = ka; key[b] = kb; idx[a] = ia; idx[b] = ib;                                  \
    } while (0)
    DSA_CE(0, 1); DSA_CE(2, 3); DSA_CE(0, 2); DSA_CE(1, 3); DSA_CE(1, 2);
#undef DSA_CE

    // node constants (x: colatitude direction, neighbours 0/1; z: longitude direction, neighbours 2/3)
    const float s2 = sq(slown);
    const float A = sq(g.ri * g.dnx), B = sq(g.risti * g.dnz);          // u^2 of the first-order forms; second order: 4A, 4B
    const float s2A = A * s2, s2B = B * s2;
    const float c1x = sqrt_pos(s2 * sq(g.ri) * sq(g.dnx)), c1z = sqrt_pos(s2 * sq(g.risti) * sq(g.dnz));   // first-order one-sided steps
    const float c2x = sqrt_pos(4.0f * s2A), c2z = sqrt_pos(4.0f * s2B);                                   // second-order ones (before the / 3)
    const float a00 = A + B, a11 = 4.0f * a00;                          // a of the quadratic: neither / both sides second order
    const float a10 = 4.0f * A + 9.0f * B, a01 = 4.0f * B + 9.0f * A;   // x only / z only

    float c = kInf;
    bool first = alive != 0u;           // pinned neighbours are alive from the start: one evaluation before the walk
#ifndef DSA_SOLVE_NO_FIRST_STEP         // (A/B switch of tools/ab_build.sh: without the block the loop below does the same step with its whole body)
    DSA_LEDGER_COUNT(11, "solve_consts");
    if (!first && key[0] < kInf) {      // no pinned neighbour: the walk's first neighbour alone (c = +inf > its key)
        DSA_LEDGER_COUNT(12, "walk_first");
        const int a = idx[0];
        const bool x = a < 2;
        const float tna = a == 0 ? tn[0] : a == 1 ? tn[1] : a == 2 ? tn[2] : tn[3];
        const float t2a = a == 0 ? t2[0] : a == 1 ? t2[1] : a == 2 ? t2[2] : t2[3];
        const float koa = a == 0 ? ko[0] : a == 1 ? ko[1] : a == 2 ? ko[2] : ko[3];
        alive = 1u << a;
        tnow = key[0];
        key[0] = key[1]; key[1] = key[2]; key[2] = key[3]; key[3] = kInf;
        idx[0] = idx[1]; idx[1] = idx[2]; idx[2] = idx[3];
        const bool o = koa < tnow || koa == 0.0f;
        const bool sw1 = o && tna > t2a;
        const bool have = x ? ((inside >> 2) & 3u) != 0u : (inside & 3u) != 0u;    // a neighbour of the other direction exists (none is alive)
        const float one = sw1 ? div3(fmaf(4.0f, tna, -t2a) + (x ? c2x : c2z)) : tna + (x ? c1x : c1z);
        c = (have && one < kInf) ? one : kInf;
    }
#endif
    bool probing = false;               // TIE: the trip after a tie
    // (round 6) TIE, the second kind of tie: an OUTER node accepted at the very clock of the neighbour taken in last (ko == tnow: only an
    // exceptional outer node -- accepted later than its value -- can be, a regular one lies below its near node's time).  Whether it was alive
    // when the reference evaluated this node for the last time -- second order or first in that direction -- is the tree's choice again; the
    // walk here counts it as not alive (ko < tnow).  The probe evaluates the stopped walk's last trip once more with such outer nodes alive.
    bool oprobe = false, odone = false;
    float c_keep = 0.0f, tnow_keep = 0.0f;
    // (round 6) TIE, the third kind: a RAISED KEY.  The reference's updtree moves an entry towards the root only (CalSurfG.f90:899-920): when an update makes a
    // trial value LARGER (second-order stencils switching on) the entry stays where its smaller key put it, the tree is no heap there any more, and the node
    // may be accepted before its key's turn -- before a neighbour whose acceptance time lies between the smallest key the node ever had (cmin) and its
    // current one.  The walk here takes that neighbour in (it orders by acceptance times); whether the reference did is its tree's layout again.  The first
    // such neighbour is noted with the value the node would have been accepted with (c_amb); the influence is what the rest of the walk changes.
    float cmin = kInf, c_amb = 0.0f;
    bool amb = false;
    if (KEYS) { *tie_out = -1.0f; cmin = c; }
    for (;;) {
        if (!first) {
            const float nk = key[0];
            bool take = true;
            if (!(nk < kInf && c > nk)) {
                bool otie = false;
                if (TIE && !odone && c < kInf) {
                    for (int q = 0; q < 4; ++q) otie = otie || (((alive >> q) & 1u) && ko[q] == tnow && ko[q] != 0.0f && tn[q] > t2[q]);
                }
                if (otie) { oprobe = true; odone = true; c_keep = c; take = false; }
                else {
                    if (!(TIE && nk < kInf && c == nk)) break;
                    probing = true; c_keep = c; tnow_keep = tnow;     // an exact tie: one more trip with the tied neighbour alive
                }
            }
            if (take) {
                if (KEYS && !amb && !probing && cmin < nk) { amb = true; c_amb = c; }
                alive |= 1u << idx[0];
                tnow = nk;
                key[0] = key[1]; key[1] = key[2]; key[2] = key[3]; key[3] = kInf;
                idx[0] = idx[1]; idx[1] = idx[2]; idx[2] = idx[3];
            }
        }
        first = false;
        DSA_LEDGER_COUNT(13, "walk_body");

        bool sw[4];
        float P[4];
        for (int q = 0; q < 4; ++q) {
            const bool o = ko[q] < tnow || ko[q] == 0.0f || (TIE && oprobe && ko[q] == tnow);      // outer node alive (0: alive before any march)
            sw[q] = ((alive >> q) & 1u) && o && tn[q] > t2[q];
            P[q] = fmaf(4.0f, tn[q], -t2[q]);                            // 4 t - t2 (4 t is exact)
        }
        const unsigned aj = alive & 3u, ak = (alive >> 2) & 3u;
        const bool k_dead = (((inside >> 2) & 3u) & ~ak) != 0u;          // some z neighbour inside the grid is not alive
        const bool j_dead = ((inside & 3u) & ~aj) != 0u;
        float best = kInf;
        for (int q = 0; q < 4; ++q) {
            const bool x = q < 2;
            const float one = sw[q] ? div3(P[q] + (x ? c2x : c2z)) : tn[q] + (x ? c1x : c1z);
            const bool have = ((alive >> q) & 1u) && (x ? k_dead : j_dead);
            best = (have && one < best) ? one : best;
        }
        unsigned pm = ((aj & 1u) ? ak : 0u) | ((aj & 2u) ? (ak << 2) : 0u);   // bit 2 j + k: quadrant with both neighbours alive
        while (pm) {
            DSA_LEDGER_COUNT(14, "walk_quadrant");
            const bool j1 = (pm & 3u) == 0u;                             // lowest set bit: quadrant (j, k)
            const unsigned pj = j1 ? (pm >> 2) : pm;
            const bool k1 = (pj & 1u) == 0u;
            pm &= pm - 1u;
            const float tj = j1 ? tn[1] : tn[0], tj2 = j1 ? t2[1] : t2[0], Pj = j1 ? P[1] : P[0];
            const float tk = k1 ? tn[3] : tn[2], tk2 = k1 ? t2[3] : t2[2];
            const bool sj = j1 ? sw[1] : sw[0], sk = k1 ? sw[3] : sw[2];
            //            sj & sk            sj only            sk only            neither
            //  p         4 tj - tj2         3 tk               3 tj               tk
            //  q         4 tk               4 tj               4 tk               tj
            //  r         tk2                tj2                tk2                0
            //  U         A  (kb2 8, kc 4)   B                  A                  A  (kb2 -2)
            //  S         4 s2 B             4 s2 A             4 s2 B             s2 B
            //  a         4 (A + B)          4 A + 9 B          4 B + 9 A          A + B
            //  tref      4 tj - tj2         tk                 tj                 tj          (sj & sk: the sum is divided by 3)
            const bool both = sj && sk, one = sj != sk;
            const float p = sj ? (sk ? Pj : 3.0f * tk) : (sk ? 3.0f * tj : tk);
            const float q_ = (sj && !sk) ? 4.0f * tj : (sk ? 4.0f * tk : tj);
            const float r = sk ? tk2 : (sj ? tj2 : 0.0f);
            const float U = (sj && !sk) ? B : A;
            const float S = (sj && !sk) ? 4.0f * s2A : (sk ? 4.0f * s2B : s2B);
            const float a = sj ? (sk ? a11 : a10) : (sk ? a01 : a00);
            const float tref = sj ? (sk ? Pj : tk) : tj;
            const float em = (p - q_) + r;
            const float b = (both ? 8.0f : (one ? 1.0f : -2.0f)) * (((one ? 6.0f : 1.0f) * em) * U);
            const float cc = (both ? 4.0f : 1.0f) * (U * (sq(em) - S));
            float rd1 = sq(b) - 4.0f * a * cc;
            if (rd1 < 0.0f) rd1 = 0.0f;
            const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
            float trav = tref + tdsh;
            if (both) trav = div3(trav);
            best = (trav < best) ? trav : best;
        }
        c = best;
        if (TIE && oprobe) { *tie_out = fabsf(c - c_keep); c = c_keep; oprobe = false; continue; }      // (back to the top: the walk stops again, now for good or on a tie of the first kind)
        if (TIE && probing) { const float ti = fabsf(c - c_keep); *tie_out = ti > *tie_out ? ti : *tie_out; c = c_keep; tnow = tnow_keep; break; }
        if (KEYS) cmin = c < cmin ? c : cmin;
    }
    if (TIE && amb) { const float ti = fabsf(c - c_amb); *tie_out = ti > *tie_out ? ti : *tie_out; }
    if (AMBONLY && amb) *tie_out = 0.0f;
    DSA_LEDGER_COUNT(15, "solve_epilogue");
    *tau_out = (c > tnow) ? c : tnow;
    return c;
}
// Round 4: the walk of solve_node_t written out for the REGULAR neighbourhood -- all four near neighbours inside the grid, no pinned
// value among the eight, every acceptance time equal to its value (tau = T: the compact field's rule for every node that has no entry
// in the exception table).  tn[q] / t2[q]: the near and outer values in Hood's order (+inf: not reached; outer outside the grid: +inf).
//
// What the general form spends on generality -- a sorting network over (key, index), the alive set as a bit mask, a loop whose trip
// count differs from lane to lane with every candidate behind a select -- collapses here:
//   * the walk takes the neighbours in the order of their values, the lower index first on a tie, so its first two are the smaller
//     one of each direction ("upwind") unless the second-smallest is the first one's opposite neighbour;
//   * with tau = T the second-order switch of a neighbour, (outer alive at the clock) && tn > t2, is just t2 < tn, at every step of the
//     walk: the one-sided candidate of the first neighbour is the same number at step one and step two;
//   * at step two exactly one quadrant has both neighbours alive, and both one-sided candidates exist (the opposite neighbours are
//     inside the grid and not alive).
// Step one: c1 = one-sided(first); the walk stops there when the second key is +inf or c1 <= it.  Step two: c2 = min(one-sided x,
// one-sided z, the quadrant's two-sided value); it stops when the third key is +inf or c2 <= it.  Anything else -- a third neighbour
// taken in, the opposite neighbour second -- returns *ok = false and the caller uses solve_node_t.  Every operation is the one
// solve_node_t performs on the same operands (tests/test_hostcheck.py: 2e7 random regular neighbourhoods and whole solves, bit for bit).
// `tie` (optional): the walk stopped at an exact tie -- its value equal, bit for bit, to the next neighbour's time: what solve_node_t<true>
// takes one more trip for (the engine's tie detector); the caller then evaluates the node with that form.
DSA_HD float solve_regular(const float* tn, const float* t2, float slown, const NodeGeom& g, float* tau_out, bool* ok, bool* tie = nullptr)
{
    const bool x1 = tn[1] < tn[0], z1 = tn[3] < tn[2];
    const float tx = x1 ? tn[1] : tn[0], ox = x1 ? tn[0] : tn[1], tx2 = x1 ? t2[1] : t2[0];
    const float tz = z1 ? tn[3] : tn[2], oz = z1 ? tn[2] : tn[3], tz2 = z1 ? t2[3] : t2[2];
    const bool xf = tx <= tz;                                   // the x neighbour leads (lower index on a tie)
    const bool std2 = xf ? (tz < ox) : (tx <= oz);              // the second of the walk is the other direction's upwind neighbour
    const float k0 = xf ? tx : tz;
    const float k1 = xf ? (std2 ? tz : ox) : (std2 ? tx : oz);
    const float k2 = ox < oz ? ox : oz;
    const float s2 = sq(slown);
    const float A = sq(g.ri * g.dnx), B = sq(g.risti * g.dnz);
    const float s2A = A * s2, s2B = B * s2;
    const bool sx = tx2 < tx, sz = tz2 < tz;                    // second order towards x / z
    const float Px = fmaf(4.0f, tx, -tx2), Pz = fmaf(4.0f, tz, -tz2);
    const float rx = sqrt_pos(sx ? 4.0f * s2A : s2 * sq(g.ri) * sq(g.dnx));
    const float rz = sqrt_pos(sz ? 4.0f * s2B : s2 * sq(g.risti) * sq(g.dnz));
    const float onex = sx ? div3(Px + rx) : tx + rx;
    const float onez = sz ? div3(Pz + rz) : tz + rz;
    const float c1 = xf ? onex : onez;
    const bool stop1 = !(k1 < kInf && c1 > k1);
    // step two: the quadrant (x upwind, z upwind); operand table of solve_node_t
    const bool both = sx && sz, one = sx != sz;
    const float p = sx ? (sz ? Px : 3.0f * tz) : (sz ? 3.0f * tx : tz);
    const float q_ = (sx && !sz) ? 4.0f * tx : (sz ? 4.0f * tz : tx);
    const float r = sz ? tz2 : (sx ? tx2 : 0.0f);
    const float U = (sx && !sz) ? B : A;
    const float S = (sx && !sz) ? 4.0f * s2A : (sz ? 4.0f * s2B : s2B);
    const float a00 = A + B;
    const float a = sx ? (sz ? 4.0f * a00 : 4.0f * A + 9.0f * B) : (sz ? 4.0f * B + 9.0f * A : a00);
    const float tref = sx ? (sz ? Px : tz) : tx;
    const float em = (p - q_) + r;
    const float b = (both ? 8.0f : (one ? 1.0f : -2.0f)) * (((one ? 6.0f : 1.0f) * em) * U);
    const float cc = (both ? 4.0f : 1.0f) * (U * (sq(em) - S));
    float rd1 = sq(b) - 4.0f * a * cc;
    if (rd1 < 0.0f) rd1 = 0.0f;
    const float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
    float trav = tref + tdsh;
    if (both) trav = div3(trav);
    float c2 = onex < kInf ? onex : kInf;
    c2 = onez < c2 ? onez : c2;
    c2 = trav < c2 ? trav : c2;
    const bool stop2 = !(k2 < kInf && c2 > k2);
    // (round 5, bundle_kernel.hip: an outer value the caller did not fetch arrives as NaN; the walk never reads the downwind ones, and a
    // member whose UPWIND outer value is missing -- its near neighbour reached -- is not this function's business)
    const bool missing = (tx2 != tx2 && tx < kInf) || (tz2 != tz2 && tz < kInf);
    *ok = (stop1 || (std2 && stop2)) && !missing;
    if (tie) *tie = stop1 ? (k1 < kInf && c1 == k1) : (std2 && stop2 && k2 < kInf && c2 == k2);
    const float c = stop1 ? c1 : c2, tnow = stop1 ? k0 : k1;
    *tau_out = (c > tnow) ? c : tnow;
    return c;
}

DSA_HD float solve_node(const Hood& h, float slown, const NodeGeom& g, float* tau_out)
{
#ifdef DSA_LEDGER
    unsigned dsa_lc[24] = {};       // (callers outside the ledger's kernel: counted into nothing)
#endif
    return solve_node_t<false>(h, slown, g, tau_out, nullptr DSA_LEDGER_PASS);
}

// ---------------------------------------------------------------------------------------------
// Serial narrow-band march used only for the few dozen accept steps around a source where the
// reference's behaviour depends on its heap (source-cell start-up and the refined->coarse band;
// reference CalSurfG.f90:288-487 with the tree of :768-921, whose update step only ever moves an
// entry towards the root).  One lane runs it per source; all state lives in caller-provided
// arrays.  `status`: -1 far, 0 alive, >0 slot in the tree.
// ---------------------------------------------------------------------------------------------
struct MarchView {
    Rec* F;               // (T, tau) records: tiled over the full grid, or -- `window` set -- only the status window, (wnz, wnx) column-major
    int window;           // 1: F covers the status window only (coarse band march: the coarse field itself is the compact one)
    const float* slow;    // tiled slowness
    int nbz;              // tiles per column of the full grid
    const float* risti;   // per-ix table, 1-based index ix -> risti[ix-1]
    int16_t* status;      // window-local status, (wnz, wnx) column-major
    int wz0, wx0;         // full-grid index of window element (1,1) minus 1
    int wnz, wnx;         // window extent
    int nnz, nnx;         // full grid extent
    float ri, dnx, dnz;
    int32_t* heap;        // packed (iz << 16 | ix), 1-based slots, capacity `cap`
    int cap;
    int ntr;
    int error;            // 1: window overflow, 2: tree overflow
    float clock;          // number of accepts so far (see mv_accept_root)
};

DSA_HD Rec& mv_rec(MarchView& m, int iz, int ix)
{
    return m.window ? m.F[(size_t)(ix - 1 - m.wx0) * (size_t)m.wnz + (size_t)(iz - 1 - m.wz0)] : m.F[rec_index(m.nbz, iz - 1, ix - 1)];
}
DSA_HD float& mv_T(MarchView& m, int iz, int ix) { return mv_rec(m, iz, ix).T; }
DSA_HD float mv_slow(const MarchView& m, int iz, int ix) { return m.slow[rec_index(m.nbz, iz - 1, ix - 1)]; }
DSA_HD bool mv_inwin(const MarchView& m, int iz, int ix)
{
    return iz > m.wz0 && iz <= m.wz0 + m.wnz && ix > m.wx0 && ix <= m.wx0 + m.wnx;
}
// status of a node; nodes outside the window but inside the grid read as far (and raise the
// overflow error if the march ever needs to write them)
DSA_HD int mv_get(const MarchView& m, int iz, int ix)
{
    if (!mv_inwin(m, iz, ix)) return -1;
    return m.status[(size_t)(ix - 1 - m.wx0) * (size_t)m.wnz + (size_t)(iz - 1 - m.wz0)];
}
DSA_HD void mv_set(MarchView& m, int iz, int ix, int v)
{
    if (!mv_inwin(m, iz, ix)) { m.error = 1; return; }
    m.status[(size_t)(ix - 1 - m.wx0) * (size_t)m.wnz + (size_t)(iz - 1 - m.wz0)] = (int16_t)v;
}
DSA_HD int hp_iz(int32_t p) { return p >> 16; }
DSA_HD int hp_ix(int32_t p) { return p & 0xffff; }
DSA_HD float mv_key(MarchView& m, int slot) { return mv_T(m, hp_iz(m.heap[slot]), hp_ix(m.heap[slot])); }

// (Round 5: the tree's entries are node coordinates and the keys sit in the field, so a comparison is two dependent loads; the entry that
// moves -- the node sifting up, the last entry sinking from the root -- keeps its key in a register, the two children of a level are fetched
// together, and mv_trial fetches its eight stencil nodes together: the serial marches of k_refined_startup / k_coarse_march wait for a
// third of the memory round trips they used to.  Same comparisons on the same values, same stores.)
DSA_HD void mv_sift_up(MarchView& m, int iz, int ix, int tpc)
{
    const float key = mv_T(m, iz, ix);
    int tpp = tpc / 2;
    while (tpp > 0) {
        const int32_t pe = m.heap[tpp];
        if (key < mv_T(m, hp_iz(pe), hp_ix(pe))) {
            mv_set(m, iz, ix, tpp);
            mv_set(m, hp_iz(pe), hp_ix(pe), tpc);
            m.heap[tpp] = m.heap[tpc]; m.heap[tpc] = pe;
            tpc = tpp;
            tpp = tpc / 2;
        } else tpp = 0;
    }
}
DSA_HD void mv_add(MarchView& m, int iz, int ix)
{
    if (m.ntr + 1 >= m.cap) { m.error = 2; return; }
    m.ntr += 1;
    mv_set(m, iz, ix, m.ntr);
    m.heap[m.ntr] = (iz << 16) | ix;
    mv_sift_up(m, iz, ix, m.ntr);
}
DSA_HD void mv_pop_root(MarchView& m)
{
    if (m.ntr == 1) { m.ntr = 0; return; }
    const int32_t se = m.heap[m.ntr];                     // the entry that sinks from the root
    const int siz = hp_iz(se), six = hp_ix(se);
    const float skey = mv_T(m, siz, six);
    mv_set(m, siz, six, 1);
    m.heap[1] = se;
    m.ntr -= 1;
    int tpp = 1, tpc = 2;
    while (tpc < m.ntr) {
        const int32_t c0 = m.heap[tpc], c1 = m.heap[tpc + 1];
        const float k0 = mv_T(m, hp_iz(c0), hp_ix(c0)), k1 = mv_T(m, hp_iz(c1), hp_ix(c1));
        const bool right = k0 > k1;
        const int32_t ce = right ? c1 : c0;
        const float ck = right ? k1 : k0;
        if (right) tpc += 1;
        if (ck < skey) {
            mv_set(m, siz, six, tpc);
            mv_set(m, hp_iz(ce), hp_ix(ce), tpp);
            m.heap[tpc] = se; m.heap[tpp] = ce;
            tpp = tpc;
            tpc = 2 * tpp;
        } else tpc = m.ntr + 1;
    }
    if (tpc == m.ntr) {
        const int32_t ce = m.heap[tpc];
        if (mv_T(m, hp_iz(ce), hp_ix(ce)) < skey) {
            mv_set(m, siz, six, tpc);
            mv_set(m, hp_iz(ce), hp_ix(ce), tpp);
            m.heap[tpc] = se; m.heap[tpp] = ce;
        }
    }
}

// trial value at (iz, ix) from the march's own alive set (status == 0).  The eight stencil nodes' statuses and values are fetched together:
// a node outside the grid or the window reads (iz, ix) itself -- which lies in both -- and is dropped.
DSA_HD float mv_trial(MarchView& m, int iz, int ix)
{
    Stencil s;
    const int jx[2] = { ix - 1, ix + 1 }, jx2[2] = { ix - 2, ix + 2 };
    const int kz[2] = { iz - 1, iz + 1 }, kz2[2] = { iz - 2, iz + 2 };
    // 0, 1: (iz, jx[d]); 2, 3: (iz, jx2[d]); 4, 5: (kz[d], ix); 6, 7: (kz2[d], ix)
    int st[8];
    float tv[8];
    bool in[8];
    for (int q = 0; q < 8; ++q) {
        const int d = q & 1;
        const int cz = q < 4 ? iz : (q < 6 ? kz[d] : kz2[d]), cx = q < 2 ? jx[d] : (q < 4 ? jx2[d] : ix);
        in[q] = cx >= 1 && cx <= m.nnx && cz >= 1 && cz <= m.nnz && mv_inwin(m, cz, cx);
        const int uz = in[q] ? cz : iz, ux = in[q] ? cx : ix;
        st[q] = m.status[(size_t)(ux - 1 - m.wx0) * (size_t)m.wnz + (size_t)(uz - 1 - m.wz0)];
        tv[q] = mv_T(m, uz, ux);
    }
    for (int d = 0; d < 2; ++d) {
        s.ej[d] = jx[d] >= 1 && jx[d] <= m.nnx;
        s.aj[d] = in[d] && st[d] == 0;
        s.tj[d] = s.aj[d] ? tv[d] : kInf;
        const bool o = in[2 + d] && st[2 + d] == 0;
        s.oj[d] = o;
        s.tj2[d] = o ? tv[2 + d] : kInf;
        s.ek[d] = kz[d] >= 1 && kz[d] <= m.nnz;
        s.ak[d] = in[4 + d] && st[4 + d] == 0;
        s.tk[d] = s.ak[d] ? tv[4 + d] : kInf;
        const bool p = in[6 + d] && st[6 + d] == 0;
        s.ok[d] = p;
        s.tk2[d] = p ? tv[6 + d] : kInf;
    }
    NodeGeom g = { m.ri, m.risti[ix - 1], m.dnx, m.dnz };
    return fouds2(s, mv_slow(m, iz, ix), g);
}

// accept the root and update its four neighbours; returns false on error
// The tree of the reference is not always a valid heap here (start values in the source cell, injected
// refined values, keys raised by an update that only sifts up), so nodes can be accepted out of order: a node
// with a SMALLER time after one with a larger time.  What matters downstream is the order, not the value:
// a node two steps away only enters a second-order stencil if it was accepted before the in-between node was
// (the reference never re-evaluates a node when a node two steps away is accepted).  So every accept of a
// serial march records its sequence number in tau, as a value far below any travel time (all marched nodes
// are accepted before every node of the fixed-point solve); 0 = alive before the march started.
constexpr float kSeqClock = 1.0e-30f;

DSA_HD bool mv_accept_root(MarchView& m)
{
    const int ix = hp_ix(m.heap[1]), iz = hp_iz(m.heap[1]);
    m.clock += 1.0f;
    mv_rec(m, iz, ix).tau = m.clock * kSeqClock;
    mv_set(m, iz, ix, 0);
    mv_pop_root(m);
    for (int i = ix - 1; i <= ix + 1; i += 2) {
        if (i < 1 || i > m.nnx) continue;
        const int st = mv_get(m, iz, i);
        if (st == -1) {
            if (!mv_inwin(m, iz, i)) { m.error = 1; return false; }
            mv_T(m, iz, i) = mv_trial(m, iz, i); mv_add(m, iz, i);
        } else if (st > 0) { mv_T(m, iz, i) = mv_trial(m, iz, i); mv_sift_up(m, iz, i, st); }
    }
    for (int i = iz - 1; i <= iz + 1; i += 2) {
        if (i < 1 || i > m.nnz) continue;
        const int st = mv_get(m, i, ix);
        if (st == -1) {
            if (!mv_inwin(m, i, ix)) { m.error = 1; return false; }
            mv_T(m, i, ix) = mv_trial(m, i, ix); mv_add(m, i, ix);
        } else if (st > 0) { mv_T(m, i, ix) = mv_trial(m, i, ix); mv_sift_up(m, i, ix, st); }
    }
    return m.error == 0;
}

// bilinear weight sum of reference CalSurfG.f90:2328-2349; nv[i][j]: i = x offset, j = z offset
DSA_HD float bilinear4(const float nv[2][2], float dnx, float dnz, float dsx, float dsz)
{
    float biv = 0.0f;
    for (int i = 1; i <= 2; ++i)
        for (int j = 1; j <= 2; ++j) {
            const float produ = (1.0f - fabsf(((float)(i - 1) * dnx - dsx) / dnx)) *
                                (1.0f - fabsf(((float)(j - 1) * dnz - dsz) / dnz));
            biv = biv + nv[i - 1][j - 1] * produ;
        }
    return biv;
}

// cubic B-spline basis, reference CalSurfG.f90:1509-1512
DSA_HD void bspline4(float u, float w[4])
{
    const float u2 = u * u, u3 = u * (u * u);
    const float m = 1.0f - u;
    w[0] = (m * (m * m)) / 6.0f;
    w[1] = (4.0f - 6.0f * u2 + 3.0f * u3) / 6.0f;
    w[2] = (1.0f + 3.0f * u + 3.0f * u2 - 3.0f * u3) / 6.0f;
    w[3] = u3 / 6.0f;
}

}  // namespace dsa
