// TEST INFRASTRUCTURE (not product): round 1's statement of the local solver -- the walk written out step by step around the
// literal stencil `fouds2` -- kept as the reference the product's leaner `solve_node` (eikonal_core.h) is compared with, bit for
// bit, on random neighbourhoods (tests/hostcheck.cpp: hc_solve_node_compare).
#pragma once
#include "../dsurftomo_amd/csrc/eikonal_core.h"

namespace dsa {

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
static inline float solve_node_walk_ref(const Hood& h, float slown, const NodeGeom& g, float* tau_out)
{
    float tn[4], key[4];
    int idx[4] = { 0, 1, 2, 3 };
    unsigned alive = 0u;            // bit q: near neighbour q is alive
    float tnow = -kInf;             // clock of the most recent neighbour acceptance taken into account
    for (int q = 0; q < 4; ++q) {
        const bool in = h.in[q];
        const float raw = in ? h.near_[q] : kInf;
        tn[q] = t_value(raw);
        const bool pin = in && t_pinned(raw);
        const float k = in ? tau_value(h.near_tau[q]) : kInf;
        if (pin) { alive |= 1u << q; tnow = k > tnow ? k : tnow; }
        key[q] = (in && !pin) ? k : kInf;          // +inf: not a candidate of the walk
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

    auto eval = [&](void) -> float {
        Stencil s;
        for (int d = 0; d < 2; ++d) {
            s.tj[d] = tn[d];           s.tk[d] = tn[2 + d];
            s.ej[d] = h.in[d];         s.ek[d] = h.in[2 + d];
            s.aj[d] = (alive >> d) & 1u;
            s.ak[d] = (alive >> (2 + d)) & 1u;
            const float oxr = h.in_outer[d] ? h.outer[d] : kInf;
            const float ozr = h.in_outer[2 + d] ? h.outer[2 + d] : kInf;
            s.tj2[d] = t_value(oxr);   s.tk2[d] = t_value(ozr);
            const float kox = tau_value(h.outer_tau[d]), koz = tau_value(h.outer_tau[2 + d]);
            s.oj[d] = h.in_outer[d] && (kox < tnow || kox == 0.0f);       // 0: alive before any march
            s.ok[d] = h.in_outer[2 + d] && (koz < tnow || koz == 0.0f);
        }
        return fouds2(s, slown, g);
    };

    float c = kInf;
    if (alive) c = eval();
    bool go = true;
#define DSA_STEP(i)                                                                          \
    if (go && key[i] < kInf && c > key[i]) { alive |= 1u << idx[i]; tnow = key[i]; c = eval(); } else go = false
    DSA_STEP(0); DSA_STEP(1); DSA_STEP(2); DSA_STEP(3);
#undef DSA_STEP
    *tau_out = (c > tnow) ? c : tnow;
    return c;
}


}  // namespace dsa
