// BUNDLE LABORATORY (test infrastructure / probe, not product): the device schedule of csrc/fim_kernel.hip (tests/tools/sched_lab.cpp
// mode 6: parity halves, odd-round rule) replayed on the CPU for G coarse problems of ONE source at once -- the G periods of a source,
// whose fronts cross the same nodes at about the same round -- under ONE shared active set, to count what sharing costs before it is
// built on the GPU.  Every member keeps its own field, slowness, window and arithmetic (the product's solve_node); only the schedule is
// shared, and the fixed point is schedule independent, so every member must come out bit-identical to its solo run (checked by the caller).
// Rules (`rule`):
//   0  solo: each member by itself (the reference point: evaluations and rounds of G separate solves)
//   1  ANY: a listed node is ready when ANY member's lower bound lies inside that member's window; all members are evaluated
//   2  PILOT: routed by member `pilot` alone (its lower bound, its window); all members are evaluated
//   4  PILOT + hold-back: like 2, and a change of ANY member at a node holds the pilot's window back by the pilot's acceptance time there
//      (what the pilot's own changes do): the window waits for the member that settles last, and a cycle in any member is seen inside it
//   5  PILOT + hold-back for SMALL changes only (the new value within a few ulps of the old one: a cycle among ulp-tied nodes, or a last
//      refinement): cycles of any member then stall the window as they do in a solo run, everything else runs as under rule 2
//   6  PILOT, and a node waits for its SECOND earliest neighbour to be inside the window (or for the earliest to lie `pilot` per cent of a window
//      back: the parameter is passed in the pilot argument, the pilot is member 0): fewer evaluations, more rounds (sched_lab mode 9)
//   3  EXACT: a pending bit per (node, member): a member is evaluated at a node exactly when its solo schedule would; the lanes of the
//      members that are not due idle (reported as member fill)
// build: g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -msse2 -mfpmath=sse -o tests/tools/libbundle_lab.so tests/tools/bundle_lab.cpp
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../dsurftomo_amd/csrc/eikonal_core.h"

using namespace dsa;

namespace {
struct Member { std::vector<Rec> F; std::vector<float> slow; float window, theta, tmin, best_tmin, freeze; unsigned hist[4], hsh; int stall; long evals, changes, freezes; };
}

// T, tau, slow: G row-major (ix * nnz + iz) fields one after the other; out: [0] rounds [1] node evaluations (each covers G members, rule 3: the members due)
// [2] member evaluations [3] sum of ready nodes [4] sum of listed nodes [5] freezes [6] max ready [7] member-evaluations that changed something
extern "C" long lab_bundle(int G, int nnx, int nnz, float* Tio, float* tauio, const float* slow_rm, const float* risti, float ri, float dnx, float dnz,
                           const float* windows, int rule, int pilot, int max_rounds, long* out)
{
    const int nbz = tiles_of(nnz);
    const size_t n = (size_t)tiles_of(nnx) * nbz * kTileRecs, nrm = (size_t)nnx * nnz;
    std::vector<Member> M(G);
    for (int g = 0; g < G; ++g) {
        Member& m = M[g];
        m.F.assign(n, Rec{ kInf, kInf }); m.slow.assign(n, 1.0f);
        for (int ix = 0; ix < nnx; ++ix)
            for (int iz = 0; iz < nnz; ++iz) {
                const int id = rec_index(nbz, iz, ix);
                m.F[id] = Rec{ Tio[g * nrm + (size_t)ix * nnz + iz], tauio[g * nrm + (size_t)ix * nnz + iz] };
                m.slow[id] = slow_rm[g * nrm + (size_t)ix * nnz + iz];
            }
        m.window = windows[g]; m.theta = kInf; m.tmin = kInf; m.best_tmin = -kInf; m.freeze = -kInf; m.hsh = 0u; m.stall = 0;
        m.hist[0] = 1u; m.hist[1] = 2u; m.hist[2] = 3u; m.hist[3] = 4u; m.evals = m.changes = m.freezes = 0;
    }
    long small_changes = 0, rounds = 0, node_evals = 0, member_evals = 0, sum_ready = 0, sum_listed = 0, max_ready = 0, changed_evals = 0, freezes = 0;
    auto in_grid = [&](int iz0, int ix0) { return ix0 >= 0 && ix0 < nnx && iz0 >= 0 && iz0 < nnz; };
    auto hood = [&](const Member& m, int iz, int ix) {   // 0-based
        Hood h;
        const int nz[4] = { iz, iz, iz - 1, iz + 1 }, nx[4] = { ix - 1, ix + 1, ix, ix };
        const int oz[4] = { iz, iz, iz - 2, iz + 2 }, ox[4] = { ix - 2, ix + 2, ix, ix };
        for (int q = 0; q < 4; ++q) {
            h.in[q] = in_grid(nz[q], nx[q]); h.in_outer[q] = in_grid(oz[q], ox[q]);
            const Rec a = h.in[q] ? m.F[rec_index(nbz, nz[q], nx[q])] : Rec{ kInf, kInf };
            const Rec b = h.in_outer[q] ? m.F[rec_index(nbz, oz[q], ox[q])] : Rec{ kInf, kInf };
            h.near_[q] = a.T; h.near_tau[q] = a.tau; h.outer[q] = b.T; h.outer_tau[q] = b.tau;
        }
        return h;
    };
    auto lower_bound = [&](const Member& m, int iz0, int ix0) {
        auto tv = [&](int z, int x) { return in_grid(z, x) ? tau_value(m.F[rec_index(nbz, z, x)].tau) : kInf; };
        return fminf(fminf(tv(iz0, ix0 - 1), tv(iz0, ix0 + 1)), fminf(tv(iz0 - 1, ix0), tv(iz0 + 1, ix0)));
    };

    // ---- one run of the shared schedule over the members [g0, g1)
    auto run = [&](int g0, int g1) -> long {
        const int ng = g1 - g0;
        std::vector<int> cur, next;
        std::vector<unsigned char> queued(n, 0);
        std::vector<unsigned short> pending(rule == 3 ? n : 0, 0);     // rule 3: members waiting for an evaluation at the node; != 0 <=> the node is listed
        std::vector<int> pushed(rule == 3 ? n : 0, -1);               // rule 3: round in which the node was put on `next`
        long r = 0;
        auto act = [&](int iz0, int ix0, int g) {
            if (!in_grid(iz0, ix0)) return;
            const int id = rec_index(nbz, iz0, ix0);
            if (t_pinned(M[g].F[id].T)) return;                  // (the pinned sets differ: each member's own band march)
            if (rule == 3) {
                const unsigned short old = pending[id];
                pending[id] = (unsigned short)(old | (1u << (g - g0)));
                if (!old) { next.push_back(id); pushed[id] = (int)r; }
                return;
            }
            if (queued[id]) return;
            queued[id] = 1; next.push_back(id);
        };
        r = -1;
        for (int ix = 0; ix < nnx; ++ix) for (int iz = 0; iz < nnz; ++iz)
            for (int g = g0; g < g1; ++g)
                if (t_pinned(M[g].F[rec_index(nbz, iz, ix)].T)) { act(iz, ix - 1, g); act(iz, ix + 1, g); act(iz - 1, ix, g); act(iz + 1, ix, g); }
        cur.swap(next);
        r = 0;
        struct R { int id; unsigned short due; };
        std::vector<R> ready;
        std::vector<float> nT, nK;
        auto apply = [&](int g, int id, float c, float k) {
            Member& m = M[g];
            if (!std::memcmp(&c, &m.F[id].T, 4) && !std::memcmp(&k, &m.F[id].tau, 4)) return;
            ++m.changes; ++changed_evals;
            const float t_lo = fminf(t_value(m.F[id].T), c), k_lo = fminf(tau_value(m.F[id].tau), k);
            const bool small = fabsf(c - t_value(m.F[id].T)) <= 4e-7f * c && fabsf(k - tau_value(m.F[id].tau)) <= 4e-7f * k;
            if (small) ++small_changes;
            m.F[id].T = c; m.F[id].tau = k;
            { unsigned a, b; std::memcpy(&a, &c, 4); std::memcpy(&b, &k, 4); m.hsh += ((unsigned)id * 2654435761u) ^ (a * 40503u) ^ (b * 2246822519u); }
            int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
            const int dz[4] = { 0, 0, -1, 1 }, dx[4] = { -1, 1, 0, 0 };
            for (int q = 0; q < 4; ++q) {
                const int yz = iz0 + dz[q], yx = ix0 + dx[q], zz = iz0 + 2 * dz[q], zx = ix0 + 2 * dx[q];
                if (!in_grid(yz, yx)) continue;
                const Rec y = m.F[rec_index(nbz, yz, yx)];
                if (k_lo <= tau_value(y.tau)) act(yz, yx, g);
                if (!in_grid(zz, zx)) continue;
                if (!(tau_value(y.tau) < kInf)) continue;
                const Rec zr = m.F[rec_index(nbz, zz, zx)];
                if (t_value(y.T) > t_lo && k_lo < tau_value(zr.tau)) act(zz, zx, g);
            }
            m.tmin = fminf(m.tmin, k);
            if (rule == 4 || (rule == 5 && small)) { Member& pm = M[std::min(std::max(pilot, g0), g1 - 1)]; if (!t_pinned(pm.F[id].T)) pm.tmin = fminf(pm.tmin, tau_value(pm.F[id].tau)); }
        };
        auto eval_batch = [&](const std::vector<R>& sub) {
            if (sub.empty()) return;
            nT.resize(sub.size() * ng); nK.resize(sub.size() * ng);
            for (size_t k = 0; k < sub.size(); ++k) {
                int iz0, ix0; rec_coords(nbz, sub[k].id, &iz0, &ix0);
                const NodeGeom geo = { ri, risti[ix0], dnx, dnz };
                ++node_evals;
                for (int g = g0; g < g1; ++g) {
                    if (!((sub[k].due >> (g - g0)) & 1)) continue;
                    const Hood h = hood(M[g], iz0, ix0);
                    nT[k * ng + g - g0] = solve_node(h, M[g].slow[sub[k].id], geo, &nK[k * ng + g - g0]); ++M[g].evals; ++member_evals;
                }
            }
            for (size_t k = 0; k < sub.size(); ++k)
                for (int g = g0; g < g1; ++g) if ((sub[k].due >> (g - g0)) & 1) apply(g, sub[k].id, nT[k * ng + g - g0], nK[k * ng + g - g0]);
        };
        while (!cur.empty()) {
            for (int g = g0; g < g1; ++g) M[g].tmin = kInf;
            ready.clear();
            sum_listed += (long)cur.size();
            for (int id : cur) {
                int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
                unsigned short due = 0, cand = 0;
                float lb[16];
                for (int g = g0; g < g1; ++g) {
                    const Member& m = M[g];
                    const unsigned short bit = (unsigned short)(1u << (g - g0));
                    lb[g - g0] = kInf;
                    if (rule == 3 && !(pending[id] & bit)) continue;
                    if (t_pinned(m.F[id].T)) { if (rule == 3) pending[id] &= (unsigned short)~bit; continue; }
                    if (tau_value(m.F[id].tau) < m.freeze) { if (rule == 3) pending[id] &= (unsigned short)~bit; continue; }
                    cand |= bit;
                    lb[g - g0] = lower_bound(m, iz0, ix0);
                    if (rule == 6 && g == g0 && m.theta < kInf) {
                        auto tv = [&](int z, int x) { return in_grid(z, x) ? tau_value(m.F[rec_index(nbz, z, x)].tau) : kInf; };
                        float a[4] = { tv(iz0, ix0 - 1), tv(iz0, ix0 + 1), tv(iz0 - 1, ix0), tv(iz0 + 1, ix0) };
                        std::sort(a, a + 4);
                        if (a[1] < m.theta || a[0] < m.theta - m.window * 0.01f * (float)pilot) due |= bit;
                    } else
                    if (!(m.theta < kInf) || lb[g - g0] < m.theta) due |= bit;
                }
                if (rule == 2 || rule == 4 || rule == 5 || rule == 6) { const int pg = rule == 6 ? 0 : std::min(std::max(pilot, g0), g1 - 1) - g0; if ((cand >> pg) & 1) due = ((due >> pg) & 1) ? cand : 0; else due = due ? cand : 0; }   // (the pilot pinned or frozen here: any member)
                if (rule == 1 || rule == 0) due = due ? cand : 0;
                const int par = (iz0 + ix0) & 1;
                if (!cand) { queued[id] = 0; continue; }
                const unsigned short waiting = (unsigned short)(cand & ~due);
                for (int g = g0; g < g1; ++g) if ((waiting >> (g - g0)) & 1) M[g].tmin = fminf(M[g].tmin, lb[g - g0]);
                if (due) ready.push_back(R{ id, due });
                if (rule == 3) {
                    if (!par) pending[id] &= (unsigned short)~due;        // (an odd node's bits stay up through the even half: what that half activates for them is dropped)
                    if (waiting) { next.push_back(id); pushed[id] = (int)r; }
                } else {
                    if (due) { if (!par) queued[id] = 0; }
                    else next.push_back(id);
                }
            }
            sum_ready += (long)ready.size(); max_ready = std::max(max_ready, (long)ready.size());
            std::vector<R> sub;
            for (int p = 0; p < 2; ++p) {
                sub.clear();
                for (auto& q : ready) { int iz0, ix0; rec_coords(nbz, q.id, &iz0, &ix0); if (((iz0 + ix0) & 1) == p) sub.push_back(q); }
                if (p == 1) for (auto& q : sub) {
                    if (rule == 3) {
                        pending[q.id] &= (unsigned short)~q.due;
                        if (pending[q.id] && pushed[q.id] != (int)r) { next.push_back(q.id); pushed[q.id] = (int)r; }    // the even half activated it for a member that was not due
                    } else queued[q.id] = 0;
                }
                eval_batch(sub);
            }
            bool all_repeat = true, any_hash = false;
            for (int g = g0; g < g1; ++g) {
                Member& m = M[g];
                if (m.tmin > m.best_tmin && m.tmin < kInf) m.best_tmin = m.tmin;
                const bool repeat = m.hsh != 0u && (m.hsh == m.hist[1] || m.hsh == m.hist[2] || m.hsh == m.hist[3] || m.hsh == m.hist[0]);
                m.hist[3] = m.hist[2]; m.hist[2] = m.hist[1]; m.hist[1] = m.hist[0]; m.hist[0] = m.hsh;
                if (m.hsh) { any_hash = true; if (!repeat) all_repeat = false; }
                m.hsh = 0u;
                if (repeat) { if (++m.stall >= 8) { m.freeze = m.best_tmin + m.window; m.stall = 0; ++m.freezes; ++freezes; } } else m.stall = 0;
                m.theta = m.tmin + m.window;
            }
            (void)all_repeat; (void)any_hash;
            cur.swap(next); next.clear(); ++r;
            if (r >= max_rounds) return -1;
        }
        rounds += r;
        return 0;
    };
    long rc = 0;
    if (rule == 0) { for (int g = 0; g < G && rc == 0; ++g) rc = run(g, g + 1); }
    else rc = run(0, G);
    for (int g = 0; g < G; ++g)
        for (int ix = 0; ix < nnx; ++ix)
            for (int iz = 0; iz < nnz; ++iz) { const Rec q = M[g].F[rec_index(nbz, iz, ix)]; Tio[g * nrm + (size_t)ix * nnz + iz] = q.T; tauio[g * nrm + (size_t)ix * nnz + iz] = q.tau; }
    out[0] = rounds; out[1] = node_evals; out[2] = member_evals; out[3] = sum_ready; out[4] = sum_listed; out[5] = freezes; out[6] = max_ready; out[7] = changed_evals; out[8] = small_changes;
    return rc;
}
