// SCHEDULE LABORATORY (test infrastructure / probe, not product): the fixed-point iteration of csrc/fim_kernel.hip replayed on the
// CPU with the product's own per-node solver (csrc/eikonal_core.h), under alternative round schedules, to count rounds and
// evaluations before anything is built on the GPU.  The fixed point is schedule independent, so every variant must return the
// same field (checked by the caller).  Variants (`mode`):
//   0  one pass per round over all ready nodes (Jacobi)
//   1  two sub-passes by node parity (the device schedule of k_fim_sorted)
//   2  K sub-bands ordered by lower bound, no parity          (param = K)
//   3  K sub-bands ordered by lower bound x parity            (param = K)
//   4  sequential Gauss-Seidel in lower-bound order (the ordering limit for this window)
//   6  like 1, but the odd nodes leave the active set only after the even sub-pass (activations made by the even half do not re-queue them)
//   7  like 6, with a narrower window for the even nodes (param = per cent of the window)
//   8  like 6, the colour that goes first alternates from round to round
//   9  like 6, a node is ready when its SECOND earliest neighbour is inside the window, or its earliest one lies param per cent of a window back
//   10 four colours (ix & 1, iz & 1) in the order given by param's decimal digits, each leaving the active set right before its sub-pass
//   11 like 6 with a window steered towards `param` ready nodes per round (proportional control, 0.25 .. 4 x the given window)
//   12 key slots (no lower bounds at all): a node is due when the time slot (width window / param%100 ... see code) of the smallest acceptance
//      time among the neighbours that activated it since its last evaluation lies inside the window; odd rule as in 6
//   13 like 6, but an activation whose key (the activator's acceptance time) is not below theta is never dropped: it re-queues an odd node
//      that the same round's odd half is about to evaluate (what a separate "needs a lower bound" mask per tile would do)
//   14 like 6, plus `param` extra evaluation passes per round WITHOUT a listing pass: the nodes activated during the round by a node whose
//      acceptance time lies below the round's theta (their lower bound is below it too) are evaluated right away, even then odd, from a
//      list filled at activation time; the rest waits for the next listing pass.  "rounds" counts listing passes.
//   5  lazy: a node waits for the acceptance time of the neighbour that activated it to enter the window (key routing), parity sub-passes
// build: g++ -O2 -std=c++17 -fPIC -shared -ffp-contract=off -msse2 -mfpmath=sse -o tests/tools/libsched_lab.so tests/tools/sched_lab.cpp
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../dsurftomo_amd/csrc/eikonal_core.h"

using namespace dsa;

namespace {
struct Field { int nnx, nnz, nbz; Rec* F; const float* slow; const float* risti; float ri, dnx, dnz; };
Hood load_hood(const Field& f, int iz, int ix)   // 1-based
{
    Hood h;
    const int nz[4] = { iz, iz, iz - 1, iz + 1 }, nx[4] = { ix - 1, ix + 1, ix, ix };
    const int oz[4] = { iz, iz, iz - 2, iz + 2 }, ox[4] = { ix - 2, ix + 2, ix, ix };
    for (int q = 0; q < 4; ++q) {
        h.in[q] = nx[q] >= 1 && nx[q] <= f.nnx && nz[q] >= 1 && nz[q] <= f.nnz;
        h.in_outer[q] = ox[q] >= 1 && ox[q] <= f.nnx && oz[q] >= 1 && oz[q] <= f.nnz;
        const Rec a = h.in[q] ? f.F[rec_index(f.nbz, nz[q] - 1, nx[q] - 1)] : Rec{ kInf, kInf };
        const Rec b = h.in_outer[q] ? f.F[rec_index(f.nbz, oz[q] - 1, ox[q] - 1)] : Rec{ kInf, kInf };
        h.near_[q] = a.T; h.near_tau[q] = a.tau; h.outer[q] = b.T; h.outer_tau[q] = b.tau;
    }
    return h;
}
}

extern "C" long lab_schedule(int nnx, int nnz, float* Tio, float* tauio, const float* slow_rm, const float* risti, float ri, float dnx,
                             float dnz, float window, int mode, int param, int max_rounds,
                             long* out /* rounds, evals, changes, freezes, subpasses, sum of ready, sum of listed, max ready */)
{
    const int nbz = tiles_of(nnz);
    std::vector<Rec> F((size_t)tiles_of(nnx) * nbz * kTileRecs, Rec{ kInf, kInf });
    std::vector<float> slow(F.size(), 1.0f);
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const int id = rec_index(nbz, iz, ix);
            F[id] = Rec{ Tio[(size_t)ix * nnz + iz], tauio[(size_t)ix * nnz + iz] };
            slow[id] = slow_rm[(size_t)ix * nnz + iz];
        }
    Field f = { nnx, nnz, nbz, F.data(), slow.data(), risti, ri, dnx, dnz };
    const size_t n = F.size();
    std::vector<int> cur, next;
    std::vector<float> curkey, nextkey;
    std::vector<unsigned char> queued(n, 0);
    std::vector<float> key(n, kInf);
    std::vector<unsigned char> pending_odd(n, 0), requeue(n, 0), was_deferred(n, 0);
    std::vector<int> tile_act(n / 64 + 1, -1);      // round of the last activation landing in the tile
    long sleeping_listings = 0, sleeping_tiles = 0, listed_tiles = 0;
    float theta = kInf;
    long rounds_now = 0;
    float slot_w = 0.0f; long slotB = 0; int ring = 1 << 30, Wslots = 2;
    auto act = [&](int iz0, int ix0, float k) {
        if (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) return;
        if (slot_w > 0.0f) { long a = (long)floorf(k / slot_w); if (a < slotB) a = slotB; if (a > slotB + ring - 1) a = slotB + ring - 1; k = (float)a; }   // the key becomes a slot number
        const int id = rec_index(nbz, iz0, ix0);
        if (t_pinned(F[id].T)) return;
        tile_act[id >> 6] = (int)rounds_now;
        if (queued[id]) { if (k < key[id]) key[id] = k; if (mode == 13 && pending_odd[id] && !(k < theta)) requeue[id] = 1; return; }
        queued[id] = 1; key[id] = k; next.push_back(id);
    };
    for (int ix = 0; ix < nnx; ++ix) for (int iz = 0; iz < nnz; ++iz)
        if (t_pinned(F[rec_index(nbz, iz, ix)].T)) { act(iz, ix - 1, 0.f); act(iz, ix + 1, 0.f); act(iz - 1, ix, 0.f); act(iz + 1, ix, 0.f); }
    cur.swap(next);
    long key_routed = 0, extra_listed = 0; long rounds = 0, evals = 0, changes = 0, subpasses = 0, sum_ready = 0, sum_listed = 0, max_ready = 0, trips256 = 0, trips128 = 0;
    float wnow = window;
    float best_tmin = -kInf, freeze = -kInf; int stall = 0; long freezes = 0;
    unsigned hist[4] = { 1u, 2u, 3u, 4u }, hsh = 0u;
    auto tv = [&](int iz0, int ix0) { return (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) ? kInf : tau_value(F[rec_index(nbz, iz0, ix0)].tau); };
    struct R { int id; float lb; };
    std::vector<R> ready;
    std::vector<float> nT, nK;
    float tmin = kInf;
    auto apply = [&](int id, float c, float k) {
        if (!std::memcmp(&c, &F[id].T, 4) && !std::memcmp(&k, &F[id].tau, 4)) return;
        ++changes;
        const float t_lo = fminf(t_value(F[id].T), c), k_lo = fminf(tau_value(F[id].tau), k);
        F[id].T = c; F[id].tau = k;
        { unsigned a, b; std::memcpy(&a, &c, 4); std::memcpy(&b, &k, 4); hsh += ((unsigned)id * 2654435761u) ^ (a * 40503u) ^ (b * 2246822519u); }
        int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
        const int dz[4] = { 0, 0, -1, 1 }, dx[4] = { -1, 1, 0, 0 };
        for (int q = 0; q < 4; ++q) {
            const int yz = iz0 + dz[q], yx = ix0 + dx[q], zz = iz0 + 2 * dz[q], zx = ix0 + 2 * dx[q];
            if (yx < 0 || yx >= nnx || yz < 0 || yz >= nnz) continue;
            const Rec y = F[rec_index(nbz, yz, yx)];
            if (k_lo <= tau_value(y.tau)) act(yz, yx, k);
            if (zx < 0 || zx >= nnx || zz < 0 || zz >= nnz) continue;
            if (!(tau_value(y.tau) < kInf)) continue;
            const Rec zr = F[rec_index(nbz, zz, zx)];
            if (t_value(y.T) > t_lo && k_lo < tau_value(zr.tau)) act(zz, zx, (mode == 12 && param / 10000) ? k : tau_value(y.tau));
        }
        tmin = fminf(tmin, k);
    };
    auto eval_batch = [&](const std::vector<int>& sub) {
        if (sub.empty()) return;
        ++subpasses;
        nT.resize(sub.size()); nK.resize(sub.size());
        for (size_t k = 0; k < sub.size(); ++k) {
            int iz0, ix0; rec_coords(nbz, sub[k], &iz0, &ix0);
            const Hood h = load_hood(f, iz0 + 1, ix0 + 1); const NodeGeom g = { ri, risti[ix0], dnx, dnz };
            nT[k] = solve_node(h, slow[sub[k]], g, &nK[k]); ++evals;
        }
        for (size_t k = 0; k < sub.size(); ++k) apply(sub[k], nT[k], nK[k]);
    };
    if (mode == 12) { Wslots = param % 100 ? param % 100 : 2; ring = (param / 100) % 100 ? (param / 100) % 100 : 4; slot_w = window / (float)Wslots;
        for (int id : cur) key[id] = 0.0f; }
    while (!cur.empty()) {
        tmin = kInf; ready.clear();
        sum_listed += (long)cur.size();
        {   // statistics: tiles all of whose listed nodes were deferred last round and that no activation has reached since ("sleeping")
            std::vector<int> tiles;
            for (int id : cur) tiles.push_back(id >> 6);
            std::sort(tiles.begin(), tiles.end()); tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
            std::vector<char> awake(tiles.size(), 0);
            for (int id : cur) { const size_t t = std::lower_bound(tiles.begin(), tiles.end(), id >> 6) - tiles.begin(); if (!was_deferred[id] || tile_act[id >> 6] >= (int)rounds_now - 1) awake[t] = 1; }
            listed_tiles += (long)tiles.size();
            for (size_t t = 0; t < tiles.size(); ++t) if (!awake[t]) ++sleeping_tiles;
            for (int id : cur) { const size_t t = std::lower_bound(tiles.begin(), tiles.end(), id >> 6) - tiles.begin(); if (!awake[t]) ++sleeping_listings; }
        }
        for (int id : cur) {
            if (tau_value(F[id].tau) < freeze) { queued[id] = 0; continue; }
            int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
            float lb = fminf(fminf(tv(iz0, ix0 - 1), tv(iz0, ix0 + 1)), fminf(tv(iz0 - 1, ix0), tv(iz0 + 1, ix0)));
            if (!(theta < kInf) || key[id] < theta) ++key_routed;          // statistics: listings that the activator's acceptance time alone routes (no neighbour loads)
            if (mode == 5) lb = fmaxf(lb, fminf(key[id], kInf));            // lazy: route by the activator's acceptance time
            if (mode == 12) lb = key[id] * slot_w;                         // slot lower edge
            float lb2 = lb;
            if (mode == 9) {
                float a[4] = { tv(iz0, ix0 - 1), tv(iz0, ix0 + 1), tv(iz0 - 1, ix0), tv(iz0 + 1, ix0) };
                std::sort(a, a + 4); lb2 = a[1];
            }
            int jz, jx; rec_coords(nbz, id, &jz, &jx);
            int par = (jx + jz) & 1;
            if (mode == 8 && (rounds & 1)) par ^= 1;
            float th = theta;
            if (mode == 7 && par == 0 && theta < kInf) th = theta - window * (1.0f - 0.01f * (float)param);
            bool rdy = !(theta < kInf) || lb < th;
            if (mode == 12) rdy = !(theta < kInf) || (long)key[id] < slotB + Wslots;
            if (mode == 9 && theta < kInf) rdy = lb2 < theta || lb < theta - window * 0.01f * (float)param;
            if (rdy) {
                ready.push_back(R{ id, lb });
                if (!(mode >= 6 && par) && mode != 10) queued[id] = 0;
                if (mode == 13 && par) pending_odd[id] = 1;
                key[id] = kInf;
            }
            else { next.push_back(id); tmin = fminf(tmin, lb); }
            was_deferred[id] = rdy ? 0 : 1;
        }
        sum_ready += (long)ready.size(); max_ready = std::max(max_ready, (long)ready.size());
        auto parity = [&](int id) { int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0); return (ix0 + iz0) & 1; };
        std::vector<int> sub;
        if (mode == 0) { for (auto& r : ready) sub.push_back(r.id); eval_batch(sub); }
        else if (mode == 10) {
            int order[4] = { (param / 1000) % 10, (param / 100) % 10, (param / 10) % 10, param % 10 };
            for (int p = 0; p < 4; ++p) {
                sub.clear();
                for (auto& r : ready) { int jz, jx; rec_coords(nbz, r.id, &jz, &jx); if (((jx & 1) * 2 + (jz & 1)) == order[p]) sub.push_back(r.id); }
                for (int id : sub) queued[id] = 0;
                eval_batch(sub);
            }
        }
        else if (mode == 1 || mode >= 5) {   // (parity sub-passes)
            for (int p = 0; p < 2; ++p) {
                { long cnt = 0; for (auto& r : ready) if (parity(r.id) == p) ++cnt; trips256 += (cnt + 255) / 256; trips128 += (cnt + 127) / 128; }
                const int want = (mode == 8 && (rounds & 1)) ? p ^ 1 : p;
                sub.clear(); for (auto& r : ready) if (parity(r.id) == want) sub.push_back(r.id);
                if (mode >= 6 && p == 1) for (int id : sub) {
                    queued[id] = 0;
                    if (mode == 13) { pending_odd[id] = 0; if (requeue[id]) { requeue[id] = 0; queued[id] = 1; next.push_back(id); } }
                }
                eval_batch(sub);
            }
        } else if (mode == 2 || mode == 3) {
            const int K = param < 1 ? 1 : param;
            float lo = kInf; for (auto& r : ready) lo = fminf(lo, r.lb);
            const float hi = theta < kInf ? theta : lo + window;
            for (int b = 0; b < K; ++b) {
                const float a0 = lo + (hi - lo) * (float)b / (float)K, a1 = b == K - 1 ? kInf : lo + (hi - lo) * (float)(b + 1) / (float)K;
                for (int p = 0; p < (mode == 3 ? 2 : 1); ++p) {
                    sub.clear();
                    for (auto& r : ready) if ((b == 0 || r.lb >= a0) && r.lb < a1 && (b > 0 || r.lb < a1) && (mode == 2 || parity(r.id) == p)) sub.push_back(r.id);
                    eval_batch(sub);
                }
            }
        } else if (mode == 4) {
            std::stable_sort(ready.begin(), ready.end(), [](const R& a, const R& b) { return a.lb < b.lb; });
            for (auto& r : ready) { sub.assign(1, r.id); eval_batch(sub); } --subpasses; subpasses -= (long)ready.size() - 1 > 0 ? (long)ready.size() - 1 : 0; ++subpasses;
        }
        if (mode == 14) {
            for (int extra = 0; extra < (param > 0 ? param : 1); ++extra) {
                // from `next`: what was activated with a key below theta; it leaves `next` (evaluated now)
                std::vector<int> fast, keep;
                for (int id : next) { if (key[id] < theta && tau_value(F[id].tau) >= freeze) fast.push_back(id); else keep.push_back(id); }
                if (fast.empty()) break;
                next.swap(keep);
                extra_listed += (long)fast.size();
                for (int p = 0; p < 2; ++p) {
                    sub.clear();
                    for (int id : fast) if (parity(id) == p) sub.push_back(id);
                    if (p == 0) for (int id : fast) if (parity(id) == 0) { queued[id] = 0; key[id] = kInf; }
                    if (p == 1) for (int id : sub) { queued[id] = 0; key[id] = kInf; }
                    { long cnt = (long)sub.size(); trips256 += (cnt + 255) / 256; }
                    eval_batch(sub);
                }
            }
        }
        if (tmin > best_tmin) best_tmin = tmin;
        {
            const bool repeat = hsh != 0u && (hsh == hist[1] || hsh == hist[2] || hsh == hist[3] || hsh == hist[0]);
            hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = hsh; hsh = 0u;
            if (repeat) { if (++stall >= 8) { freeze = best_tmin + window; stall = 0; ++freezes; } } else stall = 0;
        }
        if (mode == 11) {
            const float ratio = (float)param / (float)std::max<size_t>(ready.size(), 1);
            wnow = wnow * fminf(fmaxf(ratio, 0.7f), 1.4f);
            wnow = fminf(fmaxf(wnow, 0.25f * window), 4.0f * window);
        }
        if (mode == 12) slotB = (long)floorf(tmin / slot_w);
        cur.swap(next); next.clear(); theta = tmin + wnow; ++rounds; rounds_now = rounds;
        if (rounds >= max_rounds) break;
    }
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) { const Rec r = F[rec_index(nbz, iz, ix)]; Tio[(size_t)ix * nnz + iz] = r.T; tauio[(size_t)ix * nnz + iz] = r.tau; }
    out[0] = rounds; out[1] = evals; out[2] = changes; out[3] = freezes; out[4] = subpasses; out[5] = sum_ready; out[6] = sum_listed; out[7] = max_ready; out[8] = trips256; out[9] = trips128; out[10] = key_routed; out[11] = extra_listed; out[12] = sleeping_listings; out[13] = sleeping_tiles; out[14] = listed_tiles;
    return cur.empty() ? 0 : -1;
}
