// TEST HARNESS (not product): drives the product's __host__ __device__ per-node / per-source
// logic (dsurftomo_amd/csrc/{eikonal_core,source_stage,host_geometry}.h) on the CPU so that it
// can be compared with the oracle without a GPU.  The device iteration (block list, LDS tiles,
// ballots) is replaced by a plain worklist here; the fixed point it reaches is schedule
// independent, which is exactly what tests/test_hostcheck.py asserts against the oracle.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../dsurftomo_amd/csrc/host_geometry.h"

using namespace dsa;

namespace {

struct Field {
    int nnx, nnz;
    float* T;
    float* tau;
    const float* slow;
    const float* risti;
    float ri, dnx, dnz;
};

Hood load_hood(const Field& f, int iz, int ix)
{
    Hood h;
    const int nz[4] = { iz, iz, iz - 1, iz + 1 }, nx[4] = { ix - 1, ix + 1, ix, ix };
    const int oz[4] = { iz, iz, iz - 2, iz + 2 }, ox[4] = { ix - 2, ix + 2, ix, ix };
    for (int q = 0; q < 4; ++q) {
        h.in[q] = nx[q] >= 1 && nx[q] <= f.nnx && nz[q] >= 1 && nz[q] <= f.nnz;
        h.in_outer[q] = ox[q] >= 1 && ox[q] <= f.nnx && oz[q] >= 1 && oz[q] <= f.nnz;
        h.near_[q] = h.in[q] ? f.T[(size_t)(nx[q] - 1) * f.nnz + (nz[q] - 1)] : kInf;
        h.outer[q] = h.in_outer[q] ? f.T[(size_t)(ox[q] - 1) * f.nnz + (oz[q] - 1)] : kInf;
        h.near_tau[q] = h.in[q] ? f.tau[(size_t)(nx[q] - 1) * f.nnz + (nz[q] - 1)] : kInf;
        h.outer_tau[q] = h.in_outer[q] ? f.tau[(size_t)(ox[q] - 1) * f.nnz + (oz[q] - 1)] : kInf;
    }
    return h;
}

long fixed_point(Field& f)
{
    const size_t n = (size_t)f.nnx * f.nnz;
    std::vector<int> q(n * 2 + 64);
    std::vector<unsigned char> inq(n, 0);
    size_t head = 0, tail = 0, cap = q.size();
    long evals = 0;
    auto push = [&](int iz, int ix) {
        if (ix < 1 || ix > f.nnx || iz < 1 || iz > f.nnz) return;
        const size_t id = (size_t)(ix - 1) * f.nnz + (iz - 1);
        if (t_pinned(f.T[id]) || inq[id]) return;
        inq[id] = 1; q[tail] = (int)id; tail = (tail + 1) % cap;
    };
    for (int ix = 1; ix <= f.nnx; ++ix)
        for (int iz = 1; iz <= f.nnz; ++iz)
            if (t_pinned(f.T[(size_t)(ix - 1) * f.nnz + (iz - 1)])) {
                push(iz, ix - 1); push(iz, ix + 1); push(iz - 1, ix); push(iz + 1, ix);
            }
    while (head != tail) {
        const size_t id = (size_t)q[head]; head = (head + 1) % cap; inq[id] = 0;
        const int ix = (int)(id / f.nnz) + 1, iz = (int)(id % f.nnz) + 1;
        const Hood h = load_hood(f, iz, ix);
        const NodeGeom g = { f.ri, f.risti[ix - 1], f.dnx, f.dnz };
        float k;
        const float c = solve_node(h, f.slow[id], g, &k);
        ++evals;
        if (std::memcmp(&c, &f.T[id], 4) != 0 || std::memcmp(&k, &f.tau[id], 4) != 0) {
            f.T[id] = c; f.tau[id] = k;
            push(iz, ix - 1); push(iz, ix + 1); push(iz - 1, ix); push(iz + 1, ix);
            push(iz, ix - 2); push(iz, ix + 2); push(iz - 2, ix); push(iz + 2, ix);
        }
        if (evals > 400L * (long)n) return -1;
    }
    return evals;
}

}  // namespace

extern "C" {

// fouds2 on a field with an explicit alive mask; mirrors oracle dso_fouds2_masked for unit tests
float hc_fouds2_masked(int nnx, int nnz, float gox, float dnx, float dnz, float earth, const float* veln,
                       const float* ttn, const unsigned char* alive, int iz, int ix)
{
    Stencil s;
    auto al = [&](int z, int x) { return x >= 1 && x <= nnx && z >= 1 && z <= nnz && alive[(size_t)(x - 1) * nnz + (z - 1)] != 0; };
    auto tt = [&](int z, int x) { return ttn[(size_t)(x - 1) * nnz + (z - 1)]; };
    const int jx[2] = { ix - 1, ix + 1 }, jx2[2] = { ix - 2, ix + 2 }, kz[2] = { iz - 1, iz + 1 }, kz2[2] = { iz - 2, iz + 2 };
    for (int d = 0; d < 2; ++d) {
        s.ej[d] = jx[d] >= 1 && jx[d] <= nnx;
        s.aj[d] = al(iz, jx[d]); s.tj[d] = s.aj[d] ? tt(iz, jx[d]) : kInf;
        s.oj[d] = al(iz, jx2[d]); s.tj2[d] = s.oj[d] ? tt(iz, jx2[d]) : kInf;
        s.ek[d] = kz[d] >= 1 && kz[d] <= nnz;
        s.ak[d] = al(kz[d], ix); s.tk[d] = s.ak[d] ? tt(kz[d], ix) : kInf;
        s.ok[d] = al(kz2[d], ix); s.tk2[d] = s.ok[d] ? tt(kz2[d], ix) : kInf;
    }
    const NodeGeom g = { earth, earth * sinf(gox + (float)(ix - 1) * dnx), dnx, dnz };
    return fouds2(s, 1.0f / veln[(size_t)(ix - 1) * nnz + (iz - 1)], g);
}

// coarse velocity field through the product's per-node gridder
void hc_gridder(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv, float* veln)
{
    GridDesc g; make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    std::vector<float> velv((size_t)nx * ny), basis(4 * (gd + 1));
    for (int k = 0; k < nx * ny; ++k) velv[k] = (float)pv[k];
    basis_table(gd, basis.data());
    for (int ix = 1; ix <= g.nnx; ++ix)
        for (int iz = 1; iz <= g.nnz; ++iz)
            veln[(size_t)(ix - 1) * g.nnz + (iz - 1)] = coarse_velocity(g, velv.data(), basis.data(), iz, ix);
}

// one (period, source) unit; outputs as oracle dso_solve_source. stats: [0] start-up ended early,
// [1] error flags, [2] evals refined, [3] evals coarse. Returns 0 / -1 (outside)
int hc_solve_source(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv,
                    float x, float z, float* ttn, float* ttnr_out, int* nstsr_out, float* inj_t, int* inj_s,
                    int* box_out, long* stats)
{
    GridDesc g; make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    SourceDesc s;
    if (make_source(g, x, z, s) != 0) return -1;
    const size_t nc = (size_t)g.nnx * g.nnz, nr = (size_t)s.rnx * s.rnz;
    std::vector<float> velv((size_t)nx * ny), cbasis(4 * (gd + 1)), rbasis(4 * (gd * kSgdl + 1));
    for (int k = 0; k < nx * ny; ++k) velv[k] = (float)pv[k];
    basis_table(gd, cbasis.data());
    basis_table(gd * kSgdl, rbasis.data());
    std::vector<float> slow_c(nc), risti_c(g.nnx), slow_r(nr), T_r(nr, kInf), tau_r(nr, kInf), tau_c(nc, kInf), risti_r(kRefMax), vcorner(4);
    for (int ix = 1; ix <= g.nnx; ++ix)
        for (int iz = 1; iz <= g.nnz; ++iz)
            slow_c[(size_t)(ix - 1) * g.nnz + (iz - 1)] = 1.0f / coarse_velocity(g, velv.data(), cbasis.data(), iz, ix);
    risti_table(g.gox, g.dnx, g.earth, g.nnx, risti_c.data());
    risti_table(s.rgox, s.rdnx, g.earth, s.rnx, risti_r.data());
    for (int lx = 1; lx <= s.rnx; ++lx)
        for (int kz = 1; kz <= s.rnz; ++kz) {
            const float v = refined_velocity(g, s, velv.data(), rbasis.data(), kz, lx);
            slow_r[(size_t)(lx - 1) * s.rnz + (kz - 1)] = 1.0f / v;
            if ((lx == s.isx_r || lx == s.isx_r + 1) && (kz == s.isz_r || kz == s.isz_r + 1))
                vcorner[(lx - s.isx_r) * 2 + (kz - s.isz_r)] = v;
        }
    std::vector<int16_t> rst(kRWin * kRWin), cst((size_t)kCWinMax * kCWinMax);
    std::vector<int8_t> S_r(nr), cinit((size_t)kCWinMax * kCWinMax);
    std::vector<int32_t> heap(kHeapCap), flags(2, 0);
    SourceScratch w;
    w.slow_r = slow_r.data(); w.T_r = T_r.data(); w.tau_r = tau_r.data(); w.S_r = S_r.data(); w.risti_r = risti_r.data();
    w.vcorner = vcorner.data(); w.rst = rst.data(); w.cst = cst.data(); w.cinit = cinit.data();
    w.heap = heap.data(); w.flags = flags.data();

    const int ended = refined_startup(g, s, w);
    refined_encode(s, w, ended);
    stats[2] = 0;
    if (!ended) {
        Field fr = { s.rnx, s.rnz, T_r.data(), tau_r.data(), slow_r.data(), risti_r.data(), g.earth, s.rdnx, s.rdnz };
        stats[2] = fixed_point(fr);
    }
    // first open-edge node in acceptance order (scan order ix outer, iz inner breaks exact ties)
    uint64_t rstar = ~0ull; int ez = 0, ex = 0;
    if (!ended)
        for (int ix = 1; ix <= s.rnx; ++ix)
            for (int iz = 1; iz <= s.rnz; ++iz)
                if (is_open_edge(s, iz, ix)) {
                    const size_t id = (size_t)(ix - 1) * s.rnz + (iz - 1);
                    if (!(t_value(T_r[id]) < kInf)) continue;
                    const uint64_t r = accept_rank(T_r[id], tau_r[id]);
                    if (r < rstar) { rstar = r; ez = iz; ex = ix; }
                }
    std::vector<float> Tfin(nr);
    for (int ix = 1; ix <= s.rnx; ++ix)
        for (int iz = 1; iz <= s.rnz; ++iz) {
            const size_t id = (size_t)(ix - 1) * s.rnz + (iz - 1);
            S_r[id] = (int8_t)handoff_node(g, s, w, ended, rstar, ez, ex, iz, ix, &Tfin[id]);
        }
    if (ttnr_out) std::memcpy(ttnr_out, Tfin.data(), 4 * nr);
    if (nstsr_out) for (size_t k = 0; k < nr; ++k) nstsr_out[k] = S_r[k];

    // injection + band promotion into the coarse window / field
    for (size_t k = 0; k < nc; ++k) ttn[k] = kInf;
    for (int q = 0; q < s.cwnx * s.cwnz; ++q) cst[q] = -1;
    auto cs = [&](int iz, int ix) -> int16_t& { return cst[(size_t)(ix - 1 - s.cwx0) * s.cwnz + (iz - 1 - s.cwz0)]; };
    for (int k = 1; k <= s.rnz; k += kSgdl)
        for (int l = 1; l <= s.rnx; l += kSgdl) {
            const int cz = s.vnt + (k - 1) / kSgdl, cx = s.vnl + (l - 1) / kSgdl;
            const size_t id = (size_t)(l - 1) * s.rnz + (k - 1);
            cs(cz, cx) = S_r[id];
            if (S_r[id] >= 0) ttn[(size_t)(cx - 1) * g.nnz + (cz - 1)] = Tfin[id];
        }
    auto far = [&](int iz, int ix) {
        if (ix < 1 || ix > g.nnx || iz < 1 || iz > g.nnz) return false;
        if (!(iz > s.cwz0 && iz <= s.cwz0 + s.cwnz && ix > s.cwx0 && ix <= s.cwx0 + s.cwnx)) return true;
        return cs(iz, ix) == -1;
    };
    // the reference promotes in place while scanning ix outer / iz inner; promoted nodes become
    // status 1, which is not "far", so the scan order does not matter
    for (int ix = s.vnl; ix <= s.vnr; ++ix)
        for (int iz = s.vnt; iz <= s.vnb; ++iz)
            if (cs(iz, ix) == 0 && (far(iz - 1, ix) || far(iz + 1, ix) || far(iz, ix - 1) || far(iz, ix + 1))) cs(iz, ix) = 1;
    if (inj_t) std::memcpy(inj_t, ttn, 4 * nc);
    if (inj_s) {
        for (size_t k = 0; k < nc; ++k) inj_s[k] = -1;
        for (int ix = s.cwx0 + 1; ix <= s.cwx0 + s.cwnx; ++ix)
            for (int iz = s.cwz0 + 1; iz <= s.cwz0 + s.cwnz; ++iz) inj_s[(size_t)(ix - 1) * g.nnz + (iz - 1)] = cs(iz, ix);
    }
    coarse_band_march(g, s, w, ttn, tau_c.data(), slow_c.data(), risti_c.data());
    Field fc = { g.nnx, g.nnz, ttn, tau_c.data(), slow_c.data(), risti_c.data(), g.earth, g.dnx, g.dnz };
    stats[3] = fixed_point(fc);
    for (size_t k = 0; k < nc; ++k) ttn[k] = t_value(ttn[k]);
    stats[0] = ended; stats[1] = flags[1];
    if (box_out) { box_out[0] = s.vnl; box_out[1] = s.vnr; box_out[2] = s.vnt; box_out[3] = s.vnb; box_out[4] = s.rnx; box_out[5] = s.rnz; }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Emulation of the device schedule of fim_kernel.hip (same routing, same theta update, the whole
// ready set evaluated from the old states), used to study convergence on the CPU.
// mode 0: all ready nodes at once; mode 1: two sub-passes by node parity (even first).
extern "C" long hc_device_schedule(int nnx, int nnz, float* T, float* tau, const float* slow, const float* risti,
                                   float ri, float dnx, float dnz, float window, int mode, int max_rounds,
                                   long* out /* rounds, evals, last list size */, int* cyc_ids, int ncyc)
{
    Field f = { nnx, nnz, T, tau, slow, risti, ri, dnx, dnz };
    const size_t n = (size_t)nnx * nnz;
    std::vector<int> cur, next, ready;
    std::vector<unsigned char> queued(n, 0);
    auto act = [&](long id) { if (id < 0 || (size_t)id >= n) return; if (t_pinned(T[id]) || queued[id]) return; queued[id] = 1; next.push_back((int)id); };
    for (int ix = 1; ix <= nnx; ++ix) for (int iz = 1; iz <= nnz; ++iz) {
        const size_t id = (size_t)(ix - 1) * nnz + (iz - 1);
        if (!t_pinned(T[id])) continue;
        if (ix > 1) act(id - nnz); if (ix < nnx) act(id + nnz); if (iz > 1) act(id - 1); if (iz < nnz) act(id + 1);
    }
    cur.swap(next);
    float theta = kInf; long rounds = 0, evals = 0;
    std::vector<float> nT, nK;
    float best_tmin = -kInf, freeze = -kInf; int stall = 0; long freezes = 0;
    unsigned hist[4] = { 1u, 2u, 3u, 4u }, hsh = 0u;
    while (!cur.empty()) {
        float tmin = kInf; ready.clear();
        for (int id : cur) {
            // accepted below the freeze horizon: final (see fim_kernel.hip, "stall")
            if (tau_value(tau[id]) < freeze) { queued[id] = 0; continue; }
            const int ix = id / nnz, iz = id - ix * nnz; float lb = kInf;
            if (ix > 0) lb = fminf(lb, tau_value(tau[id - nnz])); if (ix + 1 < nnx) lb = fminf(lb, tau_value(tau[id + nnz]));
            if (iz > 0) lb = fminf(lb, tau_value(tau[id - 1])); if (iz + 1 < nnz) lb = fminf(lb, tau_value(tau[id + 1]));
            if (!(theta < kInf) || lb < theta) { ready.push_back(id); queued[id] = 0; }
            else { next.push_back(id); tmin = fminf(tmin, lb); }
        }
        for (int pass = 0; pass < (mode == 1 ? 2 : 1); ++pass) {
            std::vector<int> sub;
            for (int id : ready) { const int ix = id / nnz, iz = id - ix * nnz; if (mode == 0 || ((ix + iz) & 1) == pass) sub.push_back(id); }
            nT.resize(sub.size()); nK.resize(sub.size());
            for (size_t k = 0; k < sub.size(); ++k) {
                const int id = sub[k]; const int ix = id / nnz + 1, iz = id % nnz + 1;
                const Hood h = load_hood(f, iz, ix); const NodeGeom g = { ri, risti[ix - 1], dnx, dnz };
                nT[k] = solve_node(h, slow[id], g, &nK[k]); ++evals;
            }
            for (size_t k = 0; k < sub.size(); ++k) {
                const int id = sub[k];
                if (std::memcmp(&nT[k], &T[id], 4) || std::memcmp(&nK[k], &tau[id], 4)) {
                    T[id] = nT[k]; tau[id] = nK[k];
                    { unsigned a, b; std::memcpy(&a, &nT[k], 4); std::memcpy(&b, &nK[k], 4); hsh += ((unsigned)id * 2654435761u) ^ (a * 40503u) ^ (b * 2246822519u); }
                    const int ix = id / nnz, iz = id - ix * nnz;
                    if (ix > 0) act(id - nnz); if (ix > 1) act(id - 2 * nnz); if (ix + 1 < nnx) act(id + nnz); if (ix + 2 < nnx) act(id + 2 * nnz);
                    if (iz > 0) act(id - 1); if (iz > 1) act(id - 2); if (iz + 1 < nnz) act(id + 1); if (iz + 2 < nnz) act(id + 2);
                    tmin = fminf(tmin, nK[k]);
                }
            }
        }
        if (ncyc < 0 && rounds >= max_rounds - 4) {
            std::printf("round %ld: theta %.7f tmin %.7f ready %zu next %zu\n", rounds, theta, tmin, ready.size(), next.size());
            for (size_t k = 0; k < ready.size() && k < 12; ++k) { const int id = ready[k]; std::printf("   ready ix=%d iz=%d T=%.7f tau=%.7f\n", id / nnz + 1, id % nnz + 1, T[id], tau[id]); }
        }
        if (tmin > best_tmin) best_tmin = tmin;
        {   // same rule as fim_kernel.hip: freeze only when the set of changes repeats exactly
            const bool repeat = hsh != 0u && (hsh == hist[1] || hsh == hist[2] || hsh == hist[3] || hsh == hist[0]);
            hist[3] = hist[2]; hist[2] = hist[1]; hist[1] = hist[0]; hist[0] = hsh; hsh = 0u;
            if (repeat) { if (++stall >= 8) { freeze = best_tmin + window; stall = 0; ++freezes; } } else stall = 0;
        }
        cur.swap(next); next.clear(); theta = tmin + window; ++rounds;
        if (rounds >= max_rounds) break;
    }
    out[0] = rounds; out[1] = evals; out[2] = (long)cur.size(); out[3] = freezes;
    for (int k = 0; k < (ncyc < 0 ? -ncyc : ncyc); ++k) cyc_ids[k] = k < (int)cur.size() ? cur[k] : -1;
    return cur.empty() ? 0 : -1;
}

// set-up helper: coarse problem of one source after the serial stages (T, tau, slow, risti filled)
extern "C" int hc_coarse_problem(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv,
                                 float x, float z, float* T, float* tau, float* slow_c, float* risti_c, float* geom /* ri dnx dnz cell */)
{
    GridDesc g; make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    SourceDesc s; if (make_source(g, x, z, s) != 0) return -1;
    const size_t nc = (size_t)g.nnx * g.nnz, nr = (size_t)s.rnx * s.rnz;
    std::vector<float> velv((size_t)nx * ny), cbasis(4 * (gd + 1)), rbasis(4 * (gd * kSgdl + 1));
    for (int k = 0; k < nx * ny; ++k) velv[k] = (float)pv[k];
    basis_table(gd, cbasis.data()); basis_table(gd * kSgdl, rbasis.data());
    std::vector<float> slow_r(nr), T_r(nr, kInf), tau_r(nr, kInf), risti_r(kRefMax), vcorner(4);
    float hmin = 1e30f;
    for (int ix = 1; ix <= g.nnx; ++ix) for (int iz = 1; iz <= g.nnz; ++iz) { float sl = 1.0f / coarse_velocity(g, velv.data(), cbasis.data(), iz, ix); slow_c[(size_t)(ix - 1) * g.nnz + (iz - 1)] = sl; if (sl < hmin) hmin = sl; }
    risti_table(g.gox, g.dnx, g.earth, g.nnx, risti_c);
    risti_table(s.rgox, s.rdnx, g.earth, s.rnx, risti_r.data());
    for (int lx = 1; lx <= s.rnx; ++lx) for (int kz = 1; kz <= s.rnz; ++kz) {
        const float v = refined_velocity(g, s, velv.data(), rbasis.data(), kz, lx);
        slow_r[(size_t)(lx - 1) * s.rnz + (kz - 1)] = 1.0f / v;
        if ((lx == s.isx_r || lx == s.isx_r + 1) && (kz == s.isz_r || kz == s.isz_r + 1)) vcorner[(lx - s.isx_r) * 2 + (kz - s.isz_r)] = v; }
    std::vector<int16_t> rst(kRWin * kRWin), cst((size_t)kCWinMax * kCWinMax); std::vector<int8_t> S_r(nr), cinit((size_t)kCWinMax * kCWinMax);
    std::vector<int32_t> heap(kHeapCap), flags(2, 0);
    SourceScratch w; w.slow_r = slow_r.data(); w.T_r = T_r.data(); w.tau_r = tau_r.data(); w.S_r = S_r.data(); w.risti_r = risti_r.data();
    w.vcorner = vcorner.data(); w.rst = rst.data(); w.cst = cst.data(); w.cinit = cinit.data(); w.heap = heap.data(); w.flags = flags.data();
    const int ended = refined_startup(g, s, w); refined_encode(s, w, ended);
    if (!ended) { Field fr = { s.rnx, s.rnz, T_r.data(), tau_r.data(), slow_r.data(), risti_r.data(), g.earth, s.rdnx, s.rdnz }; fixed_point(fr); }
    uint64_t rstar = ~0ull; int ez = 0, ex = 0;
    if (!ended) for (int ix = 1; ix <= s.rnx; ++ix) for (int iz = 1; iz <= s.rnz; ++iz) if (is_open_edge(s, iz, ix)) {
        const size_t id = (size_t)(ix - 1) * s.rnz + (iz - 1); if (!(t_value(T_r[id]) < kInf)) continue;
        const uint64_t r = accept_rank(T_r[id], tau_r[id]); if (r < rstar) { rstar = r; ez = iz; ex = ix; } }
    std::vector<float> Tfin(nr);
    for (int ix = 1; ix <= s.rnx; ++ix) for (int iz = 1; iz <= s.rnz; ++iz) { const size_t id = (size_t)(ix - 1) * s.rnz + (iz - 1); S_r[id] = (int8_t)handoff_node(g, s, w, ended, rstar, ez, ex, iz, ix, &Tfin[id]); }
    for (size_t k = 0; k < nc; ++k) { T[k] = kInf; tau[k] = kInf; }
    for (int q = 0; q < s.cwnx * s.cwnz; ++q) cst[q] = -1;
    auto cs = [&](int iz, int ix) -> int16_t& { return cst[(size_t)(ix - 1 - s.cwx0) * s.cwnz + (iz - 1 - s.cwz0)]; };
    for (int k = 1; k <= s.rnz; k += kSgdl) for (int l = 1; l <= s.rnx; l += kSgdl) { const int cz = s.vnt + (k - 1) / kSgdl, cx = s.vnl + (l - 1) / kSgdl;
        const size_t id = (size_t)(l - 1) * s.rnz + (k - 1); cs(cz, cx) = S_r[id]; if (S_r[id] >= 0) T[(size_t)(cx - 1) * g.nnz + (cz - 1)] = Tfin[id]; }
    auto far = [&](int iz, int ix) { if (ix < 1 || ix > g.nnx || iz < 1 || iz > g.nnz) return false;
        if (!(iz > s.cwz0 && iz <= s.cwz0 + s.cwnz && ix > s.cwx0 && ix <= s.cwx0 + s.cwnx)) return true; return cs(iz, ix) == -1; };
    for (int ix = s.vnl; ix <= s.vnr; ++ix) for (int iz = s.vnt; iz <= s.vnb; ++iz)
        if (cs(iz, ix) == 0 && (far(iz - 1, ix) || far(iz + 1, ix) || far(iz, ix - 1) || far(iz, ix + 1))) cs(iz, ix) = 1;
    coarse_band_march(g, s, w, T, tau, slow_c, risti_c);
    geom[0] = g.earth; geom[1] = g.dnx; geom[2] = g.dnz; geom[3] = min_cell_km(g) * hmin;
    return 0;
}
