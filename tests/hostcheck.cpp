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
#include "../dsurftomo_amd/csrc/ray_core.h"
#include "solve_node_walk_ref.h"
#include "../dsurftomo_amd/csrc/exact_march.h"

using namespace dsa;

namespace {

struct Field {
    int nnx, nnz, nbz;
    Rec* F;                 // tiled records
    const float* slow;      // tiled slowness
    const float* risti;
    float ri, dnx, dnz;
};

inline int ridx(const Field& f, int iz, int ix) { return rec_index(f.nbz, iz - 1, ix - 1); }   // 1-based

Hood load_hood(const Field& f, int iz, int ix)
{
    Hood h;
    const int nz[4] = { iz, iz, iz - 1, iz + 1 }, nx[4] = { ix - 1, ix + 1, ix, ix };
    const int oz[4] = { iz, iz, iz - 2, iz + 2 }, ox[4] = { ix - 2, ix + 2, ix, ix };
    for (int q = 0; q < 4; ++q) {
        h.in[q] = nx[q] >= 1 && nx[q] <= f.nnx && nz[q] >= 1 && nz[q] <= f.nnz;
        h.in_outer[q] = ox[q] >= 1 && ox[q] <= f.nnx && oz[q] >= 1 && oz[q] <= f.nnz;
        const Rec a = h.in[q] ? f.F[ridx(f, nz[q], nx[q])] : Rec{ kInf, kInf };
        const Rec b = h.in_outer[q] ? f.F[ridx(f, oz[q], ox[q])] : Rec{ kInf, kInf };
        h.near_[q] = a.T; h.near_tau[q] = a.tau; h.outer[q] = b.T; h.outer_tau[q] = b.tau;
    }
    return h;
}

// tiled <-> row-major helpers
std::vector<Rec> tile_fill(int nnx, int nnz)
{
    return std::vector<Rec>((size_t)tiles_of(nnx) * tiles_of(nnz) * kTileRecs, Rec{ kInf, kInf });
}
void untile(int nnx, int nnz, const Rec* F, float* T, float* tau)
{
    const int nbz = tiles_of(nnz);
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const Rec r = F[rec_index(nbz, iz, ix)];
            if (T) T[(size_t)ix * nnz + iz] = r.T;
            if (tau) tau[(size_t)ix * nnz + iz] = r.tau;
        }
}

long fixed_point(Field& f)
{
    const size_t n = (size_t)f.nnx * f.nnz;
    std::vector<int> qx(n * 2 + 64), qz(n * 2 + 64);
    std::vector<unsigned char> inq(n, 0);
    size_t head = 0, tail = 0, cap = qx.size();
    long evals = 0;
    auto push = [&](int iz, int ix) {
        if (ix < 1 || ix > f.nnx || iz < 1 || iz > f.nnz) return;
        const size_t id = (size_t)(ix - 1) * f.nnz + (iz - 1);
        if (t_pinned(f.F[ridx(f, iz, ix)].T) || inq[id]) return;
        inq[id] = 1; qx[tail] = ix; qz[tail] = iz; tail = (tail + 1) % cap;
    };
    for (int ix = 1; ix <= f.nnx; ++ix)
        for (int iz = 1; iz <= f.nnz; ++iz)
            if (t_pinned(f.F[ridx(f, iz, ix)].T)) { push(iz, ix - 1); push(iz, ix + 1); push(iz - 1, ix); push(iz + 1, ix); }
    while (head != tail) {
        const int ix = qx[head], iz = qz[head]; head = (head + 1) % cap;
        inq[(size_t)(ix - 1) * f.nnz + (iz - 1)] = 0;
        const Hood h = load_hood(f, iz, ix);
        const NodeGeom g = { f.ri, f.risti[ix - 1], f.dnx, f.dnz };
        Rec& r = f.F[ridx(f, iz, ix)];
        float k;
        const float c = solve_node(h, f.slow[ridx(f, iz, ix)], g, &k);
        ++evals;
        if (std::memcmp(&c, &r.T, 4) != 0 || std::memcmp(&k, &r.tau, 4) != 0) {
            r.T = c; r.tau = k;
            push(iz, ix - 1); push(iz, ix + 1); push(iz - 1, ix); push(iz + 1, ix);
            push(iz, ix - 2); push(iz, ix + 2); push(iz - 2, ix); push(iz + 2, ix);
        }
        if (evals > 400L * (long)n) return -1;
    }
    return evals;
}

// everything up to (not including) the coarse fixed-point solve, on tiled storage
struct Prepared {
    GridDesc g; SourceDesc s;
    std::vector<Rec> F_c, F_r;
    std::vector<float> slow_c, slow_r, risti_c, risti_r, Tfin;
    std::vector<int8_t> S_r;
    std::vector<int16_t> cst;
    int ended = 0, err = 0; long evals_r = 0; float hmin = 0;
};

int prepare(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv, float x, float z,
            Prepared& P, float* inj_t, int* inj_s)
{
    GridDesc& g = P.g; SourceDesc& s = P.s;
    make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    if (make_source(g, x, z, s) != 0) return -1;
    const size_t nr = (size_t)s.rnx * s.rnz;
    std::vector<float> velv((size_t)nx * ny), cbasis(4 * (gd + 1)), rbasis(4 * (gd * kSgdl + 1)), vcorner(4);
    for (int k = 0; k < nx * ny; ++k) velv[k] = (float)pv[k];
    basis_table(gd, cbasis.data());
    basis_table(gd * kSgdl, rbasis.data());
    P.F_c = tile_fill(g.nnx, g.nnz); P.F_r = tile_fill(s.rnx, s.rnz);
    P.slow_c.assign(P.F_c.size(), 1.0f); P.slow_r.assign(P.F_r.size(), 1.0f);
    P.risti_c.resize(g.nnx); P.risti_r.resize(kRefMax);
    P.hmin = 1e30f;
    for (int ix = 1; ix <= g.nnx; ++ix)
        for (int iz = 1; iz <= g.nnz; ++iz) {
            const float sl = 1.0f / coarse_velocity(g, velv.data(), cbasis.data(), iz, ix);
            P.slow_c[rec_index(g.nbz, iz - 1, ix - 1)] = sl;
            if (sl < P.hmin) P.hmin = sl;
        }
    risti_table(g.gox, g.dnx, g.earth, g.nnx, P.risti_c.data());
    risti_table(s.rgox, s.rdnx, g.earth, s.rnx, P.risti_r.data());
    for (int lx = 1; lx <= s.rnx; ++lx)
        for (int kz = 1; kz <= s.rnz; ++kz) {
            const float v = refined_velocity(g, s, velv.data(), rbasis.data(), kz, lx);
            P.slow_r[rec_index(s.nbz_r, kz - 1, lx - 1)] = 1.0f / v;
            if ((lx == s.isx_r || lx == s.isx_r + 1) && (kz == s.isz_r || kz == s.isz_r + 1))
                vcorner[(lx - s.isx_r) * 2 + (kz - s.isz_r)] = v;
        }
    std::vector<int16_t> rst(kRWin * kRWin);
    P.cst.assign((size_t)kCWinMax * kCWinMax, -1);
    P.S_r.assign(nr, -1);
    std::vector<int8_t> cinit((size_t)kCWinMax * kCWinMax);
    std::vector<int32_t> heap(kHeapCap), flags(4, 0);
    SourceScratch w;
    w.slow_r = P.slow_r.data(); w.F_r = P.F_r.data(); w.S_r = P.S_r.data(); w.risti_r = P.risti_r.data();
    w.vcorner = vcorner.data(); w.rst = rst.data(); w.cst = P.cst.data(); w.cinit = cinit.data();
    w.heap = heap.data(); w.flags = flags.data();

    const int ended = refined_startup(g, s, w);
    refined_encode(s, w, ended);
    P.ended = ended;
    if (!ended) {
        Field fr = { s.rnx, s.rnz, s.nbz_r, P.F_r.data(), P.slow_r.data(), P.risti_r.data(), g.earth, s.rdnx, s.rdnz };
        P.evals_r = fixed_point(fr);
    }
    // first open-edge node in acceptance order (scan order ix outer, iz inner breaks exact ties)
    uint64_t rstar = ~0ull; int ez = 0, ex = 0;
    if (!ended)
        for (int ix = 1; ix <= s.rnx; ++ix)
            for (int iz = 1; iz <= s.rnz; ++iz)
                if (is_open_edge(s, iz, ix)) {
                    const Rec r = P.F_r[rec_index(s.nbz_r, iz - 1, ix - 1)];
                    if (!(t_value(r.T) < kInf)) continue;
                    const uint64_t rk = accept_rank(r.T, r.tau);
                    if (rk < rstar) { rstar = rk; ez = iz; ex = ix; }
                }
    P.Tfin.assign(nr, kInf);
    for (int ix = 1; ix <= s.rnx; ++ix)
        for (int iz = 1; iz <= s.rnz; ++iz) {
            const size_t id = (size_t)(ix - 1) * s.rnz + (iz - 1);
            P.S_r[id] = (int8_t)handoff_node(g, s, w, ended, rstar, ez, ex, iz, ix, &P.Tfin[id]);
        }
    // injection + band promotion into the coarse window (records of the window only, like on the device)
    std::vector<Rec> W((size_t)s.cwnx * s.cwnz, Rec{ kInf, kInf });
    auto cs = [&](int iz, int ix) -> int16_t& { return P.cst[(size_t)(ix - 1 - s.cwx0) * s.cwnz + (iz - 1 - s.cwz0)]; };
    for (int k = 1; k <= s.rnz; k += kSgdl)
        for (int l = 1; l <= s.rnx; l += kSgdl) {
            const int cz = s.vnt + (k - 1) / kSgdl, cx = s.vnl + (l - 1) / kSgdl;
            const size_t id = (size_t)(l - 1) * s.rnz + (k - 1);
            cs(cz, cx) = P.S_r[id];
            if (P.S_r[id] >= 0) W[(size_t)(cx - 1 - s.cwx0) * s.cwnz + (cz - 1 - s.cwz0)].T = P.Tfin[id];
        }
    auto far = [&](int iz, int ix) {
        if (ix < 1 || ix > g.nnx || iz < 1 || iz > g.nnz) return false;
        if (!(iz > s.cwz0 && iz <= s.cwz0 + s.cwnz && ix > s.cwx0 && ix <= s.cwx0 + s.cwnx)) return true;
        return cs(iz, ix) == -1;
    };
    for (int ix = s.vnl; ix <= s.vnr; ++ix)
        for (int iz = s.vnt; iz <= s.vnb; ++iz)
            if (cs(iz, ix) == 0 && (far(iz - 1, ix) || far(iz + 1, ix) || far(iz, ix - 1) || far(iz, ix + 1))) cs(iz, ix) = 1;
    if (inj_t) {
        for (size_t k = 0; k < (size_t)g.nnx * g.nnz; ++k) inj_t[k] = kInf;
        for (int ix = s.cwx0 + 1; ix <= s.cwx0 + s.cwnx; ++ix)
            for (int iz = s.cwz0 + 1; iz <= s.cwz0 + s.cwnz; ++iz) inj_t[(size_t)(ix - 1) * g.nnz + (iz - 1)] = W[(size_t)(ix - 1 - s.cwx0) * s.cwnz + (iz - 1 - s.cwz0)].T;
    }
    if (inj_s) {
        for (size_t k = 0; k < (size_t)g.nnx * g.nnz; ++k) inj_s[k] = -1;
        for (int ix = s.cwx0 + 1; ix <= s.cwx0 + s.cwnx; ++ix)
            for (int iz = s.cwz0 + 1; iz <= s.cwz0 + s.cwnz; ++iz) inj_s[(size_t)(ix - 1) * g.nnz + (iz - 1)] = cs(iz, ix);
    }
    coarse_band_march(g, s, w, W.data(), P.slow_c.data(), P.risti_c.data());
    export_window_records(g, s, W.data(), P.F_c.data());
    P.err = flags[1];
    return 0;
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
    Prepared P;
    if (prepare(nx, ny, goxd, gozd, dvxd, dvzd, gd, pv, x, z, P, inj_t, inj_s) != 0) return -1;
    const GridDesc& g = P.g; const SourceDesc& s = P.s;
    const size_t nr = (size_t)s.rnx * s.rnz;
    if (ttnr_out) std::memcpy(ttnr_out, P.Tfin.data(), 4 * nr);
    if (nstsr_out) for (size_t k = 0; k < nr; ++k) nstsr_out[k] = P.S_r[k];
    Field fc = { g.nnx, g.nnz, g.nbz, P.F_c.data(), P.slow_c.data(), P.risti_c.data(), g.earth, g.dnx, g.dnz };
    stats[3] = fixed_point(fc);
    untile(g.nnx, g.nnz, P.F_c.data(), ttn, nullptr);
    for (size_t k = 0; k < (size_t)g.nnx * g.nnz; ++k) ttn[k] = t_value(ttn[k]);
    stats[0] = P.ended; stats[1] = P.err; stats[2] = P.evals_r;
    if (box_out) { box_out[0] = s.vnl; box_out[1] = s.vnr; box_out[2] = s.vnt; box_out[3] = s.vnb; box_out[4] = s.rnx; box_out[5] = s.rnz; }
    return 0;
}

// trace_ray (ray_core.h) on oracle-format fields: ttn / veln x-major (nnz fastest), refined
// snapshot ttnr / nstsr x-major with leading dimension rnz.  fdm comes back in the oracle's layout
// fdm[vx * (nvz+2) + vz].
int hc_trace_ray(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const float* veln,
                 const float* ttn, const float* ttnr, const int* nstsr, float sx, float sz, float rx, float rz,
                 float* fdm, int* flags, int* nsteps)
{
    GridDesc g; make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    SourceDesc s;
    if (make_source(g, sx, sz, s) != 0) return -2;
    std::vector<float> F((size_t)g.nbx * g.nbz * kTileRecs, kInf);          // compact coarse field: one float per node, tiled
    for (int ix = 0; ix < g.nnx; ++ix)
        for (int iz = 0; iz < g.nnz; ++iz) F[rec_index(g.nbz, iz, ix)] = ttn[(size_t)ix * g.nnz + iz];
    std::vector<int8_t> S((size_t)s.rnx * s.rnz);
    for (size_t k = 0; k < S.size(); ++k) S[k] = (int8_t)(nstsr[k] > 0 ? 1 : (nstsr[k] < 0 ? -1 : 0));
    std::vector<float> slab((size_t)(g.nvx + 2) * (g.nvz + 2), 0.0f);
    RayFields f{ F.data(), veln, ttnr, S.data() };
    *flags = 0;
    const int rc = trace_ray(g, s, f, rx, rz, min_cell_km(g), slab.data(), flags, nsteps);
    for (int vx = 0; vx < g.nvx + 2; ++vx)
        for (int vz = 0; vz < g.nvz + 2; ++vz) fdm[(size_t)vx * (g.nvz + 2) + vz] = slab[(size_t)vz * (g.nvx + 2) + vx];
    return rc;
}

// the same with `lanes` lanes per ray (ray_core.h: PatchAcc<4>): the four lanes' traces one after the other on the one slab -- a slab row has
// one owner, so the order of the lanes does not matter
int hc_trace_ray_lanes(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const float* veln,
                       const float* ttn, const float* ttnr, const int* nstsr, float sx, float sz, float rx, float rz,
                       float* fdm, int* flags, int* nsteps, int lanes)
{
    if (lanes == 1) return hc_trace_ray(nx, ny, goxd, gozd, dvxd, dvzd, gd, veln, ttn, ttnr, nstsr, sx, sz, rx, rz, fdm, flags, nsteps);
    GridDesc g; make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    SourceDesc s;
    if (make_source(g, sx, sz, s) != 0) return -2;
    std::vector<float> F((size_t)g.nbx * g.nbz * kTileRecs, kInf);
    for (int ix = 0; ix < g.nnx; ++ix)
        for (int iz = 0; iz < g.nnz; ++iz) F[rec_index(g.nbz, iz, ix)] = ttn[(size_t)ix * g.nnz + iz];
    std::vector<int8_t> S((size_t)s.rnx * s.rnz);
    for (size_t k = 0; k < S.size(); ++k) S[k] = (int8_t)(nstsr[k] > 0 ? 1 : (nstsr[k] < 0 ? -1 : 0));
    std::vector<float> slab((size_t)(g.nvx + 2) * (g.nvz + 2), 0.0f);
    RayFields f{ F.data(), veln, ttnr, S.data() };
    int rc = 0;
    for (int sub = 3; sub >= 0; --sub) {
        *flags = 0;
        rc = trace_ray<4>(g, s, f, rx, rz, min_cell_km(g), slab.data(), flags, nsteps, nullptr, sub);
    }
    for (int vx = 0; vx < g.nvx + 2; ++vx)
        for (int vz = 0; vz < g.nvz + 2; ++vz) fdm[(size_t)vx * (g.nvz + 2) + vz] = slab[(size_t)vz * (g.nvx + 2) + vx];
    return rc;
}

float hc_sinf(float x) { return sinf_libm(x); }

// count of fp32 values in [lo, hi] where sinf_libm differs from this machine's libm sinf
long hc_sinf_sweep(float lo, float hi)
{
    long bad = 0;
    for (float v = lo; v <= hi; v = nextafterf(v, 1e30f)) {
        const float a = sinf(v), b = sinf_libm(v);
        if (std::memcmp(&a, &b, 4) != 0) ++bad;
    }
    return bad;
}

void hc_brocher_chain(float v, int shallow, float* a, float* r) { brocher_chain(v, shallow != 0, a, r); }

// dispersion_core.h on the host: pv(c, k) and, with kernels, sen_q(c, k, i) for a (nz, ncol) Vs
// model -- the arithmetic of disp_kernels.hip run serially with this machine's libm
int hc_depthkernel(int ncol, int nz, const float* vels, const float* depz, float minthk, int iwave, int igr, int kmax,
                   const double* t, int with_kernels, double* pv, double* sen_vs, double* sen_vp, double* sen_rho)
{
    LayerGeom G;
    if (make_layer_geom(nz, depz, minthk, G) != 0) return -1;
    std::vector<float> ws((size_t)4 * G.rmax);
    const int npert = with_kernels ? 1 + 6 * nz : 1;
    std::vector<double> curves((size_t)npert * kmax);
    double* out[3] = { sen_vs, sen_vp, sen_rho };
    for (int c = 0; c < ncol; ++c) {
        for (int p = 0; p < npert; ++p) {
            float vs[64], vp[64], rho[64];
            for (int k = 0; k < nz; ++k) { vs[k] = vels[(size_t)k * ncol + c]; brocher_vp_rho(vs[k], &vp[k], &rho[k]); }
            if (p > 0) {
                const int idx = p - 1, i = idx / 6, q = (idx % 6) >> 1, s = idx & 1;
                float* arr = q == 0 ? vs : (q == 1 ? vp : rho);
                const float base = arr[i], dln = 0.01f;
                arr[i] = s ? base + 0.5f * dln * base : base - 0.5f * dln * base;
            }
            Layers m;
            m.d = ws.data(); m.a = m.d + G.rmax; m.b = m.a + G.rmax; m.rho = m.b + G.rmax; m.stride = 1;
            if (iwave == 1) { build_layers<1>(G, vs, vp, rho, m); dispersion_curve<1>(m, igr, kmax, t, &curves[(size_t)p * kmax], 1); }
            else { build_layers<2>(G, vs, vp, rho, m); dispersion_curve<2>(m, igr, kmax, t, &curves[(size_t)p * kmax], 1); }
        }
        for (int k = 0; k < kmax; ++k) pv[(size_t)k * ncol + c] = curves[k];
        if (!with_kernels) continue;
        for (int i = 0; i < nz; ++i) {
            float base[3];
            base[0] = vels[(size_t)i * ncol + c];
            brocher_vp_rho(base[0], &base[1], &base[2]);
            for (int q = 0; q < 3; ++q)
                for (int k = 0; k < kmax; ++k) {
                    const double cg1 = curves[(size_t)(1 + 6 * i + 2 * q) * kmax + k], cg2 = curves[(size_t)(2 + 6 * i + 2 * q) * kmax + k];
                    out[q][((size_t)i * kmax + k) * ncol + c] = (cg2 - cg1) / (double)(0.01f * base[q]);
                }
        }
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Emulation of the device schedule of fim_kernel.hip (same routing, same theta update, same cycle
// rule; the ready set is evaluated from the old states), used to study convergence on the CPU.
// Interface arrays are row-major (nnz fastest); storage inside is tiled like on the device.
// mode 0: all ready nodes at once; mode 1: two sub-passes by node parity (even first).
extern "C" { int g_prune = 1; long g_regular_stats[8] = {}; long* hc_regular_stats() { return g_regular_stats; } }
extern "C" long hc_device_schedule(int nnx, int nnz, float* Tio, float* tauio, const float* slow_rm, const float* risti,
                                   float ri, float dnx, float dnz, float window, int mode, int max_rounds,
                                   long* out /* rounds, evals, last list size, freezes */, int* cyc_ids, int ncyc)
{
    const int nbz = tiles_of(nnz);
    std::vector<Rec> F = tile_fill(nnx, nnz);
    std::vector<float> slow(F.size(), 1.0f);
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const int id = rec_index(nbz, iz, ix);
            F[id] = Rec{ Tio[(size_t)ix * nnz + iz], tauio[(size_t)ix * nnz + iz] };
            slow[id] = slow_rm[(size_t)ix * nnz + iz];
        }
    Field f = { nnx, nnz, nbz, F.data(), slow.data(), risti, ri, dnx, dnz };
    const size_t n = F.size();
    std::vector<int> cur, next, ready;
    std::vector<unsigned char> queued(n, 0);
    auto act = [&](int iz0, int ix0) {
        if (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) return;
        const int id = rec_index(nbz, iz0, ix0);
        if (t_pinned(F[id].T) || queued[id]) return;
        queued[id] = 1; next.push_back(id);
    };
    for (int ix = 0; ix < nnx; ++ix) for (int iz = 0; iz < nnz; ++iz)
        if (t_pinned(F[rec_index(nbz, iz, ix)].T)) { act(iz, ix - 1); act(iz, ix + 1); act(iz - 1, ix); act(iz + 1, ix); }
    cur.swap(next);
    float theta = kInf; long rounds = 0, evals = 0;
    std::vector<float> nT, nK;
    float best_tmin = -kInf, freeze = -kInf; int stall = 0; long freezes = 0;
    unsigned hist[4] = { 1u, 2u, 3u, 4u }, hsh = 0u;
    auto tv = [&](int iz0, int ix0) { return (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) ? kInf : tau_value(F[rec_index(nbz, iz0, ix0)].tau); };
    while (!cur.empty()) {
        float tmin = kInf; ready.clear();
        for (int id : cur) {
            if (tau_value(F[id].tau) < freeze) { queued[id] = 0; continue; }       // accepted below the freeze horizon: final
            int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
            const float lb = fminf(fminf(tv(iz0, ix0 - 1), tv(iz0, ix0 + 1)), fminf(tv(iz0 - 1, ix0), tv(iz0 + 1, ix0)));
            if (!(theta < kInf) || lb < theta) { ready.push_back(id); queued[id] = 0; }
            else { next.push_back(id); tmin = fminf(tmin, lb); }
        }
        for (int pass = 0; pass < (mode == 1 ? 2 : 1); ++pass) {
            std::vector<int> sub;
            for (int id : ready) { int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0); if (mode == 0 || ((ix0 + iz0) & 1) == pass) sub.push_back(id); }
            nT.resize(sub.size()); nK.resize(sub.size());
            for (size_t k = 0; k < sub.size(); ++k) {
                int iz0, ix0; rec_coords(nbz, sub[k], &iz0, &ix0);
                const Hood h = load_hood(f, iz0 + 1, ix0 + 1); const NodeGeom g = { ri, risti[ix0], dnx, dnz };
                nT[k] = solve_node(h, slow[sub[k]], g, &nK[k]); ++evals;
                // (round 4) the regular-neighbourhood form of the walk beside it: where it applies and says ok, the same bits
                {
                    bool regular = true;
                    float tn[4], t2[4];
                    for (int q = 0; q < 4; ++q) {
                        regular = regular && h.in[q] && !std::signbit(h.near_[q]) && std::memcmp(&h.near_[q], &h.near_tau[q], 4) == 0;
                        if (h.in_outer[q]) regular = regular && !std::signbit(h.outer[q]) && std::memcmp(&h.outer[q], &h.outer_tau[q], 4) == 0;
                        tn[q] = h.near_[q]; t2[q] = h.in_outer[q] ? h.outer[q] : kInf;
                    }
                    ++g_regular_stats[0];
                    if (regular) {
                        ++g_regular_stats[1];
                        float kr; bool ok;
                        const float cr = solve_regular(tn, t2, slow[sub[k]], g, &kr, &ok);
                        if (ok) {
                            ++g_regular_stats[2];
                            if (std::memcmp(&cr, &nT[k], 4) || std::memcmp(&kr, &nK[k], 4)) ++g_regular_stats[3];
                            if (std::memcmp(&cr, &kr, 4)) ++g_regular_stats[4];          // (ok but not causal: the caller's slow path)
                        }
                    }
                }
            }
            for (size_t k = 0; k < sub.size(); ++k) {
                const int id = sub[k];
                if (std::memcmp(&nT[k], &F[id].T, 4) || std::memcmp(&nK[k], &F[id].tau, 4)) {
                    const float t_lo = fminf(t_value(F[id].T), nT[k]), k_lo = fminf(tau_value(F[id].tau), nK[k]);
                    F[id].T = nT[k]; F[id].tau = nK[k];
                    { unsigned a, b; std::memcpy(&a, &nT[k], 4); std::memcpy(&b, &nK[k], 4); hsh += ((unsigned)id * 2654435761u) ^ (a * 40503u) ^ (b * 2246822519u); }
                    int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
                    // dependents that can be affected (same tests as fim_kernel.hip, see there)
                    const int dz[4] = { 0, 0, -1, 1 }, dx[4] = { -1, 1, 0, 0 };
                    for (int q = 0; q < 4; ++q) {
                        const int yz = iz0 + dz[q], yx = ix0 + dx[q], zz = iz0 + 2 * dz[q], zx = ix0 + 2 * dx[q];
                        if (yx < 0 || yx >= nnx || yz < 0 || yz >= nnz) continue;
                        const Rec y = F[rec_index(nbz, yz, yx)];
                        if (!g_prune || k_lo <= tau_value(y.tau)) act(yz, yx);
                        if (zx < 0 || zx >= nnx || zz < 0 || zz >= nnz) continue;
                        if (!(tau_value(y.tau) < kInf)) continue;          // through a reached in-between node only
                        const Rec zr = F[rec_index(nbz, zz, zx)];
                        if (!g_prune || (t_value(y.T) > t_lo && k_lo < tau_value(zr.tau))) act(zz, zx);
                    }
                    tmin = fminf(tmin, nK[k]);
                }
            }
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
    untile(nnx, nnz, F.data(), Tio, tauio);
    out[0] = rounds; out[1] = evals; out[2] = (long)cur.size(); out[3] = freezes;
    for (int k = 0; k < (ncyc < 0 ? -ncyc : ncyc); ++k) cyc_ids[k] = k < (int)cur.size() ? cur[k] : -1;
    return cur.empty() ? 0 : -1;
}

// set-up helper: coarse problem of one source after the serial stages, row-major outputs
extern "C" int hc_coarse_problem(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv,
                                 float x, float z, float* T, float* tau, float* slow_c, float* risti_c, float* geom /* ri dnx dnz cell */)
{
    Prepared P;
    if (prepare(nx, ny, goxd, gozd, dvxd, dvzd, gd, pv, x, z, P, nullptr, nullptr) != 0) return -1;
    const GridDesc& g = P.g;
    untile(g.nnx, g.nnz, P.F_c.data(), T, tau);
    for (int ix = 0; ix < g.nnx; ++ix)
        for (int iz = 0; iz < g.nnz; ++iz) slow_c[(size_t)ix * g.nnz + iz] = P.slow_c[rec_index(g.nbz, iz, ix)];
    std::memcpy(risti_c, P.risti_c.data(), 4 * (size_t)g.nnx);
    geom[0] = g.earth; geom[1] = g.dnx; geom[2] = g.dnz; geom[3] = min_cell_km(g) * P.hmin;
    return 0;
}

// The product's solve_node against round 1's step-by-step form (tests/solve_node_walk_ref.h) on n random neighbourhoods: realistic
// fronts plus the awkward cases (ties, unreached and pinned neighbours, grid edges, outer nodes accepted at time 0 or late).
// Returns the number of neighbourhoods whose (T, tau) bits differ; `stat[0..3]` = evaluations that ended with 0, 1, 2, 3+ walk steps' worth of
// finite results (coverage only).
extern "C" long hc_solve_node_compare(unsigned long long seed, long n, long* stat)
{
    unsigned long long st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    auto uni = [&]() { return (float)(rnd() & 0xffffff) / 16777216.0f; };
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        NodeGeom g;
        g.ri = 6371.0f - 40.0f * uni();
        g.risti = g.ri * sinf(0.3f + 1.2f * uni());
        g.dnx = 5e-5f + 6e-4f * uni();
        g.dnz = (rnd() & 3) ? g.dnx * (0.7f + 0.6f * uni()) : g.dnx;
        const float slown = 1.0f / (1.5f + 3.5f * uni());
        const float hx = g.ri * g.dnx * slown, hz = g.risti * g.dnz * slown;      // one-cell travel times
        const float t0 = (rnd() & 7) ? 300.0f * uni() : 2.0f * uni();
        const int mode = rnd() & 15;
        Hood h;
        for (int q = 0; q < 4; ++q) {
            const float hq = q < 2 ? hx : hz;
            const unsigned r = rnd();
            h.in[q] = (r & 31) != 0;
            h.in_outer[q] = h.in[q] && ((r >> 5) & 15) != 0;
            float t = t0 + hq * (2.4f * uni() - 1.2f);
            if (mode == 1) t = t0;                                               // all equal
            if (mode == 2 && (q & 1)) t = t0 + hq * 0.25f;                       // pairs equal
            if (t < 0.0f) t = 0.0f;
            float tau = ((r >> 9) & 3) ? t : t + hq * 0.3f * uni();              // accepted late now and then
            if (((r >> 11) & 7) == 0) { t = kInf; tau = kInf; }                  // not reached
            const bool pin = ((r >> 14) & 15) == 0 && t < kInf;
            h.near_[q] = pin ? -t : t;
            h.near_tau[q] = (((r >> 18) & 7) == 0 && t < kInf) ? -tau : tau;      // the list variant's queued bit: must be ignored
            float o = t + hq * (1.6f * uni() - 1.1f);
            if (mode == 3) o = t;                                                // tn == t2: first order
            if (o < 0.0f) o = 0.0f;
            float otau = ((r >> 21) & 3) ? o : o + hq * 0.5f * uni();
            if (((r >> 23) & 15) == 0) otau = 0.0f;                              // alive before any march
            if (((r >> 27) & 7) == 0 || t == kInf) { o = kInf; otau = kInf; }
            const bool opin = ((r >> 30) & 1) && ((r >> 14) & 3) == 0 && o < kInf;
            h.outer[q] = h.in_outer[q] ? (opin ? -o : o) : kInf;
            h.outer_tau[q] = h.in_outer[q] ? otau : kInf;
            if (!h.in[q]) { h.near_[q] = kInf; h.near_tau[q] = kInf; }
        }
        float ka, kb;
        const float a = solve_node(h, slown, g, &ka), b = solve_node_walk_ref(h, slown, g, &kb);
        float kc, tie;
        const float c3 = solve_node_t<true>(h, slown, g, &kc, &tie);     // the detector must not change the result
        if (std::memcmp(&a, &c3, 4) != 0 || std::memcmp(&ka, &kc, 4) != 0) { if (bad < 5) std::fprintf(stderr, "solve_node_t<true> differs at case %ld\n", i); ++bad; }
        if (stat && tie >= 0.0f) stat[5] += 1;
        if (stat && tie > 0.0f) stat[6] += 1;
        if (std::memcmp(&a, &b, 4) != 0 || std::memcmp(&ka, &kb, 4) != 0) {
            if (bad < 5)
                std::fprintf(stderr, "solve_node differs at case %ld: T %.9g vs %.9g, tau %.9g vs %.9g\n", i, (double)a, (double)b, (double)ka, (double)kb);
            ++bad;
        }
        if (stat) {
            int alive = 0;
            for (int q = 0; q < 4; ++q) alive += h.in[q] && tau_value(h.near_tau[q]) < kb + 0.0f && tau_value(h.near_tau[q]) < kInf;
            stat[a < kInf ? (alive > 3 ? 3 : alive) : 4] += 1;
        }
    }
    return bad;
}


// solve_regular (round 4: the walk written out for neighbourhoods without pinned / late-accepted / missing near neighbours) against
// solve_node on n random regular neighbourhoods.  Returns the number of cases where it said ok and its (T, tau) bits differ;
// stat[0] = cases it accepted, stat[1] = those that stopped after one neighbour (coverage).
extern "C" long hc_solve_regular_compare(unsigned long long seed, long n, long* stat)
{
    unsigned long long st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    auto uni = [&]() { return (float)(rnd() & 0xffffff) / 16777216.0f; };
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        NodeGeom g;
        g.ri = 6371.0f - 40.0f * uni();
        g.risti = g.ri * sinf(0.3f + 1.2f * uni());
        g.dnx = 5e-5f + 6e-4f * uni();
        g.dnz = (rnd() & 3) ? g.dnx * (0.7f + 0.6f * uni()) : g.dnx;
        const float slown = 1.0f / (1.5f + 3.5f * uni());
        const float hx = g.ri * g.dnx * slown, hz = g.risti * g.dnz * slown;
        const float t0 = (rnd() & 7) ? 300.0f * uni() : 2.0f * uni();
        const int mode = rnd() & 15;
        Hood h;
        float tn[4], t2[4];
        for (int q = 0; q < 4; ++q) {
            const float hq = q < 2 ? hx : hz;
            const unsigned r = rnd();
            h.in[q] = true;
            h.in_outer[q] = ((r >> 5) & 15) != 0;
            float t = t0 + hq * (2.4f * uni() - 1.2f);
            if (mode == 1) t = t0;                                               // all equal
            if (mode == 2 && (q & 1)) t = t0 + hq * 0.25f;                       // pairs equal
            if (mode == 4) t = t0 + (float)(rnd() % 3) * hq * 0.5f;             // ties across the directions
            if (t < 0.0f) t = 0.0f;
            if (((r >> 11) & 7) == 0) t = kInf;                                  // not reached
            float o = t + hq * (1.6f * uni() - 1.1f);
            if (mode == 3) o = t;                                                // tn == t2: first order
            if (o < 0.0f) o = 0.0f;
            if (((r >> 23) & 15) == 0) o = 0.0f;
            if (((r >> 27) & 7) == 0 || t == kInf) o = kInf;
            h.near_[q] = t; h.near_tau[q] = t;
            h.outer[q] = h.in_outer[q] ? o : kInf; h.outer_tau[q] = h.outer[q];
            tn[q] = t; t2[q] = h.outer[q];
        }
        float ka, kb; bool ok, tie = false;
        const float a = solve_node(h, slown, g, &ka);
        const float b = solve_regular(tn, t2, slown, g, &kb, &ok, &tie);
        if (!ok) continue;
        {   // the tie it reports is the tie solve_node_t<true> probes (the engine's detector routes such evaluations to that form)
            float kt, ti;
            (void)solve_node_t<true>(h, slown, g, &kt, &ti);
            if (stat) { stat[2] += tie ? 1 : 0; }
            if (tie != (ti >= 0.0f)) { if (bad < 5) std::fprintf(stderr, "solve_regular tie flag differs at case %ld: %d vs influence %g\n", i, (int)tie, (double)ti); ++bad; }
        }
        if (stat) { stat[0] += 1; if (!(ka > 0.0f) || std::memcmp(&a, &ka, 4) == 0) stat[1] += 0; }
        if (std::memcmp(&a, &b, 4) != 0 || std::memcmp(&ka, &kb, 4) != 0) {
            if (bad < 5) std::fprintf(stderr, "solve_regular differs at case %ld: T %.9g vs %.9g, tau %.9g vs %.9g\n", i, (double)a, (double)b, (double)ka, (double)kb);
            ++bad;
        }
    }
    return bad;
}

// Exact mode (csrc/exact_march.h) on the CPU: the CPU model of the device's march (same order of events per accept step, same quadrant
// arithmetic), sequenced like the device's launches
// (refined stage, snapshot, hand-off, coarse stage); `lcap` tree slots in the "LDS" part, the rest in the "global" part, so that
// the split is exercised.  Outputs row-major like hc_solve_source; returns 0, or the march's error code.
extern "C" int hc_exact_solve(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd, const double* pv, float x, float z,
                              int lcap, int gcap, float* T, float* Tr, int* Sr, int* box, long* stat)
{
    GridDesc g; SourceDesc s;
    make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    if (make_source(g, x, z, s) != 0) return -1;
    std::vector<float> velv((size_t)nx * ny), cbasis(4 * (gd + 1)), rbasis(4 * (gd * kSgdl + 1)), vcorner(4);
    for (int k = 0; k < nx * ny; ++k) velv[k] = (float)pv[k];
    basis_table(gd, cbasis.data());
    basis_table(gd * kSgdl, rbasis.data());
    const size_t nrc = (size_t)g.nbx * g.nbz * kTileRecs;
    std::vector<float> slow_c(nrc, 1.0f), slow_r(kRefRecs, 1.0f), risti_c(g.nnx), risti_r(kRefMax);
    for (int ix = 1; ix <= g.nnx; ++ix)
        for (int iz = 1; iz <= g.nnz; ++iz) slow_c[rec_index(g.nbz, iz - 1, ix - 1)] = 1.0f / coarse_velocity(g, velv.data(), cbasis.data(), iz, ix);
    risti_table(g.gox, g.dnx, g.earth, g.nnx, risti_c.data());
    risti_table(s.rgox, s.rdnx, g.earth, s.rnx, risti_r.data());
    for (int lx = 1; lx <= s.rnx; ++lx)
        for (int kz = 1; kz <= s.rnz; ++kz) {
            const float v = refined_velocity(g, s, velv.data(), rbasis.data(), kz, lx);
            slow_r[rec_index(s.nbz_r, kz - 1, lx - 1)] = 1.0f / v;
            if ((lx == s.isx_r || lx == s.isx_r + 1) && (kz == s.isz_r || kz == s.isz_r + 1)) vcorner[(lx - s.isx_r) * 2 + (kz - s.isz_r)] = v;
        }
    std::vector<XEntry> hl((size_t)lcap + 1), hg((size_t)gcap + 1);
    XMarch m;
    m.hl = hl.data(); m.lcap = lcap; m.hg = hg.data(); m.gcap = gcap;
    m.ntr = 0; m.error = 0; m.pops = 0u; m.ri = g.earth;
    std::vector<XRec> Fr(kRefRecs, XRec{ 0.0f, -1 });
    m.F = Fr.data(); m.slow = slow_r.data(); m.risti = risti_r.data();
    x_set_grid(m, s.nbz_r, s.rnx, s.rnz); m.dnx = s.rdnx; m.dnz = s.rdnz;
    x_refined_start(m, s, vcorner.data());
    x_march<true>(m, s);
    if (m.error) return m.error;
    if (stat) stat[0] = (long)m.pops;
    box[0] = s.vnl; box[1] = s.vnr; box[2] = s.vnt; box[3] = s.vnb; box[4] = s.rnx; box[5] = s.rnz;
    for (int ix = 0; ix < s.rnx; ++ix)
        for (int iz = 0; iz < s.rnz; ++iz) {
            const XRec r = Fr[rec_index(s.nbz_r, iz, ix)];
            Sr[(size_t)ix * s.rnz + iz] = r.st < 0 ? -1 : r.st == 0 ? 0 : 1;
            Tr[(size_t)ix * s.rnz + iz] = r.st >= 0 ? r.T : kInf;
        }
    const int bxn = (s.rnx - 1) / kSgdl + 1, bzn = (s.rnz - 1) / kSgdl + 1;
    std::vector<int> st(bxn * bzn), pr(bxn * bzn, 0);
    std::vector<float> sT(bxn * bzn);
    for (int q = 0; q < bxn * bzn; ++q) {
        const XRec r = Fr[rec_index(s.nbz_r, (q % bzn) * kSgdl, (q / bzn) * kSgdl)];
        st[q] = r.st < 0 ? -1 : r.st == 0 ? 0 : 1; sT[q] = r.T;
    }
    for (int q = 0; q < bxn * bzn; ++q) {
        if (st[q] != 0) continue;
        const int bx = q / bzn, bz = q % bzn, cx = s.vnl + bx, cz = s.vnt + bz;
        const int dx[4] = { -1, 1, 0, 0 }, dz[4] = { 0, 0, -1, 1 };
        for (int d = 0; d < 4; ++d) {
            const int nx2 = cx + dx[d], nz2 = cz + dz[d];
            if (nx2 < 1 || nx2 > g.nnx || nz2 < 1 || nz2 > g.nnz) continue;
            const int ox = bx + dx[d], oz = bz + dz[d];
            const bool inbox = ox >= 0 && ox < bxn && oz >= 0 && oz < bzn;
            if (!inbox || st[ox * bzn + oz] == -1) pr[q] = 1;
        }
    }
    for (int q = 0; q < bxn * bzn; ++q) if (pr[q]) st[q] = 1;
    std::vector<XRec> Fc(nrc, XRec{ 0.0f, -1 });
    m.F = Fc.data(); m.slow = slow_c.data(); m.risti = risti_c.data();
    x_set_grid(m, g.nbz, g.nnx, g.nnz); m.dnx = g.dnx; m.dnz = g.dnz;
    m.ntr = 0; m.pops = 0u;
    for (int q = 0; q < bxn * bzn; ++q)
        if (st[q] == 0) Fc[rec_index(g.nbz, s.vnt + q % bzn - 1, s.vnl + q / bzn - 1)] = XRec{ sT[q], 0 };
    for (int q = 0; q < bxn * bzn; ++q) {
        if (st[q] <= 0) continue;
        const int id = rec_index(g.nbz, s.vnt + q % bzn - 1, s.vnl + q / bzn - 1);
        Fc[id].T = sT[q];
        x_add(m, id, sT[q]);
    }
    int maxtree = m.ntr;
    while (m.ntr > 0 && m.error == 0) {
        const XEntry root = xh_get(m, 1);
        int iz0, ix0;
        x_coords(m, root.id, &iz0, &ix0);
        x_accept_root(m, root, iz0, ix0);
        if (m.ntr > maxtree) maxtree = m.ntr;
    }
    if (m.error) return m.error;
    if (stat) { stat[1] = (long)m.pops; stat[2] = maxtree; }
    for (int ix = 0; ix < g.nnx; ++ix)
        for (int iz = 0; iz < g.nnz; ++iz) {
            const XRec r = Fc[rec_index(g.nbz, iz, ix)];
            T[(size_t)ix * g.nnz + iz] = r.st == 0 ? r.T : kInf;
        }
    return 0;
}


// the quadrant form of the stencil (exact_march.h: x_trial_of_quads, what the sixteen lanes of k_exact evaluate) against fouds2 on n
// random neighbourhoods; returns the number of results whose bits differ
extern "C" long hc_quads_compare(unsigned long long seed, long n)
{
    unsigned long long st = seed * 6364136223846793005ull + 1442695040888963407ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
    auto uni = [&]() { return (float)(rnd() & 0xffffff) / 16777216.0f; };
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        NodeGeom g;
        g.ri = 6371.0f - 40.0f * uni();
        g.risti = g.ri * sinf(0.3f + 1.2f * uni());
        g.dnx = 5e-5f + 6e-4f * uni();
        g.dnz = (rnd() & 3) ? g.dnx * (0.7f + 0.6f * uni()) : g.dnx;
        const float slown = 1.0f / (1.5f + 3.5f * uni());
        const float hx = g.ri * g.dnx * slown, hz = g.risti * g.dnz * slown;
        const float t0 = (rnd() & 7) ? 300.0f * uni() : 2.0f * uni();
        Stencil s;
        for (int d = 0; d < 2; ++d) {
            const unsigned r = rnd();
            s.ej[d] = (r & 15) != 0; s.ek[d] = ((r >> 4) & 15) != 0;
            s.aj[d] = s.ej[d] && ((r >> 8) & 3) != 0; s.ak[d] = s.ek[d] && ((r >> 10) & 3) != 0;
            s.oj[d] = s.aj[d] && ((r >> 12) & 3) != 0; s.ok[d] = s.ak[d] && ((r >> 14) & 3) != 0;
            s.tj[d] = s.aj[d] ? t0 + hx * (2.4f * uni() - 1.2f) : kInf;
            s.tk[d] = s.ak[d] ? t0 + hz * (2.4f * uni() - 1.2f) : kInf;
            if ((r >> 16) & 1) { if (s.aj[d] && s.ak[d]) s.tk[d] = s.tj[d]; }          // ties
            s.tj2[d] = s.oj[d] ? s.tj[d] + hx * (1.6f * uni() - 1.1f) : kInf;
            s.tk2[d] = s.ok[d] ? s.tk[d] + hz * (1.6f * uni() - 1.1f) : kInf;
            if (((r >> 17) & 7) == 0 && s.oj[d]) s.tj2[d] = s.tj[d];
        }
        XQuadState q4[4];
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                XQuadState& q = q4[2 * j + k];
                q.ej = s.ej[j]; q.aj = s.aj[j]; q.oj = s.oj[j]; q.tj = s.tj[j]; q.tj2 = s.tj2[j];
                q.ek = s.ek[k]; q.ak = s.ak[k]; q.ok = s.ok[k]; q.tk = s.tk[k]; q.tk2 = s.tk2[k];
            }
        const float a = fouds2(s, slown, g), b = x_trial_of_quads(q4, slown, g), b2 = x_trial_of_quads_literal(q4, slown, g);
        if (std::memcmp(&a, &b, 4) != 0 || std::memcmp(&a, &b2, 4) != 0) { if (bad < 5) std::fprintf(stderr, "quadrant form differs at case %ld: %.9g vs %.9g\n", i, (double)a, (double)b); ++bad; }
    }
    return bad;
}

// Tie detector study (tools only): the device schedule emulated as in hc_device_schedule with the detector on -- the largest tie
// influence met by ANY evaluation (what the kernel's inline detector reports) against the largest one a scan of the CONVERGED field
// finds (final-state ties only).  out[0] = inline maximum, out[1] = final-state maximum, out[2] / out[3] = evaluations / nodes with a tie
// above `thr`.
extern "C" int hc_tie_study(int nnx, int nnz, float* Tio, float* tauio, const float* slow_rm, const float* risti, float ri, float dnx, float dnz,
                            float window, float thr, double* out)
{
    const int nbz = tiles_of(nnz);
    std::vector<Rec> F = tile_fill(nnx, nnz);
    std::vector<float> slow(F.size(), 1.0f);
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const int id = rec_index(nbz, iz, ix);
            F[id] = Rec{ Tio[(size_t)ix * nnz + iz], tauio[(size_t)ix * nnz + iz] };
            slow[id] = slow_rm[(size_t)ix * nnz + iz];
        }
    Field f = { nnx, nnz, nbz, F.data(), slow.data(), risti, ri, dnx, dnz };
    const size_t n = F.size();
    std::vector<int> cur, next, ready;
    std::vector<unsigned char> queued(n, 0);
    auto act = [&](int iz0, int ix0) {
        if (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) return;
        const int id = rec_index(nbz, iz0, ix0);
        if (t_pinned(F[id].T) || queued[id]) return;
        queued[id] = 1; next.push_back(id);
    };
    for (int ix = 0; ix < nnx; ++ix) for (int iz = 0; iz < nnz; ++iz)
        if (t_pinned(F[rec_index(nbz, iz, ix)].T)) { act(iz, ix - 1); act(iz, ix + 1); act(iz - 1, ix); act(iz + 1, ix); }
    cur.swap(next);
    float theta = kInf;
    double inline_max = 0.0; long inline_n = 0;
    auto tv = [&](int iz0, int ix0) { return (ix0 < 0 || ix0 >= nnx || iz0 < 0 || iz0 >= nnz) ? kInf : tau_value(F[rec_index(nbz, iz0, ix0)].tau); };
    long rounds = 0;
    while (!cur.empty() && rounds < 200000) {
        float tmin = kInf; ready.clear();
        for (int id : cur) {
            int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
            const float lb = fminf(fminf(tv(iz0, ix0 - 1), tv(iz0, ix0 + 1)), fminf(tv(iz0 - 1, ix0), tv(iz0 + 1, ix0)));
            if (!(theta < kInf) || lb < theta) { ready.push_back(id); queued[id] = 0; }
            else { next.push_back(id); tmin = fminf(tmin, lb); }
        }
        for (int pass = 0; pass < 2; ++pass)
            for (int id : ready) {
                int iz0, ix0; rec_coords(nbz, id, &iz0, &ix0);
                if (((ix0 + iz0) & 1) != pass) continue;
                const Hood h = load_hood(f, iz0 + 1, ix0 + 1); const NodeGeom g = { ri, risti[ix0], dnx, dnz };
                float k, tie;
                const float c = solve_node_t<true>(h, slow[id], g, &k, &tie);
                if (tie > thr) { ++inline_n; if (tie > inline_max) inline_max = tie; }
                if (std::memcmp(&c, &F[id].T, 4) || std::memcmp(&k, &F[id].tau, 4)) {
                    F[id].T = c; F[id].tau = k;
                    const int dz[4] = { 0, 0, -1, 1 }, dx[4] = { -1, 1, 0, 0 };
                    for (int q = 0; q < 4; ++q) { act(iz0 + dz[q], ix0 + dx[q]); act(iz0 + 2 * dz[q], ix0 + 2 * dx[q]); }
                    tmin = fminf(tmin, k);
                }
            }
        cur.swap(next); next.clear(); theta = tmin + window; ++rounds;
    }
    double final_max = 0.0; long final_n = 0;
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const int id = rec_index(nbz, iz, ix);
            if (t_pinned(F[id].T)) continue;
            const Hood h = load_hood(f, iz + 1, ix + 1); const NodeGeom g = { ri, risti[ix], dnx, dnz };
            float k, tie;
            solve_node_t<true>(h, slow[id], g, &k, &tie);
            if (tie > thr) { ++final_n; if (tie > final_max) final_max = tie; }
        }
    out[0] = inline_max; out[1] = final_max; out[2] = (double)inline_n; out[3] = (double)final_n;
    return cur.empty() ? 0 : -1;
}
