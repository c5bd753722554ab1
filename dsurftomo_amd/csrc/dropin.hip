// Drop-in level of the C ABI: the reference's CalSurfG / synthetic argument lists
// (reference src/CalSurfG.f90:939-943 and :2412-2415) on top of one process-wide engine.
//
// This file only translates the reference's calling convention into engine calls: which maps and
// depth-kernel slots each wave type gets, the (period slot, source) loop nest flattened into units,
// and the two-pass handling of group-velocity data.  All numerical work happens on the device.
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"

namespace {

std::string g_dropin_error = "";
long long g_capacity = 0;             // entries rw / iw / col can take (dsa_dropin_set_capacity); 0: not told
dsa_engine* g_engine = nullptr;

int g_last_first[4] = { -1, -1, -1, -1 }, g_last_count[4] = { 0, 0, 0, 0 }, g_last_ncol = 0;   // maps of the last call: Rc, Rg, Lc, Lg
std::vector<dsa_engine*> g_pool;      // one engine per GPU of this process (DSA_DEVICES), g_engine = g_pool[0]

// DSA_DEVICES = "4" (the first four GPUs) or "0,2,3" (a list); default: the single GPU DSA_DEVICE (0).
// An index may repeat (two engines on one GPU), which is how the sharded path is tested on one GPU.
int engine()
{
    if (g_engine) return 0;
    std::vector<int> devs;
    if (const char* s = getenv("DSA_DEVICES")) {
        const std::string str(s);
        if (str.find(',') == std::string::npos) { for (int k = 0; k < atoi(s); ++k) devs.push_back(k); }
        else { size_t p = 0; while (p < str.size()) { devs.push_back(atoi(str.c_str() + p)); p = str.find(',', p); if (p == std::string::npos) break; ++p; } }
    }
    if (devs.empty()) devs.push_back(getenv("DSA_DEVICE") ? atoi(getenv("DSA_DEVICE")) : 0);
    for (int dev : devs) {
        dsa_engine* e = nullptr;
        const int rc = dsa_create(&e, dev);
        if (rc != 0) {
            g_dropin_error = dsa_error_string(nullptr);
            for (dsa_engine* q : g_pool) dsa_destroy(q);
            g_pool.clear();
            return rc;
        }
        if (const char* s = getenv("DSA_MAX_CHUNK")) dsa_set_option(e, "max_chunk", atof(s));
        if (const char* s = getenv("DSA_WINDOW_CELLS")) dsa_set_option(e, "window_cells", atof(s));
        if (const char* s = getenv("DSA_EXACT_TIES")) dsa_set_option(e, "exact_ties", atof(s));       // exact mode for an unchanged Fortran host
        if (const char* s = getenv("DSA_TIE_THRESHOLD")) dsa_set_option(e, "tie_threshold", atof(s));
        if (const char* s = getenv("DSA_DISP_FAILURE_LOG")) dsa_set_option(e, "disp_failure_log", atof(s));   // per-call unit-66 blocks (dsa_dropin_dispersion_failure)
        if (const char* s = getenv("DSA_BUNDLE")) dsa_set_option(e, "bundle", atof(s));               // 0 off, 1 automatic (default), 4 / 8 / 16 members
        g_pool.push_back(e);
    }
    g_engine = g_pool[0];
    return 0;
}

int fail(int rc)
{
    g_dropin_error = dsa_error_string(g_engine);
    return rc;
}

// where the four wave types live in the engine's map store and depth-kernel slots
struct Layout {
    int kRc, kRg, kLc, kLg, kmax;
    int oRc, oRg, oLc, oLg, nmaps;     // first map of pvRc / pvRg / pvLc / pvLg
    int sRc, sRg, sLc, sLg;            // first depth-kernel slot (reference kmax1/kmax2/kmax3 offsets, :1094-1098)
};

Layout make_layout(int kRc, int kRg, int kLc, int kLg, int kmax, bool clobber)
{
    Layout L{};
    L.kRc = kRc; L.kRg = kRg; L.kLc = kLc; L.kLg = kLg; L.kmax = kmax;
    // pvRc / pvLc are sized kmax in the reference because the group periods' phase velocities are
    // written over their head; here they only need max(phase periods, group periods) columns
    const int nRc = clobber ? std::max(kRc, kRg) : kRc, nLc = clobber ? std::max(kLc, kLg) : kLc;
    L.oRc = 0; L.oRg = L.oRc + nRc; L.oLc = L.oRg + kRg; L.oLg = L.oLc + nLc; L.nmaps = L.oLg + kLg;
    L.sRc = 0; L.sRg = kRc; L.sLc = kRc + kRg; L.sLg = kRc + kRg + kLc;
    return L;
}

long long g_tie_units = 0, g_tie_left = 0, g_tie_marched = 0;      // tie census of the last drop-in call (dsa_dropin_tie_diagnostics)
long long g_tie_tied = 0, g_tie_prone = 0, g_tie_strict = 0;       // (round 6: dsa_dropin_tie_census)
float g_tie_influence = 0.0f;
void tie_reset() { g_tie_units = g_tie_left = g_tie_marched = g_tie_tied = g_tie_prone = g_tie_strict = 0; g_tie_influence = 0.0f; }
void tie_collect(dsa_engine* e)
{
    double st[DSA_STAT_COUNT + 8];
    if (dsa_get_stats(e, st) != 0) return;
    g_tie_units += (long long)st[DSA_STAT_TIE_UNITS]; g_tie_left += (long long)st[DSA_STAT_TIE_UNITS_LEFT]; g_tie_marched += (long long)st[DSA_STAT_EXACT_UNITS];
    g_tie_influence = std::max(g_tie_influence, (float)st[DSA_STAT_TIE_INFLUENCE_MAX]);
    g_tie_tied += (long long)st[DSA_STAT_TIE_UNITS_TIED]; g_tie_prone += (long long)st[DSA_STAT_TIE_PRONE_MAPS]; g_tie_strict += (long long)st[DSA_STAT_TIE_UNITS_STRICT];
}

int g_rbint_notes = 0;                // diagnostics of the last dsa_calsurfg call (dsa_dropin_diagnostics)
long long g_disp_count = 0;
int g_disp_first[5] = { 0, 0, 0, 0, 0 };
double g_disp_period = 0.0;

struct Units {
    std::vector<int> iter;             // (period slot, source) iteration of the reference's loop nest the unit belongs to, 0-based
    int niter = 0;
    std::vector<int> map, nrec, mode, slot, data;
    std::vector<float> sx, sz, rx, rz;
    int ndata = 0;
};

// flatten CalSurfG.f90:1144-1183: returns 0 or DSA_ERR_ARGUMENT
int make_units(const Layout& L, bool rows, int nsrcsurf, int nrcf, const int* wavetype, const int* igrt, const int* periods,
               const int* nrc1, const int* nsrcsurf1, const float* scxf, const float* sczf, const float* rcxf, const float* rczf,
               Units& U)
{
    int count1 = 0;
    for (int knumi = 1; knumi <= L.kmax; ++knumi)
        for (int srcnum = 1; srcnum <= nsrcsurf1[knumi - 1]; ++srcnum) {
            const size_t sk = (size_t)(knumi - 1) * nsrcsurf + (srcnum - 1);
            const int wt = wavetype[sk], gr = igrt[sk], per = periods[sk], nr = nrc1[sk];
            int mt, mp;      // map of the travel times, map of the rays
            if (wt == 2 && gr == 0) { mt = L.oRc + per - 1; mp = mt; }
            else if (wt == 2 && gr == 1) { mt = L.oRg + per - 1; mp = L.oRc + per - 1; }
            else if (wt == 1 && gr == 0) { mt = L.oLc + per - 1; mp = mt; }
            else if (wt == 1 && gr == 1) { mt = L.oLg + per - 1; mp = L.oLc + per - 1; }
            else { g_dropin_error = "wavetype must be 1 (Love) or 2 (Rayleigh) and igrt 0 or 1"; return DSA_ERR_ARGUMENT; }
            if (per < 1 || mt >= L.nmaps || nr < 0 || nr > nrcf) { g_dropin_error = "period index or receiver count out of range"; return DSA_ERR_ARGUMENT; }
            const int passes = (rows && gr == 1) ? 2 : 1;
            for (int ig = 1; ig <= passes; ++ig) {
                U.iter.push_back(U.niter);
                U.map.push_back(ig == 1 ? mt : mp);
                U.nrec.push_back(nr);
                U.mode.push_back(!rows ? 1 : (gr == 0 ? 3 : (ig == 1 ? 1 : 2)));
                U.slot.push_back(knumi - 1);
                U.data.push_back(count1);
                U.sx.push_back(scxf[sk]);
                U.sz.push_back(sczf[sk]);
                for (int i = 0; i < nr; ++i) {
                    U.rx.push_back(rcxf[sk * (size_t)nrcf + i]);
                    U.rz.push_back(rczf[sk * (size_t)nrcf + i]);
                }
            }
            count1 += nr;
            U.niter += 1;
        }
    U.ndata = count1;
    return 0;
}

void remember(const Layout& L, int ncol, bool clobbered)
{
    // after a CalSurfG call the head of the phase-velocity blocks holds the group periods' phase velocities
    (void)clobbered;
    g_last_first[0] = L.oRc; g_last_first[1] = L.oRg; g_last_first[2] = L.oLc; g_last_first[3] = L.oLg;
    g_last_count[0] = L.kRc; g_last_count[1] = L.kRg; g_last_count[2] = L.kLc; g_last_count[3] = L.kLg;
    g_last_ncol = ncol;
}

}  // namespace

extern "C" {

const char* dsa_dropin_error(void) { return g_dropin_error.c_str(); }

// the process-wide engine of the drop-in level (created on first use), for the engine-level calls that continue a drop-in
// call on the device: dsa_iteration_system_device, dsa_lsmr, dsa_get_stats
dsa_engine* dsa_dropin_engine(void) { return engine() == 0 ? g_engine : nullptr; }

// capacity (entries) of the rw / iw(2:) / col arrays handed to dsa_calsurfg from now on; 0 = unknown (DSA_MAXNAR or unlimited)
int dsa_dropin_set_capacity(long long maxnar)
{
    if (maxnar < 0) { g_dropin_error = "dsa_dropin_set_capacity: negative capacity"; return DSA_ERR_ARGUMENT; }
    g_capacity = maxnar;
    return 0;
}

int dsa_calsurfg(const int* nx, const int* ny, const int* nz, const int* nparpi, const float* vels,
                 int* iw, float* rw, int* col, float* dsurf,
                 const float* goxdf, const float* gozdf, const float* dvxdf, const float* dvzdf,
                 const int* kmaxRc, const int* kmaxRg, const int* kmaxLc, const int* kmaxLg,
                 const double* tRc, const double* tRg, const double* tLc, const double* tLg,
                 const int* wavetype, const int* igrt, const int* periods, const float* depz,
                 const float* minthk, const float* scxf, const float* sczf, const float* rcxf,
                 const float* rczf, const int* nrc1, const int* nsrcsurf1, const int* kmax,
                 const int* nsrcsurf, const int* nrcf, int* nar)
{
    (void)nparpi;
    // rw, iw and col may be null TOGETHER: the rows then stay on the device (extension; dsa_iteration_system_device / dsa_lsmr
    // on dsa_dropin_engine() continue from there)
    const bool device_rows = !iw && !rw && !col;
    if (!nx || !ny || !nz || !vels || (!device_rows && (!iw || !rw || !col)) || !dsurf || !kmax || !nar) { g_dropin_error = "dsa_calsurfg: null argument"; return DSA_ERR_ARGUMENT; }
    int rc = engine();
    if (rc != 0) return rc;
    if (device_rows && g_pool.size() != 1) { g_dropin_error = "dsa_calsurfg: rows can only stay on the device with one engine (DSA_DEVICES unset)"; return DSA_ERR_STATE; }
    const Layout L = make_layout(*kmaxRc, *kmaxRg, *kmaxLc, *kmaxLg, *kmax, true);
    if (L.kRc + L.kRg + L.kLc + L.kLg != L.kmax) { g_dropin_error = "dsa_calsurfg: kmax must equal kmaxRc+kmaxRg+kmaxLc+kmaxLg"; return DSA_ERR_ARGUMENT; }
    Units U;
    if ((rc = make_units(L, true, *nsrcsurf, *nrcf, wavetype, igrt, periods, nrc1, nsrcsurf1, scxf, sczf, rcxf, rczf, U)) != 0) return rc;
    // the reference's interface carries no capacity for rw / iw / col: the caller states it with dsa_dropin_set_capacity
    // (the Python host does, per call) or, for an unchanged Fortran host, with DSA_MAXNAR in the environment
    long long cap = LLONG_MAX;
    if (g_capacity > 0) cap = g_capacity;
    else if (const char* s = getenv("DSA_MAXNAR")) cap = atoll(s);
    *nar = 0;
    g_rbint_notes = 0; g_disp_count = 0;
    tie_reset();

    // Every engine runs the dispersion stage for the whole model (it is small next to the solves) and then its share of the unit
    // list.  Round 3: the share is a set of SOURCES with all their units -- the engine solves the periods of a source side by side
    // (bundles, bundle_kernel.hip), so a split by contiguous slices (whole periods) would leave every engine 1/ne of each source's periods.
    // The two units of a group-velocity datum have the same source and stay together by construction.
    const int ne = (int)g_pool.size(), nu = (int)U.map.size();
    std::vector<int> owner((size_t)nu, 0);
    if (ne > 1) {
        std::map<std::pair<uint32_t, uint32_t>, int> src_of;       // source (coordinates bit for bit) -> index in order of first appearance
        std::vector<int> src_units, unit_src((size_t)nu);
        for (int u = 0; u < nu; ++u) {
            uint32_t a, b2; std::memcpy(&a, &U.sx[(size_t)u], 4); std::memcpy(&b2, &U.sz[(size_t)u], 4);
            auto it = src_of.find({ a, b2 });
            if (it == src_of.end()) { it = src_of.insert({ { a, b2 }, (int)src_units.size() }).first; src_units.push_back(0); }
            unit_src[(size_t)u] = it->second; ++src_units[(size_t)it->second];
        }
        // blocks of consecutive sources with about nu / ne units each
        std::vector<int> src_engine(src_units.size(), 0);
        long long acc = 0; int k = 0;
        for (size_t q = 0; q < src_units.size(); ++q) {
            while (k + 1 < ne && acc >= (long long)nu * (k + 1) / ne) ++k;
            src_engine[q] = k; acc += src_units[q];
        }
        for (int u = 0; u < nu; ++u) owner[(size_t)u] = src_engine[(size_t)unit_src[(size_t)u]];
    }
    std::vector<size_t> ray0(nu + 1, 0);
    for (int u = 0; u < nu; ++u) ray0[u + 1] = ray0[u] + (size_t)U.nrec[u];
    struct Part { std::vector<float> rw; std::vector<int> iw, col; long long n = 0; int rc = 0; std::string err; int first_clamped = -1;
                  long long disp_count = 0; int disp_first[5] = { 0, 0, 0, 0, 0 }; double disp_period = 0.0;
                  std::vector<int> units; std::vector<float> times; };       // (several engines: the engine's units in call order; its receiver times by datum)
    std::vector<Part> part(ne);
    size_t ndata_all = 0;
    for (int u = 0; u < nu; ++u) ndata_all = std::max(ndata_all, (size_t)U.data[(size_t)u] + (size_t)U.nrec[(size_t)u]);
    if (ne > 1) for (int u = 0; u < nu; ++u) part[(size_t)owner[(size_t)u]].units.push_back(u);
    auto work = [&](int k) {
        dsa_engine* e = g_pool[k];
        dsa::Engine* en = reinterpret_cast<dsa::Engine*>(e);
        Part& P = part[k];
        auto bad = [&](int r) { P.rc = r; P.err = dsa_error_string(e); };
        int r;
        // dispersion: depth kernels per type; the phase velocities at the group periods overwrite the head
        // of the phase-velocity block (CalSurfG.f90:1100-1140)
        if ((r = dsa_dispersion_begin(e, *nx, *ny, *nz, vels, depz, *minthk, L.kmax, L.nmaps)) != 0) return bad(r);
        if ((r = dsa_dispersion_run(e, 2, 0, L.kRc, tRc, 1, L.sRc, L.oRc)) != 0) return bad(r);
        if (L.kRg > 0) {
            if ((r = dsa_dispersion_run(e, 2, 1, L.kRg, tRg, 1, L.sRg, L.oRg)) != 0) return bad(r);
            if ((r = dsa_dispersion_run(e, 2, 0, L.kRg, tRg, 0, 0, L.oRc)) != 0) return bad(r);
        }
        if ((r = dsa_dispersion_run(e, 1, 0, L.kLc, tLc, 1, L.sLc, L.oLc)) != 0) return bad(r);
        if (L.kLg > 0) {
            if ((r = dsa_dispersion_run(e, 1, 1, L.kLg, tLg, 1, L.sLg, L.oLg)) != 0) return bad(r);
            if ((r = dsa_dispersion_run(e, 1, 0, L.kLg, tLg, 0, 0, L.oLc)) != 0) return bad(r);
        }
        if ((r = dsa_maps_from_dispersion(e, *goxdf, *gozdf, *dvxdf, *dvzdf, 8)) != 0) return bad(r);
        if ((r = dsa_kernels_from_dispersion(e)) != 0) return bad(r);
        if (ne == 1) {
            if ((r = dsa_plan_units(e, nu, U.map.data(), U.sx.data(), U.sz.data(), U.nrec.data(), U.rx.data(), U.rz.data(), U.mode.data(), U.slot.data(), U.data.data())) != 0) return bad(r);
            en->rows_on_device = device_rows;
            r = dsa_solve_rows(e, dsurf, rw, device_rows ? nullptr : iw + 1, col, cap, &P.n);          // the reference fills iw(nar+1)
        } else {
            // this engine's units, in call order, with their receivers gathered; data indices stay those of the whole call
            const std::vector<int>& mine = P.units;
            const int nm = (int)mine.size();
            std::vector<int> m_map((size_t)nm), m_nrec((size_t)nm), m_mode((size_t)nm), m_slot((size_t)nm), m_data((size_t)nm);
            std::vector<float> m_sx((size_t)nm), m_sz((size_t)nm), m_rx, m_rz;
            for (int q = 0; q < nm; ++q) {
                const size_t u = (size_t)mine[(size_t)q];
                m_map[(size_t)q] = U.map[u]; m_sx[(size_t)q] = U.sx[u]; m_sz[(size_t)q] = U.sz[u]; m_nrec[(size_t)q] = U.nrec[u];
                m_mode[(size_t)q] = U.mode[u]; m_slot[(size_t)q] = U.slot[u]; m_data[(size_t)q] = U.data[u];
                m_rx.insert(m_rx.end(), U.rx.begin() + (long)ray0[u], U.rx.begin() + (long)ray0[u + 1]);
                m_rz.insert(m_rz.end(), U.rz.begin() + (long)ray0[u], U.rz.begin() + (long)ray0[u + 1]);
            }
            if ((r = dsa_plan_units(e, nm, m_map.data(), m_sx.data(), m_sz.data(), m_nrec.data(), m_rx.data(), m_rz.data(), m_mode.data(), m_slot.data(), m_data.data())) != 0) return bad(r);
            P.times.assign(std::max<size_t>(ndata_all, 1), 0.0f);
            en->grow_rw = &P.rw; en->grow_iw = &P.iw; en->grow_col = &P.col;
            r = nm ? en->solve(P.times.data(), nullptr, nullptr, nullptr, cap, &P.n) : 0;
            en->grow_rw = nullptr; en->grow_iw = nullptr; en->grow_col = nullptr;
        }
        if (r != 0) return bad(r);
        long long nclamped = 0;
        int fu = -1;
        // (an engine that was given no sources did not solve: its diagnostics are those of an earlier call)
        const bool solved = ne == 1 || !P.units.empty();
        if (solved && dsa_ray_diagnostics(e, &nclamped, &fu) == 0 && fu >= 0 && (ne == 1 ? fu < nu : (size_t)fu < P.units.size()))
            P.first_clamped = ne == 1 ? fu : P.units[(size_t)fu];       // unit of the whole call
        dsa_dispersion_diagnostics(e, &P.disp_count, P.disp_first, &P.disp_period);
    };
    if (ne == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int k = 0; k < ne; ++k) th.emplace_back(work, k);
        for (auto& t : th) t.join();
    }
    long long n = 0;
    int first_clamped = -1;
    for (int k = 0; k < ne; ++k) {
        if (part[k].rc != 0) { g_dropin_error = part[k].err; return part[k].rc; }
        if (ne == 1 || !part[k].units.empty()) tie_collect(g_pool[(size_t)k]);
        n += part[k].n;
        if (part[k].first_clamped >= 0 && (first_clamped < 0 || part[k].first_clamped < first_clamped)) first_clamped = part[k].first_clamped;
    }
    if (ne > 1) {
        if (n > cap) { g_dropin_error = "dsa_calsurfg: more matrix entries than the stated capacity (increase sparsity fraction)"; return DSA_ERR_CAPACITY; }
        // receiver times: every engine's own data; matrix entries: the reference appends them datum by datum in call order, and a datum's
        // rows come from one unit, i.e. from one engine, whose list holds them in order -- so the whole list is the engines' runs placed
        // by datum (iw = the datum's 1-based row)
        for (int k = 0; k < ne; ++k)
            for (int u : part[k].units)
                for (int q = 0; q < U.nrec[(size_t)u]; ++q) { const size_t d = (size_t)U.data[(size_t)u] + (size_t)q; if (U.mode[(size_t)u] & 1) dsurf[d] = part[k].times[d]; }
        std::vector<long long> first_of(ndata_all + 2, 0);
        for (int k = 0; k < ne; ++k) for (long long q = 0; q < part[k].n; ++q) ++first_of[(size_t)part[k].iw[(size_t)q] + 1];       // (iw 1-based: count of datum d-1 at [d+1])
        for (size_t d = 1; d < first_of.size(); ++d) first_of[d] += first_of[d - 1];
        std::vector<long long> fill(first_of.begin(), first_of.end());
        for (int k = 0; k < ne; ++k)
            for (long long q = 0; q < part[k].n; ++q) {
                const long long pos = fill[(size_t)part[k].iw[(size_t)q]]++;
                rw[pos] = part[k].rw[(size_t)q]; iw[1 + pos] = part[k].iw[(size_t)q]; col[pos] = part[k].col[(size_t)q];
            }
    }
    // every engine ran the same dispersion stage: engine 0 speaks for all
    g_disp_count = part[0].disp_count; g_disp_period = part[0].disp_period;
    for (int q = 0; q < 5; ++q) g_disp_first[q] = part[0].disp_first[q];
    // the reference tests rbint after every (period, source) iteration and never clears it inside a call (CalSurfG.f90:1088, :1447)
    g_rbint_notes = first_clamped >= 0 ? U.niter - U.iter[(size_t)first_clamped] : 0;
    if (n > INT_MAX) { g_dropin_error = "dsa_calsurfg: more than 2^31-1 matrix entries"; return DSA_ERR_ARGUMENT; }
    *nar = (int)n;
    remember(L, *nx * *ny, true);
    return 0;
}

int dsa_dropin_dispersion_failure(int index, int* info, double* vals, float* table, double* c)
{
    dsa_engine* e = dsa_dropin_engine();          // (every engine of a multi-device call runs the whole dispersion stage: the first one's log)
    if (!e) return DSA_ERR_STATE;
    return dsa_dispersion_failure(e, index, info, vals, table, c);
}

// Tie census of the last dsa_calsurfg / dsa_synthetic call (all its engines): units that hold an exact time tie with an influence above the
// threshold, how many of them were left to the fixed point (exact_ties = 0: their travel times may differ from the reference's Fast Marching by
// more than 1e-4 s) and how many were solved again by the reference's march (exact_ties = 1), and the largest influence met.
// (round 6) ... and what the per-unit rule cannot see: units that stayed with the fixed point although they hold a tie with a (small) influence -- their
// times are the reference's to 1e-4 s by measurement, not by construction --, the maps found tie-prone and the units marched because of their map.
int dsa_dropin_tie_census(long long* tied_units_left, long long* tie_prone_maps, long long* flagged_by_map)
{
    if (tied_units_left) *tied_units_left = g_tie_tied;
    if (tie_prone_maps) *tie_prone_maps = g_tie_prone;
    if (flagged_by_map) *flagged_by_map = g_tie_strict;
    return 0;
}

int dsa_dropin_tie_diagnostics(long long* flagged_units, long long* left_to_fixed_point, long long* marched_units, float* largest_influence)
{
    if (flagged_units) *flagged_units = g_tie_units;
    if (left_to_fixed_point) *left_to_fixed_point = g_tie_left;
    if (marched_units) *marched_units = g_tie_marched;
    if (largest_influence) *largest_influence = g_tie_influence;
    return 0;
}

int dsa_dropin_diagnostics(int* rbint_notes, long long* disp_count, int* disp_first, double* disp_period)
{
    if (rbint_notes) *rbint_notes = g_rbint_notes;
    if (disp_count) *disp_count = g_disp_count;
    if (disp_first) for (int q = 0; q < 5; ++q) disp_first[q] = g_disp_first[q];
    if (disp_period) *disp_period = g_disp_period;
    return 0;
}

// aprod with the reference's argument list (aprod.f90:7; LSMR calls it with the same matrix hundreds of times per
// inversion step, lsmrModule.f90:390-497).  The matrix goes to the device the first time it is seen; it is
// recognised again by its address, its size and a sample of its entries (`force`: load regardless).
static bool g_matrix_stale = true;       // dsa_aprod_invalidate, or nothing loaded yet
static int g_last_mode = 2;

static int load_matrix_cached(int m, int n, const int* iw, const float* rw, bool force)
{
    if (g_matrix_stale) force = true;
    static const void *s_iw = nullptr, *s_rw = nullptr;
    static long long s_nar = -1;
    static int s_m = 0, s_n = 0;
    static double s_sum = 0.0;
    const long long nar = iw[0];
    double sum = 0.0;
    const long long step = std::max<long long>(1, nar / 997);
    for (long long k = 0; k < nar; k += step) sum += (double)rw[k] * (double)(1 + (k & 7)) + (double)iw[1 + k] + 3.0 * (double)iw[1 + nar + k];
    if (force || iw != s_iw || rw != s_rw || nar != s_nar || m != s_m || n != s_n || sum != s_sum) {
        const int rc = dsa_spmv_load(g_engine, m, n, nar, rw, iw + 1, iw + 1 + nar);
        if (rc != 0) return rc;
        s_iw = iw; s_rw = rw; s_nar = nar; s_m = m; s_n = n; s_sum = sum;
        g_matrix_stale = false;
    }
    return 0;
}

int dsa_aprod(const int* mode, const int* m, const int* n, float* x, float* y, const int* leniw, const int* lenrw,
              const int* iw, const float* rw)
{
    (void)leniw; (void)lenrw;
    if (!mode || !m || !n || !x || !y || !iw || !rw) { g_dropin_error = "dsa_aprod: null argument"; return DSA_ERR_ARGUMENT; }
    int rc = engine();
    if (rc != 0) return rc;
    // Inside one LSMR solve the products alternate (lsmrModule.f90:390, then :484 / :497 per iteration: 2, 1, 2, 1, 2 ...)
    // and a solve ends on mode 2, so two mode-2 products in a row mean a new solve -- the reference rebuilds rw / iw in
    // place before each one (main.f90:361-466), same addresses, same size: reload.  Anything else a caller edits in
    // place must be announced with dsa_aprod_invalidate().
    const bool new_solve = *mode == 2 && g_last_mode == 2;
    g_last_mode = *mode;
    if ((rc = load_matrix_cached(*m, *n, iw, rw, new_solve)) != 0) return fail(rc);
    if ((rc = dsa_spmv(g_engine, *mode, x, y)) != 0) return fail(rc);
    return 0;
}

// the matrix behind the next dsa_aprod call has been edited in place: upload it again
int dsa_aprod_invalidate(void)
{
    g_matrix_stale = true;
    return 0;
}

// LSMR with the reference's argument list (lsmrModule.f90:36-39, called at main.f90:487); nout is ignored (the
// reference's main program never opens that unit, main.f90:47,107).  dsurftomo_amd/fortran/lsmr_shim.f90 exports
// the module procedure.
int dsa_lsmr_dropin(const int* m, const int* n, const int* leniw, const int* lenrw, const int* iw, const float* rw,
                    const float* b, const float* damp, const float* atol, const float* btol, const float* conlim,
                    const int* itnlim, const int* localSize, const int* nout, float* x, int* istop, int* itn,
                    float* normA, float* condA, float* normr, float* normAr, float* normx)
{
    (void)leniw; (void)lenrw; (void)nout;
    if (!m || !n || !iw || !rw || !b || !damp || !atol || !btol || !conlim || !itnlim || !localSize || !x) { g_dropin_error = "dsa_lsmr_dropin: null argument"; return DSA_ERR_ARGUMENT; }
    int rc = engine();
    if (rc != 0) return rc;
    if ((rc = load_matrix_cached(*m, *n, iw, rw, true)) != 0) return fail(rc);      // one solve per matrix: always current
    if ((rc = dsa_lsmr(g_engine, b, *damp, *atol, *btol, *conlim, *itnlim, *localSize, x, istop, itn, normA, condA, normr, normAr, normx)) != 0) return fail(rc);
    return 0;
}

// phase / group velocity maps of the last drop-in call, pv(nx*ny, count) fp64 in the reference's layout;
// which = 0 Rayleigh phase, 1 Rayleigh group, 2 Love phase, 3 Love group.  (The reference's `synthetic`
// writes them to velmap2d*.dat, CalSurfG.f90:2559-2617; the Fortran shim does that with this call.)
int dsa_dropin_velocity_maps(const int* which, double* pv)
{
    if (!which || !pv || *which < 0 || *which > 3 || !g_engine || g_last_first[*which] < 0) { g_dropin_error = "dsa_dropin_velocity_maps: no maps (call dsa_synthetic / dsa_calsurfg first)"; return DSA_ERR_STATE; }
    const int rc = dsa_dispersion_fetch(g_engine, g_last_first[*which], g_last_count[*which], pv, 0, 0, nullptr, nullptr, nullptr);
    return rc != 0 ? fail(rc) : 0;
}


int dsa_synthetic(const int* nx, const int* ny, const int* nz, const int* nparpi, const float* vels,
                  float* obst,
                  const float* goxdf, const float* gozdf, const float* dvxdf, const float* dvzdf,
                  const int* kmaxRc, const int* kmaxRg, const int* kmaxLc, const int* kmaxLg,
                  const double* tRc, const double* tRg, const double* tLc, const double* tLg,
                  const int* wavetype, const int* igrt, const int* periods, const float* depz,
                  const float* minthk, const float* scxf, const float* sczf, const float* rcxf,
                  const float* rczf, const int* nrc1, const int* nsrcsurf1, const int* kmax,
                  const int* nsrcsurf, const int* nrcf, const float* noiselevel)
{
    (void)nparpi;
    if (!nx || !ny || !nz || !vels || !obst || !kmax) { g_dropin_error = "dsa_synthetic: null argument"; return DSA_ERR_ARGUMENT; }
    int rc = engine();
    if (rc != 0) return rc;
    dsa_engine* e = g_engine;
    const Layout L = make_layout(*kmaxRc, *kmaxRg, *kmaxLc, *kmaxLg, *kmax, false);
    // caldespersion per type, no depth kernels, dicing 5 (CalSurfG.f90:2487-2617)
    if ((rc = dsa_dispersion_begin(e, *nx, *ny, *nz, vels, depz, *minthk, std::max(L.kmax, 1), std::max(L.nmaps, 1))) != 0) return fail(rc);
    if ((rc = dsa_dispersion_run(e, 2, 0, L.kRc, tRc, 0, 0, L.oRc)) != 0) return fail(rc);
    if ((rc = dsa_dispersion_run(e, 2, 1, L.kRg, tRg, 0, 0, L.oRg)) != 0) return fail(rc);
    if ((rc = dsa_dispersion_run(e, 1, 0, L.kLc, tLc, 0, 0, L.oLc)) != 0) return fail(rc);
    if ((rc = dsa_dispersion_run(e, 1, 1, L.kLg, tLg, 0, 0, L.oLg)) != 0) return fail(rc);
    if ((rc = dsa_maps_from_dispersion(e, *goxdf, *gozdf, *dvxdf, *dvzdf, 5)) != 0) return fail(rc);
    remember(L, *nx * *ny, false);
    Units U;
    if ((rc = make_units(L, false, *nsrcsurf, *nrcf, wavetype, igrt, periods, nrc1, nsrcsurf1, scxf, sczf, rcxf, rczf, U)) != 0) return rc;
    if ((rc = dsa_plan_units(e, (int)U.map.size(), U.map.data(), U.sx.data(), U.sz.data(), U.nrec.data(), U.rx.data(), U.rz.data(),
                             U.mode.data(), nullptr, U.data.data())) != 0) return fail(rc);
    tie_reset();
    if ((rc = dsa_solve(e, obst)) != 0) return fail(rc);
    tie_collect(e);
    // obst = t + t * gaussian() * noiselevel (:2840).  The reference draws from the compiler's unseeded
    // random_number; the Fortran shim calls this entry with noiselevel 0 and adds the reference's own
    // gaussian() on its side.  Called directly with a non-zero level, a private generator is used.
    if (noiselevel && *noiselevel != 0.0f) {
        unsigned long long s = 0x9E3779B97F4A7C15ull;
        auto uni = [&s]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) * (1.0 / 16777216.0)); };
        for (int k = 0; k < U.ndata; ++k) {
            float x1, x2, w = 2.0f;
            while (w >= 1.0f) { x1 = 2.0f * uni() - 1.0f; x2 = 2.0f * uni() - 1.0f; w = x1 * x1 + x2 * x2; }
            w = sqrtf((-2.0f * logf(w)) / w);
            obst[k] = obst[k] + obst[k] * (x1 * w) * *noiselevel;
        }
    }
    return 0;
}

}  // extern "C"
