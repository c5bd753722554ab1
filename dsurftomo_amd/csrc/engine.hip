// Host side of the engine-level C ABI (include/dsurftomo_amd.h): owns device memory, turns the
// caller's (map, source, receivers) units into descriptors, and sequences the kernels per chunk
// of sources on one HIP stream.  No numerical work happens here except exact fp32 geometry and
// the libm sine tables (host_geometry.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"

namespace dsa {

static std::string g_create_error;

void Engine::fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    error = buf;
    status = code;
}

#define HIP_TRY(e, call)                                                                       \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) {                                                                \
            (e)->fail(DSA_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, __LINE__); \
            return DSA_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

template <class T>
int Engine::ensure(DevBuf<T>& b, size_t n)
{
    if (b.cap >= n) return 0;
    if (b.p) { HIP_TRY(this, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    if (n == 0) return 0;
    HIP_TRY(this, hipMalloc(reinterpret_cast<void**>(&b.p), n * sizeof(T)));
    b.cap = n;
    return 0;
}

// like ensure, but the first `used` elements survive a reallocation (growth by at least a half)
template <class T>
int Engine::ensure_keep(DevBuf<T>& b, size_t n, size_t used)
{
    if (b.cap >= n) return 0;
    const size_t want = std::max(n, b.cap + b.cap / 2);
    T* q = nullptr;
    HIP_TRY(this, hipMalloc(reinterpret_cast<void**>(&q), want * sizeof(T)));
    if (b.p && used) HIP_TRY(this, hipMemcpyAsync(q, b.p, used * sizeof(T), hipMemcpyDeviceToDevice, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    if (b.p) HIP_TRY(this, hipFree(b.p));
    b.p = q; b.cap = want;
    return 0;
}
template int Engine::ensure_keep<float>(DevBuf<float>&, size_t, size_t);
template int Engine::ensure_keep<int>(DevBuf<int>&, size_t, size_t);

void release_spmv(SpmvState* s);

template int Engine::ensure<long long>(DevBuf<long long>&, size_t);
template int Engine::ensure<float>(DevBuf<float>&, size_t);
template int Engine::ensure<int>(DevBuf<int>&, size_t);
template int Engine::ensure<unsigned char>(DevBuf<unsigned char>&, size_t);
template int Engine::ensure<unsigned long long>(DevBuf<unsigned long long>&, size_t);
template int Engine::ensure<unsigned short>(DevBuf<unsigned short>&, size_t);
template int Engine::ensure<FimBundle>(DevBuf<FimBundle>&, size_t);

Engine::~Engine()
{
    release_spmv(spmv);
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    rel(velv); rel(veln); rel(slow); rel(risti_c); rel(cbasis); rel(rbasis);
    rel(src); rel(rays); rel(out); rel(err);
    rel(slow_r); rel(F_r); rel(Tfin_r); rel(S_r); rel(risti_r); rel(vcorner); rel(seed_r); rel(nseed_r); rel(launch_rank); rel(slowI); rel(B_pool); rel(exc_b); rel(lists_b); rel(bpool_gen); rel(member_flag); rel(bundles_d);
    rel(rst); rel(cst); rel(cinit); rel(heap); rel(flags); rel(T_c); rel(exc_c); rel(W_c); rel(seed_c); rel(nseed_c);
    rel(prob_r); rel(prob_c); rel(paths); rel(path_n); rel(info); rel(clocks); rel(lists);
    rel(Srow); rel(sen_vs); rel(sen_vp); rel(sen_rho); rel(vels_d); rel(trace_ids); rel(vlist); rel(nvv); rel(counts); rel(offsets);
    rel(coo_col); rel(coo_iw); rel(slabs); rel(coo_rw); rel(rayinfo); rel(G_rw); rel(G_row); rel(G_col);
    rel(geom); rel(pvstore); rel(curves); rel(tper); rel(disp_ws);
    if (stream2) { (void)hipStreamDestroy(stream2); (void)hipEventDestroy(ev_b0); (void)hipEventDestroy(ev_b1); stream2 = nullptr; }
    rel(cand_b); rel(bundles_r_d); rel(ends_r); rel(Br_pool); rel(slowIr); rel(exc_br); rel(lists_br); rel(cand_br);
    rel(lists_c); rel(pool_gen); rel(ends_c); rel(disp_diag); rel(disp_fail_list); rel(X_pool); rel(X_heap); rel(X_tt); rel(X_tp); rel(X_ring); rel(X_free); rel(X_pins); rel(x_starts); rel(x_nstart); rel(x_units); rel(xinfo); rel(tieinfo);
    for (auto& ev : events) if (ev) (void)hipEventDestroy(ev);
    if (stream) (void)hipStreamDestroy(stream);
}

int Engine::init(int device_index)
{
    int ndev = 0;
    hipError_t r = hipGetDeviceCount(&ndev);
    if (r != hipSuccess || ndev <= 0) {
        fail(DSA_ERR_DEVICE, "no HIP device available (%s); this engine has no CPU path", r == hipSuccess ? "device count 0" : hipGetErrorString(r));
        return DSA_ERR_DEVICE;
    }
    if (device_index < 0 || device_index >= ndev) { fail(DSA_ERR_ARGUMENT, "device index %d out of range (0..%d)", device_index, ndev - 1); return DSA_ERR_ARGUMENT; }
    device = device_index;
    HIP_TRY(this, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(this, hipGetDeviceProperties(&prop, device));
    arch = prop.gcnArchName;
    HIP_TRY(this, hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& ev : events) HIP_TRY(this, hipEventCreate(&ev));
    return 0;
}

int Engine::set_maps(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int dicing, int nm, const double* pv)
{
    if (nx < 4 || ny < 4 || nm < 1 || !pv || dicing < 1 || dicing > 16) { fail(DSA_ERR_ARGUMENT, "set_maps: bad arguments (nx=%d ny=%d nmaps=%d dicing=%d)", nx, ny, nm, dicing); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    make_grid(g, nx, ny, goxd, gozd, dvxd, dvzd, dicing);
    const size_t nv = (size_t)nx * ny;
    std::vector<float> hv(nv * nm);
    for (size_t k = 0; k < nv * nm; ++k) hv[k] = (float)pv[k];       // velv = real(pv), CalSurfG.f90:1492
    if (ensure(velv, hv.size())) return status;
    HIP_TRY(this, hipMemcpyAsync(velv.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, stream));
    hmin_slow = 1e30f;
    for (float v : hv) if (v > 0.0f && 1.0f / v < hmin_slow) hmin_slow = 1.0f / v;
    mean_slowness_of_maps(hv.data(), nv, nm);
    return finish_maps(nm);
}

// mean slowness of each map's vertices: the scale of a unit's travel times (tie_verdicts: the default mode's validated envelope)
void Engine::mean_slowness_of_maps(const float* hv, size_t nv, int nm)
{
    map_mean_slow.assign((size_t)nm, 0.0f);
    for (int m = 0; m < nm; ++m) {
        double sum = 0.0; size_t k = 0;
        for (size_t i = 0; i < nv; ++i) { const float v = hv[(size_t)m * nv + i]; if (v > 0.0f) { sum += 1.0 / (double)v; ++k; } }
        map_mean_slow[(size_t)m] = k ? (float)(sum / (double)k) : 0.0f;
    }
}

// velv (nm maps of fp32 vertex values) is on the device and g / hmin_slow are set: tables + dicing
int Engine::finish_maps(int nm)
{
    if (g.nnx > 32767 || g.nnz > 32767) { fail(DSA_ERR_ARGUMENT, "grid %dx%d exceeds the 32767-node index range", g.nnx, g.nnz); return DSA_ERR_ARGUMENT; }
    nmaps = nm;
    nfield = (size_t)g.nnx * g.nnz;
    nrec_c = (size_t)g.nbx * g.nbz * kTileRecs;
    const size_t nv = (size_t)g.nx * g.ny;
    const int dicing = g.gdx;
    std::vector<float> cb(4 * (dicing + 1)), rb(4 * (dicing * kSgdl + 1)), rc(g.nnx);
    basis_table(dicing, cb.data());
    basis_table(dicing * kSgdl, rb.data());
    risti_table(g.gox, g.dnx, g.earth, g.nnx, rc.data());
    dpl = min_cell_km(g);
    if (ensure(veln, nfield * nm) || ensure(slow, nrec_c * nm) || ensure(risti_c, rc.size()) ||
        ensure(cbasis, cb.size()) || ensure(rbasis, rb.size())) return status;
    HIP_TRY(this, hipMemcpyAsync(cbasis.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(rbasis.p, rb.data(), rb.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(risti_c.p, rc.data(), rc.size() * 4, hipMemcpyHostToDevice, stream));
    for (int m = 0; m < nm; ++m)
        launch_gridder(g, velv.p + nv * m, cbasis.p, veln.p + nfield * m, slow.p + nrec_c * m, stream);
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipStreamSynchronize(stream));
    planned = false;
    have_maps = true;
    slowI_ready = false;
    bundles_failed = false; refined_bundles_failed = false;
    if (g.nnx != grown_nnx || g.nnz != grown_nnz) { exc_log2cap_grown = 0; grown_nnx = g.nnx; grown_nnz = g.nnz; march_pool_kept = false; }      // (another grid: a kept marching pool is of the wrong size)
    return 0;
}

// ---- dispersion stage -----------------------------------------------------------------------------
int Engine::dispersion_begin(int nx, int ny, int nz, const float* vels, const float* depz, float minthk, int kmax_total, int nmaps_total)
{
    if (nx < 1 || ny < 1 || nz < 2 || nz > 64 || !vels || !depz || kmax_total < 1 || nmaps_total < 1 || !(minthk > 0.0f)) { fail(DSA_ERR_ARGUMENT, "dispersion: bad arguments (nx=%d ny=%d nz=%d kmax=%d)", nx, ny, nz, kmax_total); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    if (make_layer_geom(nz, depz, minthk, h_geom) != 0) { fail(DSA_ERR_ARGUMENT, "dispersion: the refined column exceeds %d layers", kMaxLayers); return DSA_ERR_ARGUMENT; }
    const size_t ncol = (size_t)nx * ny;
    disp_nx = nx; disp_ny = ny; disp_nz = nz; disp_kmax_total = kmax_total; disp_nmaps = nmaps_total;
    h_depz.assign(depz, depz + nz);
    const size_t nsen = ncol * kmax_total * nz;
    if (ensure(geom, 1) || ensure(vels_d, ncol * nz) || ensure(pvstore, ncol * nmaps_total) || ensure(sen_vs, nsen) || ensure(sen_vp, nsen) || ensure(sen_rho, nsen)) return status;
    HIP_TRY(this, hipMemcpyAsync(geom.p, &h_geom, sizeof(LayerGeom), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(vels_d.p, vels, ncol * nz * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemsetAsync(pvstore.p, 0, ncol * nmaps_total * 8, stream));
    HIP_TRY(this, hipMemsetAsync(sen_vs.p, 0, nsen * 8, stream));
    HIP_TRY(this, hipMemsetAsync(sen_vp.p, 0, nsen * 8, stream));
    HIP_TRY(this, hipMemsetAsync(sen_rho.p, 0, nsen * 8, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    disp_ready = true;
    disp_fail_count = 0;
    disp_failures.clear();
    h_vels.assign(vels, vels + ncol * nz);
    have_sens = false;
    stats[DSA_STAT_MS_DISPERSION] = 0.0;
    stats[DSA_STAT_CURVES] = 0.0;
    return 0;
}

// one wave type: nper periods -> maps [map_first, map_first + nper) and, with kernels, depth-kernel
// slots [sen_slot, sen_slot + nper)
int Engine::dispersion_run(int iwave, int igr, int nper, const double* t, int with_kernels, int sen_slot, int map_first)
{
    if (!disp_ready) { fail(DSA_ERR_STATE, "dispersion: call dsa_dispersion_begin first"); return DSA_ERR_STATE; }
    if (nper <= 0) return 0;
    if ((iwave != 1 && iwave != 2) || nper > kMaxPeriods || !t || map_first < 0 || map_first + nper > disp_nmaps ||
        (with_kernels && (sen_slot < 0 || sen_slot + nper > disp_kmax_total))) { fail(DSA_ERR_ARGUMENT, "dispersion: bad arguments (iwave=%d nper=%d map=%d slot=%d)", iwave, nper, map_first, sen_slot); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    const int ncol = disp_nx * disp_ny;
    const int npert = with_kernels ? 1 + 6 * disp_nz : 1;
    const size_t nlanes = (size_t)ncol * npert;
    if (ensure(curves, nlanes * nper) || ensure(disp_ws, (size_t)4 * h_geom.rmax * nlanes) || ensure(tper, kMaxPeriods) || ensure(disp_diag, 2)) return status;
    HIP_TRY(this, hipMemcpyAsync(tper.p, t, (size_t)nper * 8, hipMemcpyHostToDevice, stream));
    const unsigned long long diag0[2] = { 0ull, ~0ull };
    HIP_TRY(this, hipMemcpyAsync(disp_diag.p, diag0, 16, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipEventRecord(events[1], stream));
    // layer tables in LDS when they fit (64 curves x 4 arrays x rmax layers x 4 B <= 64 KB, i.e. rmax <= 64) -- see k_dispersion
    const int in_lds = disp_layers_lds >= 0 ? disp_layers_lds : ((size_t)h_geom.rmax * 1024 <= (size_t)64 * 1024 ? 1 : 0);
    // the curves of a small call share the lanes of a group (k_dispersion): 8 lanes per curve for a few thousand
    // curves, 4 up to 32 k curves, one lane per curve beyond (the group repeats the root search and the matrix products)
    int gshift = 0;
    if (in_lds && disp_group_shift != 0) {
        if (disp_group_shift > 0) gshift = disp_group_shift;
        else gshift = nlanes <= 4096 ? 3 : nlanes <= 32768 ? 2 : 0;      // measured: 324 curves 21.6 -> 7.8 ms, 17 820 curves 22.7 -> 16.9 ms, 944 k curves 166 -> 285 ms
    }
    // (the device lists failing curves in arrival order: room for every curve of the run, so that the FIRST disp_failure_log of the
    // reference's call order can be picked after the host sort -- a list of disp_failure_log entries kept an arbitrary subset)
    if (disp_failure_log > 0 && ensure(disp_fail_list, nlanes)) return status;
    launch_dispersion(iwave, geom.p, vels_d.p, ncol, npert, igr, nper, tper.p, disp_ws.p, nlanes, curves.p, h_geom.rmax, in_lds, gshift, disp_diag.p,
                      disp_failure_log > 0 ? disp_fail_list.p : nullptr, disp_failure_log > 0 ? (int)std::min<size_t>(nlanes, (size_t)0x7fffffff) : 0, stream);
    launch_depth_kernels(vels_d.p, ncol, disp_nz, nper, curves.p, with_kernels, pvstore.p + (size_t)map_first * ncol, sen_vs.p, sen_vp.p, sen_rho.p,
                         disp_kmax_total, sen_slot, stream);
    HIP_TRY(this, hipEventRecord(events[2], stream));
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipStreamSynchronize(stream));
    float ms = 0;
    HIP_TRY(this, hipEventElapsedTime(&ms, events[1], events[2]));
    stats[DSA_STAT_MS_DISPERSION] += ms;
    stats[DSA_STAT_CURVES] += (double)nlanes;
    unsigned long long diag[2];
    HIP_TRY(this, hipMemcpy(diag, disp_diag.p, 16, hipMemcpyDeviceToHost));
    if (diag[0]) {
        if (disp_fail_count == 0) {
            const unsigned long long curve = diag[1] >> 16;
            const int k = (int)(diag[1] & 0xffffull);
            disp_fail_first[0] = iwave; disp_fail_first[1] = igr; disp_fail_first[2] = (int)(curve % (unsigned long long)ncol) + 1;
            disp_fail_first[3] = (int)(curve / (unsigned long long)ncol); disp_fail_first[4] = k;
            disp_fail_period = k >= 1 && k <= nper ? t[k - 1] : 0.0;
        }
        disp_fail_count += (long long)diag[0];
        if (disp_failure_log > 0) {
            // the failing curves of this run in the reference's call order: column by column, the model itself, then its perturbations
            // (CalSurfG.f90:44-150 on one thread; the device reports them in any order)
            const size_t nl = (size_t)std::min<unsigned long long>(diag[0], (unsigned long long)nlanes);
            std::vector<unsigned long long> list(nl);
            HIP_TRY(this, hipMemcpy(list.data(), disp_fail_list.p, nl * 8, hipMemcpyDeviceToHost));
            std::vector<DispFailRec> recs;
            for (unsigned long long v : list) {
                const unsigned long long curve = v >> 8;
                DispFailRec r{};
                r.iwave = iwave; r.igr = igr; r.nper = nper; r.k = (int)(v & 0xffull);
                r.column = (int)(curve % (unsigned long long)ncol) + 1; r.pert = (int)(curve / (unsigned long long)ncol);
                for (int q = 0; q < nper && q < 60; ++q) r.t[q] = t[q];
                recs.push_back(r);
            }
            std::sort(recs.begin(), recs.end(), [](const DispFailRec& a, const DispFailRec& b) { return a.column != b.column ? a.column < b.column : a.pert < b.pert; });
            for (const DispFailRec& r : recs) if ((int)disp_failures.size() < disp_failure_log) disp_failures.push_back(r);
        }
    }
    return 0;
}

// One logged failure (option disp_failure_log), replayed on the host: info = { iwave, igr, column, perturbation, k, periods of the call,
// layers, failures logged }, vals = { t(k), cc, cm, c1 }, table = d, a, b, rho of the flattened layers (4 x 200 floats), c = the roots of
// the periods before k (60 doubles; c[k-1] on: 0 -- the reference prints an element of its array it never assigned there)
int Engine::dispersion_failure(int index, int* info, double* vals, float* table, double* c) const
{
    if (index < 0 || index >= (int)disp_failures.size() || !info || !vals || !table || !c) return DSA_ERR_ARGUMENT;
    const DispFailRec& r = disp_failures[(size_t)index];
    const size_t ncol = (size_t)disp_nx * disp_ny;
    float vs[64];
    for (int k = 0; k < disp_nz; ++k) vs[k] = h_vels[(size_t)k * ncol + (size_t)(r.column - 1)];
    int mmax = 0;
    double ccc[3] = { 0.0, 0.0, 0.0 };
    const int k = disp_replay_failure(h_geom, vs, r.pert, r.iwave, r.igr, r.nper, r.t, &mmax, table, ccc, c);
    info[0] = r.iwave; info[1] = r.igr; info[2] = r.column; info[3] = r.pert; info[4] = r.k; info[5] = r.nper; info[6] = mmax; info[7] = (int)disp_failures.size();
    vals[0] = r.k >= 1 && r.k <= r.nper ? r.t[r.k - 1] : 0.0; vals[1] = ccc[0]; vals[2] = ccc[1]; vals[3] = ccc[2];
    return k == r.k ? 0 : DSA_ERR_INTERNAL;          // (the replay must fail where the device did)
}

int Engine::dispersion_copy_map(int from, int to, int n)
{
    if (!disp_ready || from < 0 || to < 0 || n < 0 || from + n > disp_nmaps || to + n > disp_nmaps) { fail(DSA_ERR_ARGUMENT, "dispersion: bad map copy"); return DSA_ERR_ARGUMENT; }
    const size_t ncol = (size_t)disp_nx * disp_ny;
    HIP_TRY(this, hipSetDevice(device));
    if (n) HIP_TRY(this, hipMemcpyAsync(pvstore.p + (size_t)to * ncol, pvstore.p + (size_t)from * ncol, (size_t)n * ncol * 8, hipMemcpyDeviceToDevice, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    return 0;
}

// host copies in the reference's per-type layouts: pv(ncol, nper), sen(ncol, nper, nz)
int Engine::dispersion_fetch(int map_first, int nper, double* pv, int with_kernels, int sen_slot, double* svs, double* svp, double* srho)
{
    if (!disp_ready || nper < 0 || map_first < 0 || map_first + nper > disp_nmaps) { fail(DSA_ERR_ARGUMENT, "dispersion: bad fetch"); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    const size_t ncol = (size_t)disp_nx * disp_ny;
    if (pv && nper) HIP_TRY(this, hipMemcpy(pv, pvstore.p + (size_t)map_first * ncol, (size_t)nper * ncol * 8, hipMemcpyDeviceToHost));
    if (with_kernels && nper) {
        if (sen_slot < 0 || sen_slot + nper > disp_kmax_total || !svs || !svp || !srho) { fail(DSA_ERR_ARGUMENT, "dispersion: bad kernel fetch"); return DSA_ERR_ARGUMENT; }
        double* dst[3] = { svs, svp, srho };
        const double* srcs[3] = { sen_vs.p, sen_vp.p, sen_rho.p };
        for (int q = 0; q < 3; ++q)
            for (int i = 0; i < disp_nz; ++i)
                HIP_TRY(this, hipMemcpy(dst[q] + (size_t)i * nper * ncol, srcs[q] + ((size_t)i * disp_kmax_total + sen_slot) * ncol, (size_t)nper * ncol * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

int Engine::maps_from_dispersion(float goxd, float gozd, float dvxd, float dvzd, int dicing)
{
    if (!disp_ready) { fail(DSA_ERR_STATE, "maps: call dsa_dispersion_begin / run first"); return DSA_ERR_STATE; }
    if (disp_nx < 4 || disp_ny < 4 || dicing < 1 || dicing > 16) { fail(DSA_ERR_ARGUMENT, "maps: bad arguments"); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    make_grid(g, disp_nx, disp_ny, goxd, gozd, dvxd, dvzd, dicing);
    const size_t n = (size_t)disp_nx * disp_ny * disp_nmaps;
    if (ensure(velv, n)) return status;
    launch_to_float(pvstore.p, velv.p, n, stream);
    std::vector<float> hv(n);
    HIP_TRY(this, hipMemcpyAsync(hv.data(), velv.p, n * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    hmin_slow = 1e30f;
    for (float v : hv) if (v > 0.0f && 1.0f / v < hmin_slow) hmin_slow = 1.0f / v;
    if (!(hmin_slow < 1e30f)) { fail(DSA_ERR_INTERNAL, "maps: no positive phase velocity came out of the dispersion stage"); return DSA_ERR_INTERNAL; }
    mean_slowness_of_maps(hv.data(), (size_t)disp_nx * disp_ny, disp_nmaps);
    return finish_maps(disp_nmaps);
}

int Engine::kernels_from_dispersion()
{
    if (!disp_ready) { fail(DSA_ERR_STATE, "depth kernels: call dsa_dispersion_begin / run first"); return DSA_ERR_STATE; }
    HIP_TRY(this, hipSetDevice(device));
    const size_t ncol = (size_t)disp_nx * disp_ny;
    if (ensure(Srow, ncol * disp_kmax_total * (disp_nz - 1))) return status;
    launch_sen_combine((int)ncol, disp_kmax_total, disp_nz, vels_d.p, sen_vs.p, sen_vp.p, sen_rho.p, h_depz[disp_nz - 2] < 35.0f ? 1 : 0, Srow.p, stream);
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipStreamSynchronize(stream));
    sens_nz = disp_nz; sens_kmax = disp_kmax_total; have_sens = true;
    return 0;
}

int Engine::plan(int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec,
                 const float* rcx, const float* rcz, const int* mode, const int* sen_slot, const int* data_first)
{
    if (!have_maps) { fail(DSA_ERR_STATE, "plan: call dsa_set_maps first"); return DSA_ERR_STATE; }
    if (nunits < 0 || (nunits > 0 && (!map_index || !scx || !scz || !nrec))) { fail(DSA_ERR_ARGUMENT, "plan: bad arguments"); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    release_march_pool();
    h_src.resize(nunits);
    h_unit_reach_km.assign((size_t)nunits, 0.0f);
    h_risti_r.assign((size_t)nunits * kRefMax, 1.0f);
    size_t nr = 0;
    for (int u = 0; u < nunits; ++u) {
        SourceDesc& s = h_src[u];
        if (map_index[u] < 0 || map_index[u] >= nmaps) { fail(DSA_ERR_ARGUMENT, "plan: unit %d uses map %d of %d", u, map_index[u], nmaps); return DSA_ERR_ARGUMENT; }
        if (make_source(g, scx[u], scz[u], s) != 0) {
            fail(DSA_ERR_OUTSIDE, "Source lies outside bounds of model (lat,long)= %g %g", 90.0 - scx[u] * 180.0 / kPi, scz[u] * 180.0 / kPi);
            return DSA_ERR_OUTSIDE;
        }
        s.period = map_index[u];
        s.first_ray = (int)nr;
        s.nrec = nrec[u];
        s.sen_slot = sen_slot ? sen_slot[u] : 0;
        if (nrec[u] < 0) { fail(DSA_ERR_ARGUMENT, "plan: negative receiver count"); return DSA_ERR_ARGUMENT; }
        nr += (size_t)nrec[u];
        risti_table(s.rgox, s.rdnx, g.earth, s.rnx, &h_risti_r[(size_t)u * kRefMax]);
    }
    if (nr > 0 && (!rcx || !rcz)) { fail(DSA_ERR_ARGUMENT, "plan: receivers missing"); return DSA_ERR_ARGUMENT; }
    h_rays.resize(nr);
    h_trace.clear();
    ndata = 0;
    for (int u = 0; u < nunits; ++u)
        for (int k = 0; k < h_src[u].nrec; ++k) {
            const size_t r = (size_t)h_src[u].first_ray + k;
            const float rx = rcx[r], rz = rcz[r];
            const int irx = (int)((rx - g.gox) / g.dnx) + 1, irz = (int)((rz - g.goz) / g.dnz) + 1;
            if (irx < 1 || irx > g.nnx || irz < 1 || irz > g.nnz) {
                fail(DSA_ERR_OUTSIDE, "Receiver lies outside model (lat,long)= %g %g", 90.0 - rx * 180.0 / kPi, rz * 180.0 / kPi);
                return DSA_ERR_OUTSIDE;
            }
            const int fl = mode ? (mode[u] & (kRayTime | kRayPath)) : (kRayTime | kRayPath);
            const int data = (data_first ? data_first[u] : h_src[u].first_ray) + k;
            if (data < 0 || (k == 0 && u > 0 && h_src[u - 1].nrec > 0 && data < h_rays[h_src[u - 1].first_ray].data)) { fail(DSA_ERR_ARGUMENT, "plan: data indices must be non-negative and non-decreasing (unit %d)", u); return DSA_ERR_ARGUMENT; }
            h_rays[r] = RayDesc{ u, rx, rz, sinf(rx), data, fl };
            if (fl & (kRayTime | kRayPath)) {      // the unit's reach: the great-circle distance to its farthest receiver (km; the scale of its travel times, tie_verdicts)
                const float dc = rx - h_src[u].scx, dl = (rz - h_src[u].scz) * sinf(0.5f * (rx + h_src[u].scx));
                const float km = g.earth * sqrtf(dc * dc + dl * dl);
                if (km > h_unit_reach_km[(size_t)u]) h_unit_reach_km[(size_t)u] = km;
            }
            if (fl & kRayPath) h_trace.push_back((int)r);
            ndata = std::max(ndata, (size_t)data + 1);
        }
    // chunk size from the memory budget
    size_t free_b = 0, total_b = 0;
    HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
    // (what an earlier plan / solve of this engine allocated for the same purpose is reused, not needed again: without this the second
    // call of an inversion saw half the memory and cut its units into two launches -- and its sources' periods into two bundles each)
    free_b += solve_stage_bytes();
    // default: 60 % of free HBM, at most 150 GB per chunk of sources (one launch for the 16 000 units of the
    // headline configuration: +3 % over two launches, whose tails leave CUs idle)
    size_t budget = mem_budget ? mem_budget : std::min<size_t>((size_t)(0.6 * (double)free_b), (size_t)150 << 30);
    const size_t rr = (size_t)kRefMax * kRefMax;
    {   // the refined and the coarse solve of a unit share one list region: size it for the larger shape
        shape_c = launch_shape(g.nnx, g.nnz); shape_r = launch_shape(kRefMax, kRefMax);
        shape_c.sorted = shape_c.tile_words * 4 <= 36 * 1024 ? 1 : 0;      // the coarse solve runs on the compact field: ordered variant only
        if (!shape_c.sorted) { fail(DSA_ERR_ARGUMENT, "plan: a %d x %d grid needs %d bytes of LDS tile bitmap (limit 36 KB): the coarse solve has no other variant", g.nnx, g.nnz, shape_c.tile_words * 4); return DSA_ERR_ARGUMENT; }
        shape_c.compact = 1; shape_r.compact = 0;
        exc_log2cap = std::max(exc_log2cap_opt > 0 ? exc_log2cap_opt : exc_log2cap_of(g.nnx, g.nnz), exc_log2cap_grown);
        const FimLaunch &lr = shape_r;
        lists_stride = std::max((size_t)4 * lr.list_cap + lr.ready_cap, (size_t)kFimMaskInts * kRefTiles * kRefTiles + 2);      // refined solve, per unit: list or tile-record variant
        lists_stride = (lists_stride + 1) & ~(size_t)1;      // the masks are 8-byte words
        lists_c_stride = ((size_t)kFimMaskInts * g.nbx * g.nbz + 2 + 1) & ~(size_t)1;                                             // coarse solve, per field slot: tile records
    }
    // what lives per field slot (the coarse field, its exception table, its tile records) and what lives per unit of a launch
    per_slot_bytes = nrec_c * 4 + ((size_t)8 << exc_log2cap) + lists_c_stride * 4;
    per_unit_bytes = (size_t)kCWinMax * kCWinMax * 8 + lists_stride * 4 + (size_t)kRefRecs * 12 + rr * 5 + kRefMax * 4 + (size_t)kSeedR * 4 + (size_t)kSeedC * 4 + kRWin * kRWin * 2 +
                     (size_t)kCWinMax * kCWinMax * 3 + kHeapCap * 4 + 256 + sizeof(FimProblem) * 2 + sizeof(FimEnds) + sizeof(SourceDesc);
    {
        // field slots: by default four times the workgroups the chip holds at once (256 CUs, up to four workgroups each), never more than the units,
        // within 70 % of the budget; keep_fields or option field_pool = -1: a slot per unit
        const int resident = 256 * std::max(1, std::min(4, 1024 / std::max(shape_c.threads, 1)));
        // (small grids keep many more workgroups in flight than the estimate, and their slots cost little: up to 16 384 slots within 16 GB --
        // measured at 121^2, 8000 units: 186 k solves/s with a slot per unit, 75 k through 1024 slots, profiles/r03_recycle_stress.log)
        const size_t roomy = std::min<size_t>(16384, ((size_t)16 << 30) / per_slot_bytes);
        size_t P = field_pool_opt > 0 ? (size_t)field_pool_opt : std::max<size_t>((size_t)4 * resident, roomy);
        // units that will be solved inside bundles (bundle_kernel.hip) need no slot of their own unless rays follow (solve() grows the
        // pool then): the pool shrinks to what the left-over units can use, and the memory goes to the bundle fields
        if (field_pool_opt == 0 && !keep_fields && exact_ties != 2) {
            long solo_units = 0;
            if (choose_bundle_size(nunits, &solo_units) > 0) P = std::min<size_t>(P, (size_t)std::max<long>(256, 2 * solo_units));
        }
        // exact_ties = 2: the march has fields of its own and writes its units' receiver times itself; the compact slots only serve
        // dsa_get_field / rays afterwards (a launch with more units than slots leaves none behind, as with recycled slots): 16 GB of them at most
        if (field_pool_opt == 0 && !keep_fields && exact_ties == 2) P = std::min<size_t>(P, std::max<size_t>(1, ((size_t)16 << 30) / per_slot_bytes));
        if (field_pool_opt < 0 || keep_fields) P = (size_t)std::max(nunits, 1);
        P = std::min<size_t>(P, (size_t)std::max(nunits, 1));
        if (max_chunk > 0) P = std::min<size_t>(P, (size_t)max_chunk);
        const size_t fit = (size_t)(0.7 * (double)budget) / per_slot_bytes;
        if (fit < 1) { fail(DSA_ERR_DEVICE, "memory budget %zu B cannot hold one coarse field slot (%zu B)", budget, per_slot_bytes); return DSA_ERR_DEVICE; }
        if (keep_fields && fit < P) { fail(DSA_ERR_CAPACITY, "plan: keep_fields is set but only %zu of the %d units' fields fit the memory budget", fit, nunits); return DSA_ERR_CAPACITY; }
        pool_slots = (int)std::min(P, fit);
    }
    plan_budget = budget;
    size_t c = (budget - (size_t)pool_slots * per_slot_bytes) / per_unit_bytes;
    if (c < 1) { fail(DSA_ERR_DEVICE, "memory budget %zu B cannot hold one source (%zu B per unit + %zu B per field slot)", budget, per_unit_bytes, per_slot_bytes); return DSA_ERR_DEVICE; }
    chunk = (int)std::min<size_t>(c, (size_t)std::max(nunits, 1));
    if (max_chunk > 0) chunk = std::min(chunk, max_chunk);
    if (keep_fields && (chunk < nunits || pool_slots < nunits)) { fail(DSA_ERR_CAPACITY, "plan: keep_fields is set but only %d of the %d units fit one resident chunk (memory budget / max_chunk)", std::min(chunk, pool_slots), nunits); return DSA_ERR_CAPACITY; }
    const size_t C = (size_t)chunk;
    const size_t PS = (size_t)pool_slots;
    if (ensure(lists, C * lists_stride) || ensure(lists_c, PS * lists_c_stride) || ensure(pool_gen, PS) || ensure(ends_c, C) || ensure(src, C) || ensure(rays, std::max<size_t>(nr, 1)) || ensure(out, std::max<size_t>(ndata, 1)) || ensure(trace_ids, std::max<size_t>(h_trace.size(), 1)) || ensure(err, 4) ||
        ensure(slow_r, C * kRefRecs) || ensure(F_r, C * kRefRecs) || ensure(Tfin_r, C * rr) || ensure(S_r, C * rr) ||
        ensure(risti_r, C * kRefMax) || ensure(vcorner, C * 4) || ensure(seed_r, C * kSeedR) || ensure(nseed_r, C) ||
        ensure(rst, C * kRWin * kRWin) || ensure(cst, C * kCWinMax * kCWinMax) || ensure(cinit, C * kCWinMax * kCWinMax) ||
        ensure(heap, C * kHeapCap) || ensure(flags, C * 4) || ensure(T_c, PS * nrec_c) || ensure(exc_c, PS << exc_log2cap) || ensure(W_c, C * kCWinMax * kCWinMax) ||
        ensure(seed_c, C * kSeedC) || ensure(nseed_c, C) || ensure(launch_rank, C) || ensure(prob_r, C) || ensure(prob_c, C) || ensure(info, C * 16) || ensure(clocks, C * kClockSlots) ||
        ensure(tieinfo, C * kTieWords) || ensure(xinfo, C * 4) || ensure(x_units, C) ||
        ensure(replay_list, kHandoffReplayCap + 1) || ensure(replay_scratch, (size_t)kHandoffReplayCap * handoff_replay_bytes())) return status;
    h_unit_flags.assign((size_t)nunits, 0);
    h_unit_tie.assign((size_t)nunits, 0.0f);
    h_unit_tie_sum.assign((size_t)nunits, 0.0f); h_unit_tie_count.assign((size_t)nunits, 0); h_unit_froze.assign((size_t)nunits, 0);
    h_unit_rounds.assign((size_t)nunits, 0);
    HIP_TRY(this, hipMemsetAsync(out.p, 0, std::max<size_t>(ndata, 1) * sizeof(float), stream));     // data indices without a kRayTime ray read as 0
    if (nr) HIP_TRY(this, hipMemcpyAsync(rays.p, h_rays.data(), nr * sizeof(RayDesc), hipMemcpyHostToDevice, stream));
    if (!h_trace.empty()) HIP_TRY(this, hipMemcpyAsync(trace_ids.p, h_trace.data(), h_trace.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    planned = true;
    last_chunk_first = -1; fields_resident = false;
    rays_clamped = 0; first_clamped_unit = -1;      // (diagnostics belong to the plan they were measured on)
    return 0;
}

// The second stream of a launch: the bundles beyond the first generation beside it (plan_bundles).
hipError_t Engine::make_stream2()
{
    hipError_t rc = hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking);
    if (rc != hipSuccess) return rc;
    rc = hipEventCreateWithFlags(&ev_b0, hipEventDisableTiming);
    if (rc != hipSuccess) return rc;
    return hipEventCreateWithFlags(&ev_b1, hipEventDisableTiming);
}

// An exact_ties = 2 call keeps its marching pool -- 80 % of the free memory -- for the next one; a plan or a solve in another mode needs that
// memory for its own fields (at 4097^2 the bundles found none after an exact_ties = 2 call and the solve ran unit by unit, 5.3 s instead of 1.9).
void Engine::release_march_pool()
{
    if (exact_ties == 2 || (!X_pool.cap && !X_tp.cap)) return;
    if (exact_ties == 1 && march_pool_kept) return;          // (round 6: kept by the last march because the device had room for it beside everything else)
    (void)hipStreamSynchronize(stream);
    auto release = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    release(X_pool); release(X_heap); release(X_tt); release(X_tp); release(X_ring); release(X_free); release(X_pins);
    march_pool_kept = false;
}

// The unit field pool back to its regular size (four times the resident workgroups, within the plan's budget) when plan() had cut it
// down for a call it expected to bundle and the call runs unit by unit after all: through a pool of 256 slots a unit-by-unit launch
// crawls (75 k against 186 k solves/s at 121^2).  Returns false on an allocation error.
bool Engine::grow_unit_pool()
{
    if (field_pool_opt != 0 || keep_fields || exact_ties == 2) return true;
    const int resident = 256 * std::max(1, std::min(4, 1024 / std::max(shape_c.threads, 1)));
    const size_t roomy = std::min<size_t>(16384, ((size_t)16 << 30) / per_slot_bytes);
    size_t P = std::min<size_t>(std::max<size_t>((size_t)4 * resident, roomy), (size_t)std::max<size_t>(h_src.size(), 1));
    if (max_chunk > 0) P = std::min<size_t>(P, (size_t)max_chunk);
    const size_t units_b = (size_t)chunk * per_unit_bytes;
    const size_t fit = plan_budget > units_b ? (size_t)(0.7 * (double)(plan_budget - units_b)) / per_slot_bytes : 0;
    P = std::min(P, fit);
    if (P <= (size_t)pool_slots) return true;
    if (ensure(T_c, P * nrec_c) || ensure(exc_c, P << exc_log2cap) || ensure(lists_c, P * lists_c_stride) || ensure(pool_gen, P)) return false;
    pool_slots = (int)P;
    return true;
}

// bytes held by the buffers plan() and solve() size (reused by the next plan)
size_t Engine::solve_stage_bytes() const
{
    auto b = [](const auto& d) { return d.cap * sizeof(*d.p); };
    return b(lists) + b(lists_c) + b(pool_gen) + b(ends_c) + b(src) + b(slow_r) + b(F_r) + b(Tfin_r) + b(S_r) + b(risti_r) + b(vcorner) + b(seed_r) + b(nseed_r) +
           b(rst) + b(cst) + b(cinit) + b(heap) + b(flags) + b(T_c) + b(exc_c) + b(W_c) + b(seed_c) + b(nseed_c) + b(launch_rank) + b(prob_r) + b(prob_c) +
           b(info) + b(clocks) + b(tieinfo) + b(xinfo) + b(x_units) + b(B_pool) + b(exc_b) + b(lists_b) + b(bpool_gen) + b(bundles_d) + b(member_flag);
}

// bundles the chip holds at a time: one workgroup of 512 threads per CU, two of 256 (four members per lane: 204 VGPRs) or three (two
// members per lane: 168 VGPRs, round 4)
size_t Engine::bundles_resident(int G, int mpl, int threads) const
{
    if (threads <= 0) threads = bundle_threads();
#ifdef DSA_BUNDLE_WAVES          // (probe builds: bundle_kernel.hip compiled for that many waves per SIMD whatever the members per lane)
    if (threads == 256) return (size_t)256 * DSA_BUNDLE_WAVES;
#endif
    // (bundle_kernel.hip: DSA_BUNDLE_OCC -- three workgroups of 256 threads per CU with two members per lane and for bundles of 16)
    return (size_t)256 * (threads >= 512 ? 1 : (mpl == 2 || G == 16) ? 3 : 2);
}

// Members per lane of a launch of nb bundles of G: the option, or 16 members four per lane; 8 and 4 members two per lane (three workgroups
// per CU) when the launch holds more than the 512 bundles that two per CU take, else four
int Engine::bundle_mpl_of(int G, long nb) const
{
    if (bundle_mpl) return bundle_mpl;
    if (bundle_threads() != 256 || G == 16) return 4;
    return nb > 512 ? 2 : 4;
}

// Causal window of the bundles in cells: the option (default 0.6: at full occupancy the evaluations are what costs); a small launch of
// small bundles, wide with a CU per bundle, is bound by the length of its rounds' chain instead: 1.25 cells there (24 % fewer
// rounds, 15 % more evaluations: 250 bundles of 8 at 1025^2 102.2 -> 96.8 ms; bundles of 16 show no difference; profiles/r04_bundle_occupancy.log)
float Engine::bundle_window() const
{
    if (bundle_window_opt > 0.0f) return bundle_window_opt;
    // (round 5, profiles/r05_ab_windows.log: wide bundles of 16 -- 250 sources: 0.6 / 1.0 / 1.5 / 2.5 cells 134.4 / 129.6 / 131.9 / 133.0 ms;
    // wide bundles of 8 -- 125 sources: 1.25 / 1.75 / 2.5 / 4.0 cells 97.8 / 96.0 / 95.6 / 111.5 ms)
    if (bundle_wide && bundle_G_now > 0) return bundle_G_now < 16 ? 1.75f : 1.0f;
    return 0.6f;
}

// ... and of the bundles that run 768 threads wide behind the first generation of a large launch (plan_bundles: member flag 2)
float Engine::bundle_window_tail() const
{
    if (bundle_window_opt > 0.0f) return bundle_window_opt;
    return bundle_threads_b == 768 && bundle_threads() == 256 ? 1.0f : bundle_window();
}

// Workgroup size of the bundle kernel: the option; 256 threads, three workgroups per CU, as a rule.  Where a bundle has a CU to itself
// anyway it runs 768 threads WIDE -- twelve waves, three per SIMD, share a round's node trips instead of four: beyond 1500 nodes per
// side (a 4097^2 front does not fit the ready lists of 256 threads, and a bundle's field slot is so large that a CU's worth of them
// fills the memory), and on smaller grids when choose_bundle_size found the launch so small that every bundle gets a CU (bundle_wide:
// at most 256 bundles).  Round 4, profiles/r04_bundle_occupancy.log: 250 bundles of 8 at 1025^2 131 ms with 256 threads, 102 with 512,
// 95 with 768; 256 bundles of 8 at 4097^2 1 229 ms with 512, 1 122 with 768.
int Engine::bundle_threads() const
{
    if (bundle_threads_opt == 256 || bundle_threads_opt == 512 || bundle_threads_opt == 768) return bundle_threads_opt;
    if (std::max(g.nnx, g.nnz) > 1500) return 768;
    return bundle_wide ? 768 : 256;
}

BatchPtrs Engine::batch() const
{
    BatchPtrs b;
    b.src = src.p; b.slow_r = slow_r.p; b.F_r = F_r.p; b.Tfin_r = Tfin_r.p; b.S_r = S_r.p; b.risti_r = risti_r.p;
    b.vcorner = vcorner.p; b.seed_r = seed_r.p; b.nseed_r = nseed_r.p; b.rst = rst.p; b.cst = cst.p; b.cinit = cinit.p;
    b.heap = heap.p; b.flags = flags.p; b.T_c = T_c.p; b.exc_c = exc_c.p; b.exc_log2cap = exc_log2cap; b.W_c = W_c.p; b.seed_c = seed_c.p; b.nseed_c = nseed_c.p;
    b.lists = lists.p; b.lists_stride = lists_stride;
    b.lists_c = lists_c.p; b.lists_c_stride = lists_c_stride; b.pool = pool_slots; b.pool_gen = pool_gen.p;
    return b;
}

// The census' verdict on one unit (round 6): the unit's tie record (kernels.h: kTieWords, the refined stage's half first) and solve info into the
// per-unit arrays dsa_unit_ties / dsa_unit_tie_sums report; true = the unit goes to the march (exact_ties = 1) / counts as flagged (exact_ties = 0).
// Flag rule: a tie whose influence exceeds tie_threshold (counted by the kernels), OR the influences of all its ties adding up to more than
// tie_sum_threshold, OR more than tie_count_threshold ties with an influence, OR a frozen cycle (unit-by-unit solves: always; members of a
// bundle that froze one: option tie_frozen_bundles).
// The verdicts of one launch (units first .. first + n - 1).  Round 6: a map on which some unit holds a tie above tie_threshold is TIE-PRONE -- a medium with
// sharp contrasts, where second-order stencils switch along ridges and a one-ulp difference grows downstream (profiles/r06_tie_*: the units the
// per-unit rule leaves alone end beyond 1e-4 s at a rate of ~1 in 10 000 there, whatever their own largest, summed or counted influences; on a map
// without such a tie none of 16 000 did) -- and on a tie-prone map every unit that holds a tie with any influence at all goes to the march
// (option tie_map_strict, default on).  The maps are judged launch by launch: a call that fits one launch -- the rule -- is judged as a whole.
std::vector<char> Engine::tie_verdicts(int first, int n, const int32_t* tie_words, const int32_t* info16, bool bundled)
{
    std::vector<char> fl((size_t)n, 0);
    std::vector<char> prone((size_t)std::max(nmaps, 1), 0);
    for (int u = 0; u < n; ++u) {
        const bool member = bundled && h_member_flag[(size_t)u] != 0;
        const int code = tie_verdict(first + u, tie_words + (size_t)u * kTieWords, info16 + (size_t)u * 16, member);
        fl[(size_t)u] = code != 0;
        const int p = h_src[(size_t)(first + u)].period;
        if (code == 2 && p >= 0 && p < nmaps) prone[(size_t)p] = 1;
    }
    if (tie_map_strict)
        for (int u = 0; u < n; ++u) {
            const int p = h_src[(size_t)(first + u)].period;
            // (a member of a bundle that froze a cycle counts here too: a frozen 2-cycle sits an ulp from a tie state the census of the frozen field cannot see)
            if (!fl[(size_t)u] && p >= 0 && p < nmaps && prone[(size_t)p] && (h_unit_tie_count[(size_t)(first + u)] > 0 || h_unit_froze[(size_t)(first + u)] > 0)) { fl[(size_t)u] = 1; stats[DSA_STAT_TIE_UNITS_STRICT] += 1.0; }
        }
    // The validated envelope of "small ties stay with the fixed point" (round 6, late).  What the fixed point's field differs by from the reference's
    // downstream of one-ulp ties is a number of ULPS of the travel time, growing with the grid: at most 26 ulps at a receiver on grids up to 1025
    // nodes per side (2 M fuzzed units: 9.5e-5 s at 32-64 s), 35 at 2049^2 (2.7e-4 s), 110 at 4097^2 (1.7e-3 s; profiles/r06_tie_scale_*.log).
    // Against the absolute bar that is a statement about the TIMES: a unit whose farthest receiver lies beyond
    // tie_tolerance / (26 ulps x max(N, 1025) / 1025) -- 64 s on grids up to 1025^2, 32 s at 2049^2, 16 s at 4097^2 -- is outside what was
    // measured, and is marched if it holds a tie with any influence, like a unit on a tie-prone map.  The unit's time scale: its reach (plan) x the
    // mean slowness of its map, plus a tenth (option tie_scale_guard, default on).
    if (tie_scale_guard && exact_ties != 2) {
        const float N = (float)std::max(g.nnx, g.nnz);
        const float ulps = kTieUlpsAt1025 * std::max(N, 1025.0f) / 1025.0f;
        for (int u = 0; u < n; ++u) {
            if (fl[(size_t)u]) continue;
            const size_t gu = (size_t)(first + u);
            if (!(h_unit_tie_count[gu] > 0 || h_unit_froze[gu] > 0)) continue;
            const int p = h_src[gu].period;
            // (raised by a tenth: a path may run through parts of the map slower than its average)
            const float t_scale = 1.1f * h_unit_reach_km[gu] * (p >= 0 && p < (int)map_mean_slow.size() ? map_mean_slow[(size_t)p] : hmin_slow);
            int e2 = 0;
            (void)std::frexp(std::max(t_scale, 1e-30f), &e2);            // t_scale in [2^(e2-1), 2^e2): ulp = 2^(e2-24)
            if (ulps * std::ldexp(1.0f, e2 - 24) > tie_tolerance) { fl[(size_t)u] = 1; stats[DSA_STAT_TIE_UNITS_BY_SCALE] += 1.0; }
        }
    }
    if (getenv("DSA_DEBUG_TIES")) {      // (probe: what flagged the launch's units)
        long big = 0, hand = 0, band = 0, froze = 0, any = 0;
        const int h = kTieWords / 2;
        for (int u = 0; u < n; ++u) {
            const int32_t* t = tie_words + (size_t)u * kTieWords;
            big += t[0] > 0 || t[h] > 0; hand += t[6] > 0; band += t[h + 6] > 0; froze += h_unit_froze[(size_t)(first + u)] > 0; any += fl[(size_t)u] != 0;
        }
        fprintf(stderr, "dsa ties: %d units: flagged %ld (a tie above the threshold %ld, hand-off status %ld, band march's tree %ld, frozen cycle %ld)\n", n, any, big, hand, band, froze);
    }
    for (char c : prone) stats[DSA_STAT_TIE_PRONE_MAPS] += c ? 1.0 : 0.0;
    for (int u = 0; u < n; ++u) if (!fl[(size_t)u] && h_unit_tie_count[(size_t)(first + u)] > 0) stats[DSA_STAT_TIE_UNITS_TIED] += 1.0;
    return fl;
}

// returns 0: not flagged, 1: flagged, 2: flagged by a tie above tie_threshold (what makes the unit's map tie-prone)
int Engine::tie_verdict(int unit, const int32_t* t, const int32_t* inf, bool member)
{
    const int h = kTieWords / 2;
    float a, c2;
    std::memcpy(&a, t + 1, 4); std::memcpy(&c2, t + h + 1, 4);
    h_unit_tie[(size_t)unit] = std::max(a, c2);
    const double sum = ((double)(uint32_t)t[3] + (double)(uint32_t)t[h + 3]) * (double)kTieSumUnit;
    const long cnt = (long)(uint32_t)t[2] + (long)(uint32_t)t[h + 2];
    h_unit_tie_sum[(size_t)unit] = (float)sum;
    h_unit_tie_count[(size_t)unit] = (int)std::min<long>(cnt, 0x7fffffff);
    const int froze = inf[3] + inf[11] + (member ? t[4] + t[h + 4] : 0);
    h_unit_froze[(size_t)unit] = froze;
    const bool frozen = inf[3] > 0 || (!member && inf[11] > 0) || (member && tie_frozen_bundles && (t[4] > 0 || t[h + 4] > 0));
    if (t[0] > 0 || t[h] > 0) return 2;
    if (t[6] > 0 || t[h + 6] > 0) return 1;            // (a rank tie at the hand-off that changes what the coarse grid receives: k_handoff_probe; a band march that left its tree no heap: k_coarse_march)
    return (frozen || (tie_sum_threshold > 0.0f && sum > (double)tie_sum_threshold) || (tie_count_threshold > 0 && cnt > tie_count_threshold)) ? 1 : 0;
}

int Engine::solve(float* dsurf, float* rw, int* iw, int* col, long long cap, long long* nar)
{
    if (!planned) { fail(DSA_ERR_STATE, "solve: call dsa_plan first"); return DSA_ERR_STATE; }
    const bool grow = grow_rw && grow_iw && grow_col;
    const bool rows = ((rw && iw && col) || grow || rows_on_device) && nar;
    if (rows && !have_sens) { fail(DSA_ERR_STATE, "solve: Frechet rows need the depth kernels (dsa_set_depth_kernels / dsa_depthkernel) first"); return DSA_ERR_STATE; }
    rays_clamped = 0; first_clamped_unit = -1;
    if (rows) {
        *nar = 0;
        G_nar = 0;
        for (const SourceDesc& s : h_src)
            if (s.sen_slot < 0 || s.sen_slot >= sens_kmax) { fail(DSA_ERR_ARGUMENT, "solve: a unit uses depth-kernel slot %d of %d", s.sen_slot, sens_kmax); return DSA_ERR_ARGUMENT; }
    }
    HIP_TRY(this, hipSetDevice(device));
    const int nunits = (int)h_src.size();
    { const double keep_ms = stats[DSA_STAT_MS_DISPERSION], keep_n = stats[DSA_STAT_CURVES];
      std::fill(stats, stats + DSA_STAT_COUNT, 0.0);
      stats[DSA_STAT_MS_DISPERSION] = keep_ms; stats[DSA_STAT_CURVES] = keep_n; }
    std::fill(phase_ticks, phase_ticks + kClockSlots, 0.0);
    release_march_pool();
    marched_in_tiles = false;
    std::fill(h_unit_flags.begin(), h_unit_flags.end(), (unsigned char)0);      // (per solve: a unit marched by an earlier call with other options is not "marched")
    std::fill(h_unit_tie.begin(), h_unit_tie.end(), 0.0f);
    std::fill(h_unit_tie_sum.begin(), h_unit_tie_sum.end(), 0.0f); std::fill(h_unit_tie_count.begin(), h_unit_tie_count.end(), 0); std::fill(h_unit_froze.begin(), h_unit_froze.end(), 0);
    // units per launch: with recycled field slots a launch takes every unit the per-unit arrays hold; when the fields are needed after
    // the solve (rays and rows, the exact mode, keep_fields) a launch takes one unit per slot
    // (round 5: exact_ties = 1 runs like exact_ties = 0 -- recycled slots, receiver times from the solve's own field -- and the units the
    // detector flags are marched afterwards from scratch, their receiver times written again from the marched fields)
    const bool may_recycle = !rows && exact_ties != 2 && !keep_fields;
    // exact_ties = 2 with nothing wanted but receiver times: every unit the per-unit arrays hold in one launch, times from the marched fields
    const bool exact_fused = exact_ties == 2 && !rows && !keep_fields;
    if (!may_recycle && !exact_fused && field_pool_opt == 0 && pool_slots < std::min(chunk, nunits)) {
        // every field of a launch is needed after it (rays and rows, the literal march): a slot per unit for as many units as the
        // budget holds, so that the call is one launch if it can be (four launches of 4096 have four tails: +4 % on the headline call)
        const size_t units_b = (size_t)chunk * per_unit_bytes;
        const size_t fit = plan_budget > units_b ? (plan_budget - units_b) / per_slot_bytes : 0;
        const size_t want = std::min<size_t>((size_t)std::min(chunk, nunits), fit);
        if (want > (size_t)pool_slots) {
            if (ensure(T_c, want * nrec_c) || ensure(exc_c, want << exc_log2cap) || ensure(lists_c, want * lists_c_stride) || ensure(pool_gen, want)) return status;
            pool_slots = (int)want;
        }
    }
    const int step = (may_recycle || exact_fused) ? chunk : std::min(chunk, pool_slots);
    const bool fused_times = exact_ties != 2;          // the coarse solve writes its unit's receiver times itself
    stats[DSA_STAT_UNITS] = nunits;
    stats[DSA_STAT_CHUNK] = step;
    stats[DSA_STAT_FIELD_SLOTS] = pool_slots;
    stats[DSA_STAT_FOOTPRINT_MB] = ((double)pool_slots * (double)per_slot_bytes + (double)chunk * (double)per_unit_bytes) / 1.0e6;
    // causal window: a few cells' worth of travel time (narrowest cell, fastest velocity of the model)
    const float cell_c = dpl * hmin_slow;
    const float window_c = window_cells * cell_c;
    // the refined boxes are small (129^2): a wider window there costs nothing and saves rounds
    const float window_r = std::max(window_cells, 1.5f) * cell_c / (float)kSgdl;
    HIP_TRY(this, hipMemsetAsync(err.p, 0, 4 * sizeof(int32_t), stream));
    const int bundle_G = exact_ties != 2 ? choose_bundle_size(step) : 0;      // (exact_ties = 1: the bundle kernel's tie-detector variant)
    stats[DSA_STAT_BUNDLE_SIZE] = bundle_G;
    bundle_G_now = bundle_G;
    if (bundle_G == 0 && !grow_unit_pool()) return status;      // (plan() shrank the unit pool for bundles this call will not use)
    HIP_TRY(this, hipEventRecord(events[0], stream));
    std::vector<int32_t> h_info, h_flags;
    for (int first = 0; first < nunits; first += step) {
        const int n = std::min(step, nunits - first);
        bool redo_chunk = false, bundle_off_chunk = false;
      do {
        redo_chunk = false;
        const BatchPtrs b = batch();
        HIP_TRY(this, hipMemcpyAsync(src.p, h_src.data() + first, (size_t)n * sizeof(SourceDesc), hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipMemcpyAsync(risti_r.p, h_risti_r.data() + (size_t)first * kRefMax, (size_t)n * kRefMax * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipEventRecord(events[1], stream));
        HIP_TRY(this, hipMemsetAsync(pool_gen.p, 0, (size_t)pool_slots * sizeof(int), stream));       // (the coarse solve itself resets its field slot: FimEnds)
        {   // Launch order of the coarse solves: a launch ends with its slowest workgroups, and the rounds of a solve grow with the
            // distance from the source to the farthest corner of the grid, so the units with the longest fronts get the lowest
            // workgroup numbers (dispatched first) and the short ones fill the tail.
            h_launch_rank.resize((size_t)n);
            std::vector<std::pair<float, int>> far((size_t)n);
            for (int u = 0; u < n; ++u) {
                const SourceDesc& s = h_src[first + u];
                const float fx = (s.scx - g.gox) / g.dnx, fz = (s.scz - g.goz) / g.dnz;
                const float dx = std::max(fx, (float)(g.nnx - 1) - fx), dz = std::max(fz, (float)(g.nnz - 1) - fz);
                far[u] = { -(dx * dx + dz * dz), u };
            }
            std::stable_sort(far.begin(), far.end());
            for (int r = 0; r < n; ++r) h_launch_rank[(size_t)far[r].second] = r;
            HIP_TRY(this, hipMemcpyAsync(launch_rank.p, h_launch_rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
        }
        // the tie detector runs in the fixed-point modes (round 5: in exact_ties = 0 too, option tie_detect, so that a call that leaves tie-prone
        // units to the fixed point can say so: DSA_STAT_TIE_UNITS / DSA_STAT_TIE_UNITS_LEFT, dsa_unit_ties)
        const bool detect = exact_ties == 1 || (exact_ties == 0 && tie_detect);
        // bundles: the units of one source side by side in one workgroup (bundle_kernel.hip).  Solo units keep the launch ranks
        // 0 .. nsolo-1 (and the field slots those select), the members of the bundles follow, bundle by bundle.
        int nsolo = n, nbundles = 0;
        if (bundle_G > 0 && !bundle_off_chunk) {
            if (plan_bundles(first, n, bundle_G, &nsolo, &nbundles) != 0) return status;
        }
        const bool refined_b = nbundles > 0 && refined_bundles_now && !bundle_off_chunk;
        // (what the bundles hold, recorded now: a memory-bound march below gives the bundle buffers back before the statistic is written)
        const double bundle_mb = nbundles ? (double)(B_pool.cap * 4 + exc_b.cap * 8 + lists_b.cap * 4 + slowI.cap * 4 + cand_b.cap * 4 + Br_pool.cap * 4 + exc_br.cap * 8 + lists_br.cap * 4 + cand_br.cap * 4 + slowIr.cap * 4) / 1.0e6 : 0.0;
        launch_make_problems(g, b, n, slow.p, nrec_c, risti_c.p, window_r, window_c, prob_r.p, prob_c.p, info.p, clocks.p, launch_rank.p,
                             detect ? tieinfo.p : nullptr, tie_threshold, ends_c.p, fused_times ? rays.p : nullptr, veln.p, nfield, dpl, out.p, err.p,
                             nbundles ? member_flag.p : nullptr, bundle_window() * cell_c, bundle_max_rounds, stream, bundle_window_tail() * cell_c,
                             refined_b ? ends_r.p : nullptr);
        FimLaunch sr = shape_r, sc = shape_c;
        sr.tie = sc.tie = detect ? 1 : 0;
        launch_refine(g, b, n, velv.p, (size_t)g.nx * g.ny, rbasis.p, stream);
        if (exact_ties != 2) launch_refined_startup(g, b, n, stream);
        HIP_TRY(this, hipEventRecord(events[2], stream));
        if (exact_ties != 2 && refined_b) {
            // solo units (launch ranks 0 .. nsolo-1) one by one; the bundles' members side by side: member-minor slowness, the bundle kernel on the
            // boxes (256 threads), the converged members back into their records
            launch_fim(prob_r.p, nsolo, sr, stream);
            const int cnt[2] = { bundles_a, nbundles - bundles_a }, GG[2] = { bundle_G, bundle_Gb }, mpl[2] = { bundle_mpl_now, bundles_b > 0 && bundle_threads_b == 256 ? bundle_mpl_b : bundle_mpl_of(bundle_Gb ? bundle_Gb : bundle_G, nbundles) };
            for (int q = 0; q < 2; ++q) {
                if (cnt[q] <= 0) continue;
                const FimBundle* const bq = bundles_r_d.p + (q ? bundles_a : 0);
                launch_bundle_refined_slowness(bq, cnt[q], GG[q], prob_r.p, kRefRecs, slowIr.p + (q ? slowIr_off_b : 0), stream);
                launch_fim_bundles(bq, cnt[q], GG[q], 256, prob_r.p, ends_r.p, sr.tile_words, stream, GG[q] == 16 ? 4 : mpl[q], detect);
                launch_bundle_export_records(bq, cnt[q], GG[q], prob_r.p, kRefRecs, stream);
            }
        } else if (exact_ties != 2) launch_fim(prob_r.p, n, sr, stream);
        HIP_TRY(this, hipEventRecord(events[3], stream));
        if (exact_ties != 2) {
            const bool replay = detect && handoff_replay;
            if (replay) HIP_TRY(this, hipMemsetAsync(replay_list.p, 0, sizeof(int32_t), stream));
            launch_handoff(g, b, n, stream, detect ? tieinfo.p : nullptr, tie_threshold, replay ? replay_list.p : nullptr, kHandoffReplayCap, replay_scratch.p, xinfo.p);
            launch_coarse_march(g, b, n, slow.p, nrec_c, risti_c.p, stream, detect ? tieinfo.p : nullptr);
        }
        HIP_TRY(this, hipEventRecord(events[4], stream));
        if (exact_ties != 2) {
            launch_fim(prob_c.p, nsolo, sc, stream, ends_c.p);
            if (nbundles && bundles_b > 0) {
                // the bundles cut in halves on a second stream beside the whole ones: both wait for the stages before, the stream after waits for both
                if (!stream2) HIP_TRY(this, make_stream2());
                HIP_TRY(this, hipEventRecord(ev_b0, stream));
                launch_fim_bundles(bundles_d.p, bundles_a, bundle_G, bundle_threads(), prob_c.p, ends_c.p, sc.tile_words, stream, bundle_mpl_now, detect);
                HIP_TRY(this, hipStreamWaitEvent(stream2, ev_b0, 0));
                launch_fim_bundles(bundles_d.p + bundles_a, bundles_b, bundle_Gb, bundle_threads_b, prob_c.p, ends_c.p, sc.tile_words, stream2, bundle_mpl_b, detect);
                HIP_TRY(this, hipEventRecord(ev_b1, stream2));
                HIP_TRY(this, hipStreamWaitEvent(stream, ev_b1, 0));
            } else if (nbundles) launch_fim_bundles(bundles_d.p, nbundles, bundle_G, bundle_threads(), prob_c.p, ends_c.p, sc.tile_words, stream, bundle_mpl_now, detect);
        }
        HIP_TRY(this, hipEventRecord(events[5], stream));
        if (exact_ties) {
            // which units go through the literal march: all (2), or those whose fixed point met a tie / froze a cycle (1)
            std::vector<int> xl;
            if (exact_ties == 2) { xl.resize((size_t)n); for (int u = 0; u < n; ++u) xl[(size_t)u] = u; }
            else {
                std::vector<int32_t> h_tie((size_t)n * kTieWords), h_inf((size_t)n * 16);
                HIP_TRY(this, hipMemcpyAsync(h_tie.data(), tieinfo.p, (size_t)n * kTieWords * 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(this, hipMemcpyAsync(h_inf.data(), info.p, (size_t)n * 64, hipMemcpyDeviceToHost, stream));
                int32_t h_replayed = 0;
                if (handoff_replay) HIP_TRY(this, hipMemcpyAsync(&h_replayed, replay_list.p, 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(this, hipStreamSynchronize(stream));
                stats[DSA_STAT_HANDOFFS_REPLAYED] += (double)std::min<int32_t>(h_replayed, kHandoffReplayCap);
                //  a chunk that is about to be solved again -- exception table overflow, a refined or a coarse bundle that gave up: the
                // checks behind the receivers below -- is not marched now: its flagged units would march twice and keep flags of the abandoned attempt
                bool will_redo = false;
                for (int u = 0; u < n && !will_redo; ++u) {
                    const int32_t* fi = &h_inf[(size_t)u * 16];
                    const bool member = nbundles > 0 && h_member_flag[(size_t)u] != 0;
                    will_redo = fi[10] == -2 || (refined_b && member && fi[2] < 0) || (member && fi[10] == -1);
                }
                if (!will_redo) {
                    const std::vector<char> fl = tie_verdicts(first, n, h_tie.data(), h_inf.data(), nbundles > 0);
                    for (int u = 0; u < n; ++u) if (fl[(size_t)u]) { h_unit_flags[(size_t)(first + u)] |= 1; xl.push_back(u); }
                }
                stats[DSA_STAT_TIE_UNITS] += (double)xl.size();
            }
            if (!xl.empty()) {
                const auto w0 = std::chrono::steady_clock::now();
                // exact_ties = 1: the flagged units' receiver times once more, from the marched fields; their compact fields into the units'
                // slots when every unit of the launch has one (rays, keep_fields, small calls)
                // (times-only calls -- no rays, no keep_fields -- may march in pooled tiles: then no compact copy is made and the units' field slots hold nothing)
                const bool times_only = !rows && !keep_fields;
                bool compact = exact_ties == 1 ? n <= pool_slots : (!exact_fused || n <= pool_slots);
                if (times_only && exact_tiles_opt == 1) compact = false;
                marched_in_tiles = false;
                if (run_exact(first, n, xl, exact_fused || exact_ties == 1, compact, times_only) != 0) return status;
                stats[DSA_STAT_MS_EXACT] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
            }
        }
        // receivers of this chunk
        const int r0 = h_src[first].first_ray;
        const int r1 = h_src[first + n - 1].first_ray + h_src[first + n - 1].nrec;
        if (!fused_times && !exact_fused && r1 > r0) launch_srtimes_chunk(r0, r1 - r0, first);
        HIP_TRY(this, hipEventRecord(events[6], stream));
        HIP_TRY(this, hipGetLastError());
        h_info.resize((size_t)n * 16);
        h_flags.resize((size_t)n * 4);
        std::vector<int32_t> h_tie0;
        if (exact_ties == 0 && detect) { h_tie0.resize((size_t)n * kTieWords); HIP_TRY(this, hipMemcpyAsync(h_tie0.data(), tieinfo.p, (size_t)n * kTieWords * 4, hipMemcpyDeviceToHost, stream)); }
        HIP_TRY(this, hipMemcpyAsync(h_info.data(), info.p, (size_t)n * 16 * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(this, hipMemcpyAsync(h_flags.data(), flags.p, (size_t)n * 4 * 4, hipMemcpyDeviceToHost, stream));
        if (dsurf && r1 > r0) {
            const int d0 = h_rays[r0].data, d1 = h_rays[r1 - 1].data + 1;
            HIP_TRY(this, hipMemcpyAsync(dsurf + d0, out.p + d0, (size_t)(d1 - d0) * 4, hipMemcpyDeviceToHost, stream));
        }
        std::vector<unsigned long long> h_clk((size_t)n * kClockSlots);
        HIP_TRY(this, hipMemcpyAsync(h_clk.data(), clocks.p, (size_t)n * kClockSlots * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(this, hipStreamSynchronize(stream));
        for (int u = 0; u < n; ++u) {
            for (int q = 0; q < 6; ++q) phase_ticks[q] += (double)h_clk[(size_t)u * kClockSlots + q];
            phase_ticks[7] += (double)h_clk[(size_t)u * kClockSlots + 7];
            phase_ticks[6] = std::max(phase_ticks[6], (double)h_clk[(size_t)u * kClockSlots + 6]);
            for (int q = 8; q < kClockSlots; ++q) phase_ticks[q] += (double)h_clk[(size_t)u * kClockSlots + q];
        }
        float ms = 0;
        float a = 0, c2 = 0, d = 0;
        HIP_TRY(this, hipEventElapsedTime(&ms, events[2], events[3])); stats[DSA_STAT_MS_FIM_REFINED] += ms;
        HIP_TRY(this, hipEventElapsedTime(&ms, events[4], events[5])); stats[DSA_STAT_MS_FIM_COARSE] += ms;
        HIP_TRY(this, hipEventElapsedTime(&a, events[1], events[2]));
        HIP_TRY(this, hipEventElapsedTime(&c2, events[3], events[4]));
        HIP_TRY(this, hipEventElapsedTime(&d, events[5], events[6]));
        stats[DSA_STAT_MS_STAGES] += a + c2 + d;
        stats[DSA_STAT_LAUNCHES_FIM_COARSE] += 1;
        for (int u = 0; u < n; ++u) {      // the exception table first: a solve that ran out of table space may also have run out of rounds
            const int32_t* fi = &h_info[(size_t)u * 16];
            if (fi[10] != -2) continue;
            // more nodes off the causal order than the table holds: four times the table and the chunk once more
            const size_t grown_bytes = (size_t)pool_slots * (((size_t)8 << (exc_log2cap + 2)) - ((size_t)8 << exc_log2cap));
            size_t free_b = 0, total_b = 0;
            HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
            if (exc_log2cap + 2 > 26 || grown_bytes + ((size_t)pool_slots << exc_log2cap) * 8 > free_b || ensure(exc_c, (size_t)pool_slots << (exc_log2cap + 2))) {
                fail(DSA_ERR_INTERNAL, "unit %d: the exception table of the compact field overflowed (%d entries) and cannot grow", first + u, 1 << exc_log2cap);
                return DSA_ERR_INTERNAL;
            }
            exc_log2cap += 2;
            exc_log2cap_grown = exc_log2cap;
            per_slot_bytes += grown_bytes / (size_t)pool_slots;
            stats[DSA_STAT_RESCANS] += 1;
            redo_chunk = true;
            break;
        }
        if (redo_chunk) continue;
        if (refined_b) {    // a refined bundle that ran out of rounds or table space: the chunk once more with the refined boxes unit by unit
            bool bad = false;
            for (int u = 0; u < n && !bad; ++u) bad = h_member_flag[(size_t)u] && h_info[(size_t)u * 16 + 2] < 0;
            if (bad) { refined_bundles_failed = true; refined_bundles_now = false; redo_chunk = true; stats[DSA_STAT_RESCANS] += 1; continue; }
        }
        if (nbundles) {     // a bundle that did not converge under the shared schedule: its chunk once more, every unit by itself
            bool bad = false;
            for (int u = 0; u < n && !bad; ++u) bad = h_member_flag[(size_t)u] && h_info[(size_t)u * 16 + 10] == -1;
            if (bad) {
                if (getenv("DSA_DEBUG_BUNDLE"))
                    for (int u = 0; u < n; ++u) if (h_member_flag[(size_t)u] && h_info[(size_t)u * 16 + 10] == -1) { fprintf(stderr, "bundle member %d (unit %d) gave up after %d rounds (freezes %d)\n", u, first + u, h_info[(size_t)u * 16 + 8], h_info[(size_t)u * 16 + 11]); break; }
                bundle_off_chunk = true; redo_chunk = true; stats[DSA_STAT_RESCANS] += 1;
                if (!grow_unit_pool()) return status;
                bundles_failed = true;          // these maps do not bundle (fronts of the periods too different): unit by unit until the maps change
                continue;
            }
            stats[DSA_STAT_BUNDLES] += nbundles; stats[DSA_STAT_BUNDLED_UNITS] += n - nsolo;
            stats[DSA_STAT_FOOTPRINT_MB] = std::max(stats[DSA_STAT_FOOTPRINT_MB], ((double)pool_slots * (double)per_slot_bytes + (double)chunk * (double)per_unit_bytes) / 1.0e6 + bundle_mb);
        }
        for (int u = 0; u < n; ++u) {
            const int32_t* fi = &h_info[(size_t)u * 16];
            if (fi[10] == -3) { fail(DSA_ERR_INTERNAL, "unit %d: no free field slot within the wait bound (pool of %d slots, claimed by compare-and-swap: a pool smaller than the resident workgroups on very long solves, or corrupted busy flags; set option field_pool = -1 for a slot per unit)", first + u, pool_slots); return DSA_ERR_INTERNAL; }
            if (fi[2] < 0 || fi[10] == -1) { fail(DSA_ERR_INTERNAL, "unit %d: fixed-point solve did not converge (rounds %d/%d)", first + u, fi[0], fi[8]); return DSA_ERR_INTERNAL; }
            if (h_flags[(size_t)u * 4 + 1]) { fail(DSA_ERR_INTERNAL, "unit %d: serial march guard %d (1/17 window, 2/18 tree, 32 exception table)", first + u, h_flags[(size_t)u * 4 + 1]); return DSA_ERR_INTERNAL; }
            // (a coarse solve that never ran a round is NOT an error: for some sources in the last cell before a high edge the reference's own start-up
            // march ends with nothing alive and its field stays zero -- tests/tools/fuzz_parity.py "degenerate in the reference" --, and the engine
            // returns the same zeros.  The one case of round 6 that looked the same and was a bug -- refined boxes in bundles overwriting the records of
            // a unit whose start-up march had ended the refined stage -- is covered by tests/test_gpu_bundles.py.)
            stats[DSA_STAT_ROUNDS_MAX] = std::max(stats[DSA_STAT_ROUNDS_MAX], (double)fi[8]);
            h_unit_rounds[(size_t)(first + u)] = fi[8];
            unsigned long long ev;
            std::memcpy(&ev, fi + 12, 8);
            stats[DSA_STAT_EVALS_TOTAL] += (double)ev;
            std::memcpy(&ev, fi + 14, 8);
            stats[DSA_STAT_CHANGES_TOTAL] += (double)ev;
            stats[DSA_STAT_RESCANS] += fi[1] + fi[9];
            stats[DSA_STAT_FREEZES] += fi[3] + fi[11];
        }
        if (!h_tie0.empty()) {      // exact_ties = 0: what the detector saw is reported, nothing is solved again
            const std::vector<char> fl = tie_verdicts(first, n, h_tie0.data(), h_info.data(), nbundles > 0);
            for (int u = 0; u < n; ++u) if (fl[(size_t)u]) { h_unit_flags[(size_t)(first + u)] |= 1; stats[DSA_STAT_TIE_UNITS] += 1.0; stats[DSA_STAT_TIE_UNITS_LEFT] += 1.0; }
        }
        for (int u = 0; u < n; ++u) stats[DSA_STAT_TIE_INFLUENCE_MAX] = std::max(stats[DSA_STAT_TIE_INFLUENCE_MAX], (double)h_unit_tie[(size_t)(first + u)]);
        last_chunk_first = first;                               // the per-unit arrays (refined snapshots, ...) of this chunk stay resident ...
        last_chunk_n = n;
        fields_resident = n <= pool_slots && !marched_in_tiles;  // ... the coarse fields only when every unit had a slot (recycled slots: gone; marched in pooled tiles: never there)
        if (rows && trace_chunk(first, n, rw, iw, col, cap, nar) != 0) return status;
      } while (redo_chunk);
    }
    HIP_TRY(this, hipEventRecord(events[7], stream));
    HIP_TRY(this, hipEventSynchronize(events[7]));
    float ms = 0;
    HIP_TRY(this, hipEventElapsedTime(&ms, events[0], events[7]));
    stats[DSA_STAT_MS_TOTAL] = ms;
    int32_t herr[4];
    HIP_TRY(this, hipMemcpy(herr, err.p, sizeof herr, hipMemcpyDeviceToHost));
    if (herr[0]) { fail(DSA_ERR_OUTSIDE, "Receiver lies outside model (ray %d)", herr[0] - 1); return DSA_ERR_OUTSIDE; }
    return 0;
}

// What the bundle field slots may take: the free memory, or -- a memory budget is set (dsa_set_memory_budget: several ranks on one device) --
// what the budget leaves beside the unit slots and the per-unit arrays, so that two ranks planning at once cannot both claim the device
// (ADVICE r03).  (The callers take 70 % of it.)
size_t Engine::bundle_room(size_t free_b) const
{
    if (mem_budget == 0) return free_b;
    const size_t held = (size_t)pool_slots * per_slot_bytes + (size_t)chunk * per_unit_bytes;
    const size_t room = plan_budget > held ? plan_budget - held : 0;
    return std::min(free_b, (size_t)((double)room / 0.7));
}

// Members per bundle for this call: the option, or (automatic) the largest of 16 / 8 / 4 that still gives the chip enough workgroups
// and whose field slots fit the memory; 0 = no bundles.  `step` = units per launch.
int Engine::choose_bundle_size(int step, long* solo_units)
{
    if (solo_units) *solo_units = (long)h_src.size();
    bundle_wide = false;
    if (bundle_opt == 0 || h_src.empty() || (bundles_failed && bundle_opt == 1)) return 0;
    // units per source (same coordinates bit for bit), in planned order
    std::map<std::pair<uint32_t, uint32_t>, int> count;
    for (const SourceDesc& sd : h_src) { uint32_t a, b2; std::memcpy(&a, &sd.scx, 4); std::memcpy(&b2, &sd.scz, 4); ++count[{ a, b2 }]; }
    auto bundles_with = [&](int G) { long nb = 0; for (auto& kv : count) { nb += kv.second / G; if (kv.second % G >= 2) ++nb; } return nb; };
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    free_b += B_pool.cap * 4 + exc_b.cap * 8 + lists_b.cap * 4;          // (what a previous call holds is reused)
    free_b = bundle_room(free_b);
    auto slot_bytes = [&](int G) { int lg = G == 16 ? 4 : G == 8 ? 3 : 2; return (size_t)(G * DSA_BSTRIDE + 1) * nrec_c * 4 + ((size_t)8 << (exc_log2cap + lg)) + lists_c_stride * 4; };
    auto fits = [&](int G) {
        if ((unsigned long long)nrec_c * (unsigned long long)G * 4ull * DSA_BSTRIDE >= (1ull << 32)) return false;          // 32-bit byte offsets inside a bundle field
        if (nrec_c >= ((size_t)1 << 27)) return false;                                                          // (record indices share the ready-list word with the far-load code, and -- shifted by four -- the tie candidates' word with a flag bit)
        if ((unsigned long long)nrec_c * (unsigned long long)G >= (1ull << 30)) return false;                // exception keys
        if ((unsigned long long)nrec_c * (unsigned long long)nmaps * 4ull >= (1ull << 32)) return false;      // ... and inside the member-minor slowness
        const long nb = std::min<long>(bundles_with(G), (long)step);
        const size_t want = (size_t)std::min<long>(std::max<long>(nb, 1), (long)(bundles_resident(G, bundle_mpl_of(G, nb)) * 9 / 8));      // (bundles resident at a time, and a few more)
        return want * slot_bytes(G) + (size_t)nmaps * nrec_c * 4 < (size_t)(0.7 * (double)free_b);      // (what plan_bundles allows itself)
    };
    if (bundle_opt == 4 || bundle_opt == 8 || bundle_opt == 16) {
        if (!(fits(bundle_opt) && bundles_with(bundle_opt) > 0)) return 0;
        if (solo_units) { long cov = 0; for (auto& kv : count) { cov += (kv.second / bundle_opt) * bundle_opt; if (kv.second % bundle_opt >= 2) cov += kv.second % bundle_opt; } *solo_units = (long)h_src.size() - cov; }
        return bundle_opt;
    }
    // automatic: a bundle is one workgroup where its members would have been G, so it pays only while the bundles still fill the chip.
    // The best of the estimates below wins: a launch-time model from times measured at 1025^2 (only the ratios decide; they hold from 129^2
    // to 4097^2: profiles/r03_bundle_sizes.log, r04_bundle_occupancy.log).  Beyond 1500 nodes per side the bundle kernel runs wide (768
    // threads, one workgroup per CU: a third as many bundles fill the chip); round 3's table of rates serves there.
    // Grid size: round 3 kept grids below 400 nodes per side unit by unit.  With round 4's kernel the bundles win there too when the call
    // has the sources (16 periods x 1000 sources: 385^2 66 k -> 141 k solves/s, 257^2 119 k -> 200 k, 129^2 232 k -> 288 k; 200 sources at
    // 257^2: 101 k -> 116 k; profiles/r04_bundle_occupancy.log), so the floor is 120 nodes per side.
    if (std::min(g.nnx, g.nnz) < 120) return 0;
    // ... but only for launches that fill the chip: below 400 nodes per side a solve is a few hundred short rounds, many unit-by-unit
    // workgroups share a CU, and a handful of bundles -- the Taipei example: 34 bundles for 449 units at 137^2 -- take longer than the
    // units by themselves (coarse solves 2.9 -> 7.0 ms); there a size needs 384 bundles of at least 8 members and the wide variant stays off
    const bool small_grid = std::min(g.nnx, g.nnz) < 400;
    const double n_units = (double)std::min<size_t>(h_src.size(), (size_t)step);
    const double solo_rate = 10.0 * std::min(1.0, n_units / 1100.0);          // k solves/s
    double best = solo_rate * 1.05;
    int pick = 0;
    const int sizes[3] = { 16, 8, 4 };
    // Round 4 (256-thread kernel): a launch's time from the measured time of ONE bundle at one / two / three workgroups per CU at 1025^2 (ms;
    // the ratios hold at other sizes: only ratios decide) -- a bundle takes what its rounds take, whatever shares the chip with it, so a
    // launch of nb <= 768 bundles takes one bundle's time at that occupancy, a longer one whole generations plus a last partial one that
    // costs at least 45 % of a generation (profiles/r04_bundle_occupancy.log); between 768 and 1 500 bundles plan_bundles cuts the last
    // ones in halves (1 000 bundles of 16: 383 ms)
    const double t_one[3][3] = { { 189.5, 210.0, 241.0 }, { 130.8, 151.7, 187.2 }, { 104.8, 125.4, 164.3 } };
    const double rate512[3] = { 24.5, 20.4, 15.5 };                             // (large grids, one wide workgroup per CU: round 3's table; only the ratios decide)
    const double t_wide[3] = { 106.0, 77.0, 69.0 };                             // (768 threads at 1025^2, a CU per bundle: ms of one bundle of 16 / 8 / 4)
    bool pick_wide = false;
    for (int k = 0; k < 3; ++k) {
        const int G = sizes[k];
        long nb = 0, covered = 0;
        for (auto& kv : count) { nb += kv.second / G; covered += (kv.second / G) * G; if (kv.second % G >= 2) { ++nb; covered += kv.second % G; } }
        if (nb == 0 || !fits(G)) continue;
        nb = std::min<long>(nb, (long)step);
        if (small_grid && (nb < 384 || G < 8)) continue;      // (bundles of 4 lose there: 100 sources x 16 at 257^2 73 k against 93 k unit by unit)
        const double frac = (double)covered / (double)h_src.size();                  // units that end up in bundles ...
        const double fill = (double)covered / ((double)nb * G);                       // ... and how full the bundles are
        double est;
        if (bundle_threads() >= 512) est = frac * rate512[k] * fill * std::min(1.0, (double)nb / 280.0) + (1.0 - frac) * solo_rate;
        else {
            const double occ = (double)nb / 256.0;
            double ms;
            if (occ <= 1.0) ms = t_one[k][0];
            else if (occ <= 2.0) ms = t_one[k][0] + (t_one[k][1] - t_one[k][0]) * (occ - 1.0);
            else if (occ <= 3.0) ms = t_one[k][1] + (t_one[k][2] - t_one[k][1]) * (occ - 2.0);
            else {
                const double gens = std::floor((double)nb / 768.0), rem = (double)nb / 768.0 - gens;
                ms = t_one[k][2] * (gens + (rem > 0.0 ? 0.45 + 0.55 * rem : 0.0));
                if (G >= 8 && nb > 768 && nb < 1500) ms *= 0.98;                    // (the halved last bundles)
                // (round 5) at most 256 bundles beyond the first generation: whole, 768 threads wide, a CU each as the first generation leaves
                // (1 000 bundles of 16: 325 ms against 368 with halves, profiles/r05_ab_bundle_kernel.log)
                if (bundle_tail_opt == 1 && G >= 8 && bundle_mpl == 0 && nb > 768 && nb <= 1024) ms = t_one[k][2] + 0.8 * (t_wide[k] + 12.0 * (double)(nb - 768) / 256.0);
            }
            // k solves/s (= units per ms) of the whole launch: the bundles in `ms` (a little less when they are not full: idle member lanes save no trips), the rest unit by unit behind them
            const double units_s = n_units * (1.0 - frac);
            est = n_units / (ms * (0.65 + 0.35 * fill) + units_s / std::max(solo_rate, 1e-9));
        }
        bool wide = false;
        if (bundle_threads_opt == 0 && bundle_threads() == 256 && nb <= 256 && !small_grid) {
            // ... or a CU per bundle, 768 threads wide (one bundle's time at that width, nearly flat in the number of bundles)
            const double ms_w = t_wide[k] + 12.0 * (double)nb / 256.0;
            const double est_w = n_units / (ms_w * (0.65 + 0.35 * fill) + n_units * (1.0 - frac) / std::max(solo_rate, 1e-9));
            if (est_w > est) { est = est_w; wide = true; }
        }
        if (est > best) { best = est; pick = G; pick_wide = wide; if (solo_units) *solo_units = (long)h_src.size() - covered; }
    }
    bundle_wide = pick_wide;
    return pick;
}

// Bundles of the resident chunk [first, first + n): launch ranks (solo units first, longest fronts first in both groups), the member
// flags for k_make_problems, the bundle descriptors and their field slots.
int Engine::plan_bundles(int first, int n, int G, int* nsolo_out, int* nbundles_out)
{
    std::map<std::pair<uint32_t, uint32_t>, std::vector<int>> groups;
    for (int u = 0; u < n; ++u) {
        const SourceDesc& sd = h_src[(size_t)(first + u)];
        uint32_t a, b2; std::memcpy(&a, &sd.scx, 4); std::memcpy(&b2, &sd.scz, 4);
        groups[{ a, b2 }].push_back(u);
    }
    auto farness = [&](int u) {
        const SourceDesc& sd = h_src[(size_t)(first + u)];
        const float fx = (sd.scx - g.gox) / g.dnx, fz = (sd.scz - g.goz) / g.dnz;
        const float dx = std::max(fx, (float)(g.nnx - 1) - fx), dz = std::max(fz, (float)(g.nnz - 1) - fz);
        return -(dx * dx + dz * dz);
    };
    h_member_flag.assign((size_t)n, 0);
    std::vector<std::pair<float, std::vector<int>>> pieces;
    for (auto& kv : groups) {
        const std::vector<int>& v = kv.second;
        for (size_t k = 0; k < v.size(); k += (size_t)G) {
            const size_t m = std::min<size_t>((size_t)G, v.size() - k);
            if (m < 2) continue;
            pieces.push_back({ farness(v[k]), std::vector<int>(v.begin() + (long)k, v.begin() + (long)(k + m)) });
            for (size_t q = k; q < k + m; ++q) h_member_flag[(size_t)v[q]] = 1;
        }
    }
    std::stable_sort(pieces.begin(), pieces.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
    std::vector<std::pair<float, int>> solo;
    for (int u = 0; u < n; ++u) if (!h_member_flag[(size_t)u]) solo.push_back({ farness(u), u });
    std::stable_sort(solo.begin(), solo.end());
    const int nsolo = (int)solo.size();
    int nb = (int)pieces.size();
    *nsolo_out = nsolo; *nbundles_out = nb;
    bundles_a = nb; bundles_b = 0; bundle_Gb = 0;
    if (nb == 0) return 0;
    // Members per lane and workgroups per CU (round 4, profiles/r04_bundle_occupancy.log; bundle_mpl_of, bundle_kernel.hip: DSA_BUNDLE_OCC):
    // bundles of 16 run three workgroups per CU (768 resident), bundles of 8 / 4 too when the launch has more than 512 of them.  A bundle
    // takes longer with two neighbours on its CU than with one (768 bundles of 16: 241 ms; 500: 210; 250: 190), but a CU finishes more of
    // them per second.  The catch is a last generation that is nearly empty -- 1 000 bundles = 768 + 232 -- so, automatic mode, between 768
    // and 1 500 bundles: the first 768 (the longest) as they are and the REST CUT IN HALVES -- bundles of G / 2, two members per lane, on a
    // second stream, which fill the CUs the first launch frees one by one (1 000 sources x 16 periods: 383 ms against ~430 uncut).
    const bool auto_mpl = bundle_mpl == 0 && bundle_threads() == 256;
    const int res3 = 768;
    bundle_mpl_now = bundle_mpl_of(G, nb);
    bundle_mpl_b = bundle_mpl ? bundle_mpl : 2;
    bundle_threads_b = bundle_threads();
    std::vector<std::pair<float, std::vector<int>>> tail;
    if (auto_mpl && bundle_opt == 1 && G >= 8 && nb > res3 && nb < 1500 && bundle_tail_opt == 1 && nb - res3 <= 256) {
        // Round 5, option bundle_tail = 1: the bundles beyond the first generation stay WHOLE and run 768 threads wide on the second stream, a CU
        // each as the first generation's workgroups leave (a bundle of 16 alone on a CU: 106 ms wide against 190 with 256 threads)
        tail.assign(pieces.begin() + res3, pieces.end());
        pieces.resize((size_t)res3);
        for (const auto& pc : tail) for (int u : pc.second) h_member_flag[(size_t)u] = 2;      // (their own causal window: bundle_window_tail)
        bundles_a = res3; bundles_b = (int)tail.size(); bundle_Gb = G;
        bundle_mpl_b = 4; bundle_threads_b = 768;
    } else if (auto_mpl && bundle_opt == 1 && G >= 8 && nb > res3 && nb < 1500) {
        std::vector<std::pair<float, std::vector<int>>> keep(pieces.begin(), pieces.begin() + res3);
        for (size_t k = (size_t)res3; k < pieces.size(); ++k) {
            const std::vector<int>& v = pieces[k].second;
            if ((int)v.size() >= G / 2 + 2) {
                tail.push_back({ pieces[k].first, std::vector<int>(v.begin(), v.begin() + G / 2) });
                tail.push_back({ pieces[k].first, std::vector<int>(v.begin() + G / 2, v.end()) });
            } else keep.push_back(pieces[k]);
        }
        pieces.swap(keep);
        bundles_a = (int)pieces.size(); bundles_b = (int)tail.size(); bundle_Gb = G / 2;
        nb = bundles_a + bundles_b;
        *nbundles_out = nb;
    }
    if (bundle_order_opt && bundles_b > 0 && bundle_threads_b == 768 && (int)pieces.size() == res3) {
        // Round 6, option bundle_order (A/B switch, default 0): which bundles share a CU.  If the first generation's 768 workgroups went out round robin --
        // k, k + 256 and k + 512 on one CU -- "longest first" would give every CU a long, a middle and a short bundle, and the wide tail, whose workgroups
        // need whole CUs, could only start when the generation is all but over; position p holding rank 3 (p mod 256) + p / 256 would then put bundles of
        // similar length on one CU.  Measured (profiles/r06_ab_bundle_order.log): 350.4 ms against 342.0 -- the dispatcher fills a CU with CONSECUTIVE
        // workgroups, "longest first" already is the grouped order (the rocprof trace shows the tail starting with the generation and ending 84 ms after it),
        // and the permutation un-groups it.
        std::vector<std::pair<float, std::vector<int>>> perm((size_t)res3);
        const int ncu = res3 / 3;
        for (int p2 = 0; p2 < res3; ++p2) {
            const int rank2 = bundle_order_opt == 1 ? 3 * (p2 % ncu) + p2 / ncu : p2;
            perm[(size_t)p2] = pieces[(size_t)rank2];
        }
        if (bundle_order_opt == 1) pieces.swap(perm);
    }
    h_launch_rank.assign((size_t)n, 0);
    for (int r = 0; r < nsolo; ++r) h_launch_rank[(size_t)solo[(size_t)r].second] = r;
    size_t free_b = 0, total_b = 0;
    HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
    free_b += B_pool.cap * 4 + exc_b.cap * 8 + lists_b.cap * 4;
    free_b = bundle_room(free_b);
    // field slots per group of bundles: one per bundle, or -- more bundles than the chip holds at a time -- as many as can be resident and a
    // few more; a bundle claims a free one when it starts (FimBundle::slot_busy)
    struct Group { int G, count, slots, xlog; size_t b_stride, b_off, exc_off, slot0; };
    Group gr[2] = { { G, bundles_a, 0, 0, 0, 0, 0, 0 }, { bundle_Gb, bundles_b, 0, 0, 0, 0, 0, 0 } };
    size_t b_total = 0, exc_total = 0, slots_total = 0;
    for (int q = 0; q < 2; ++q) {
        Group& r = gr[q];
        if (r.count == 0) continue;
        const int lg = r.G == 16 ? 4 : r.G == 8 ? 3 : 2;
        r.xlog = exc_log2cap + lg;
        r.b_stride = (size_t)(r.G * DSA_BSTRIDE + 1) * nrec_c;
        const size_t slot_b = r.b_stride * 4 + ((size_t)8 << r.xlog) + lists_c_stride * 4;
        const size_t room = (size_t)(0.7 * (double)free_b) / slot_b / (bundles_b ? 2 : 1);
        if (room < 1) { fail(DSA_ERR_DEVICE, "bundles: no room for one bundle field slot (%zu B)", slot_b); return DSA_ERR_DEVICE; }
        const size_t resident = bundles_resident(r.G, q == 0 ? bundle_mpl_now : bundle_mpl_b, q == 0 ? bundle_threads() : bundle_threads_b);
        r.slots = (int)std::min<size_t>({ (size_t)r.count, (size_t)(bundle_pool_opt > 0 ? bundle_pool_opt : (int)(resident + resident / 8)), room });
        r.b_off = b_total; r.exc_off = exc_total; r.slot0 = slots_total;
        b_total += (size_t)r.slots * r.b_stride; exc_total += (size_t)r.slots << r.xlog; slots_total += (size_t)r.slots;
    }
    bundle_slots = gr[0].slots;
    // tie candidates per bundle slot (the census' list, bundle_kernel.hip): an eighth of the field's nodes, at least 65 535 entries
    const bool want_cand = exact_ties == 1 || (exact_ties == 0 && tie_detect);
    const size_t cand_cap = std::max<size_t>(65535, nrec_c / 8), cand_stride = ((cand_cap + 4) & ~(size_t)3) + kTieSeenSlots;
    if (want_cand && ensure(cand_b, slots_total * cand_stride)) return status;
    if (ensure(B_pool, b_total) || ensure(exc_b, exc_total) || ensure(lists_b, slots_total * lists_c_stride) || ensure(bpool_gen, slots_total) ||
        ensure(bundles_d, (size_t)nb) || ensure(member_flag, (size_t)n) || ensure(slowI, (size_t)nmaps * nrec_c)) return status;
    if (!slowI_ready) { launch_interleave_maps(slow.p, nrec_c, nmaps, slowI.p, stream); slowI_ready = true; }
    h_bundles.assign((size_t)nb, FimBundle{});
    int rank = nsolo;
    for (int k = 0; k < nb; ++k) {
        const Group& r = gr[k < bundles_a ? 0 : 1];
        const std::vector<int>& mem = k < bundles_a ? pieces[(size_t)k].second : tail[(size_t)(k - bundles_a)].second;
        const int kk = k < bundles_a ? k : k - bundles_a;
        FimBundle& bd = h_bundles[(size_t)k];
        bd.B = B_pool.p + r.b_off; bd.b_stride = r.b_stride; bd.p_offset = (size_t)r.G * DSA_BSTRIDE * nrec_c;
        bd.exc = exc_b.p + r.exc_off; bd.exc_stride = (size_t)1 << r.xlog; bd.exc_log2cap = r.xlog;
        bd.lists = lists_b.p + r.slot0 * lists_c_stride; bd.lists_stride = lists_c_stride;
        bd.slot_busy = r.slots < r.count ? bpool_gen.p + r.slot0 : nullptr; bd.nslots = r.slots; bd.slot = r.slots < r.count ? 0 : kk;
        bd.slowI = slowI.p; bd.np = nmaps; bd.far_all = bundle_far_all;
        bd.cand = want_cand ? cand_b.p + r.slot0 * cand_stride : nullptr; bd.cand_stride = cand_stride; bd.cand_cap = (int)cand_cap; bd.cand_list = tie_list_opt;
        bd.nmem = (int)mem.size();
        for (int m = 0; m < kBundleMax; ++m) { bd.member[m] = 0; bd.map[m] = 0; }
        for (int m = 0; m < bd.nmem; ++m) {
            const int u = mem[(size_t)m];
            h_launch_rank[(size_t)u] = rank;
            bd.member[m] = rank++;
            bd.map[m] = h_src[(size_t)(first + u)].period;
        }
    }
    // the same bundles once more for the refined boxes (engine.h: bundle_refined_opt): a slot per bundle, the group's member count per node
    // (from 128 bundles on: a dozen bundles leave the chip emptier than their 200 unit-by-unit workgroups did -- 12 bundles of 16 at 1025^2: refined stage
    // 2.1 -> 4.9 ms; 1 000 bundles: 29.1 -> 14.7 ms, profiles/r05_ab_refined_bundles.log; option bundle_refined = 2 forces it for tests)
    refined_bundles_now = bundle_refined_opt && (nb >= 128 || bundle_refined_opt == 2) && !refined_bundles_failed && exact_ties != 2;
    if (refined_bundles_now) {
        const size_t nrec_r = kRefRecs;
        const int xlog_r0 = exc_log2cap_of(kRefMax, kRefMax);
        const size_t lists_r_stride = ((size_t)kFimMaskInts * kRefTiles * kRefTiles + 2 + 1) & ~(size_t)1;
        const size_t cand_cap_r = 65535, cand_stride_r = ((cand_cap_r + 4) & ~(size_t)3) + kTieSeenSlots;
        size_t b_tot = 0, x_tot = 0, s_tot = 0;
        std::vector<size_t> b_off((size_t)nb), x_off((size_t)nb), s_off((size_t)nb);
        for (int k = 0; k < nb; ++k) {
            const int GG = k < bundles_a ? G : bundle_Gb, lg = GG == 16 ? 4 : GG == 8 ? 3 : 2;
            b_off[(size_t)k] = b_tot; x_off[(size_t)k] = x_tot; s_off[(size_t)k] = s_tot;
            b_tot += (size_t)(GG * DSA_BSTRIDE + 1) * nrec_r; x_tot += (size_t)1 << (xlog_r0 + lg); s_tot += (size_t)GG * nrec_r;
            if (k + 1 == bundles_a) slowIr_off_b = s_tot;
        }
        if (ensure(Br_pool, b_tot) || ensure(exc_br, x_tot) || ensure(lists_br, (size_t)nb * lists_r_stride) || ensure(slowIr, s_tot) || ensure(bundles_r_d, (size_t)nb) ||
            ensure(ends_r, (size_t)n) || (want_cand && ensure(cand_br, (size_t)nb * cand_stride_r))) return status;
        h_bundles_r = h_bundles;
        for (int k = 0; k < nb; ++k) {
            const int GG = k < bundles_a ? G : bundle_Gb, lg = GG == 16 ? 4 : GG == 8 ? 3 : 2;
            FimBundle& bd = h_bundles_r[(size_t)k];
            bd.B = Br_pool.p + b_off[(size_t)k]; bd.b_stride = 0; bd.p_offset = (size_t)GG * DSA_BSTRIDE * nrec_r;
            bd.exc = exc_br.p + x_off[(size_t)k]; bd.exc_stride = 0; bd.exc_log2cap = xlog_r0 + lg;
            bd.lists = lists_br.p + (size_t)k * lists_r_stride; bd.lists_stride = 0;
            bd.slot_busy = nullptr; bd.nslots = 1; bd.slot = 0;
            bd.slowI = slowIr.p + s_off[(size_t)k]; bd.np = GG;
            bd.cand = want_cand ? cand_br.p + (size_t)k * cand_stride_r : nullptr; bd.cand_stride = 0; bd.cand_cap = (int)cand_cap_r; bd.cand_list = tie_list_opt;
            for (int m = 0; m < kBundleMax; ++m) bd.map[m] = m < GG ? m : 0;
        }
        HIP_TRY(this, hipMemcpyAsync(bundles_r_d.p, h_bundles_r.data(), (size_t)nb * sizeof(FimBundle), hipMemcpyHostToDevice, stream));
    }
    HIP_TRY(this, hipMemsetAsync(bpool_gen.p, 0, slots_total * sizeof(int), stream));
    HIP_TRY(this, hipMemcpyAsync(bundles_d.p, h_bundles.data(), (size_t)nb * sizeof(FimBundle), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(member_flag.p, h_member_flag.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(launch_rank.p, h_launch_rank.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream));
    stats[DSA_STAT_BUNDLE_SLOTS] = (double)slots_total;
    stats[DSA_STAT_BUNDLE_THREADS] = (double)bundle_threads();
    return 0;
}

// literal Fast Marching (exact_kernel.hip) for the chunk-local units `xl` of the resident chunk [first, first + n), in batches of
// as many units as the pool holds
// `receivers`: the batches write their units' receiver times themselves; `compact`: the units' compact fields go to their slots (T_c)
int Engine::run_exact(int first, int n, const std::vector<int>& xl, bool receivers, bool compact, bool may_pool_tiles)
{
    (void)n;
    // (a tree holds at most 65 535 nodes -- sixteen levels, one per lane of a unit's group; narrow bands are a few times nnx + nnz)
    const int gcap_max = std::min(16 * (g.nnx + g.nnz) + 4096, 65534);
    // (the tree's global part: slot by slot, or -- whole levels in LDS -- in blocks of three levels, whose last generation is allocated whole)
    size_t heap_bytes = (size_t)gcap_max * 8;
    if (exact_heap_blocked) for (int l = 6; l <= 12; ++l) heap_bytes = std::max(heap_bytes, exact_heap_blocked_entries((1 << l) - 1, std::min(gcap_max, 65534 - ((1 << l) - 1)) & ~1) * 8);
    size_t per = nrec_c * 4 + heap_bytes + exact_start_bytes() + 4;      // (one packed word per node, exact_kernel.hip)
    // Pooled tiles (round 5, kernels.h XTiles): a times-only batch whose whole fields would not all fit marches in pooled tiles -- tcap tiles of
    // 8 x 8 nodes per unit (the band and what lies within a tile of it: a few times the tiles along the grid's perimeter) instead of a word per
    // node.  At 4097^2: 3.2 MB per unit instead of 67, so the batch is bounded by exact_pool_max, not by memory.  Option exact_tiles: 0 automatic,
    // 1 always (times-only calls), -1 never; exact_tile_cap: tiles per unit, 0 = 8 (nbx + nbz).
    const int ntile = g.nbx * g.nbz;
    const int tcap = exact_tile_cap > 0 ? exact_tile_cap : std::min(65000, std::max(512, 8 * (g.nbx + g.nbz)));
    const size_t per_tiles = exact_tile_unit_bytes(ntile, tcap) + heap_bytes + exact_start_bytes() + 4;
    bool tiles = false;
    if (may_pool_tiles && receivers && !compact && exact_tiles_opt >= 0 && tcap < ntile) {
        if (exact_tiles_opt == 1) tiles = true;
        else {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
            const size_t have = X_pool.cap * 4 + X_heap.cap * 8 + x_starts.cap * 8 + x_nstart.cap * 4 + B_pool.cap * 4 + exc_b.cap * 8 + lists_b.cap * 4;
            const size_t fit_full = (size_t)(0.80 * (double)(free_b + have)) / per;
            // (the tile table is one more dependent look-up per access and the march is bound by exactly that latency: per accept the pooled march
            // runs at 55-63 % of the march on whole fields -- 468 against 741 M accepts/s with 3 072 units at 4097^2 -- and wins by the units it
            // holds side by side: 857 M accepts/s with 9 600.  So: only when the batch would otherwise take three rounds or more.  profiles/r05_exact_pooled_tiles.log)
            tiles = 2 * fit_full < std::min<size_t>(xl.size(), exact_pool > 0 ? (size_t)exact_pool : exact_pool_max) && per_tiles * 4 < per;
        }
    }
    if (tiles) per = per_tiles;
    marched_in_tiles = tiles;
    released_bundles_for_march = false;
    size_t pool = (size_t)exact_pool;
    if (!pool) {
        // units marching at a time: as many as 80 % of the free memory holds, at most exact_pool_max (four units per wavefront; at 4097^2 the
        // rate is the units in flight: 1 536 units 28 solves/s, 2 688 43, 3 456 52 -- profiles/r04_exact_rates.log)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
        const size_t have = X_pool.cap * 4 + X_heap.cap * 8 + x_starts.cap * 8 + x_nstart.cap * 4 + X_tt.cap * 2 + X_tp.cap * 4 + X_ring.cap * 4 + X_free.cap * 2 + X_pins.cap * 4;
        pool = std::min<size_t>(exact_pool_max, std::max<size_t>(1, (size_t)(0.80 * (double)(free_b + have)) / per));
        if (pool < xl.size() && B_pool.cap > 0) {
            // exact_ties = 1 on a large grid: the bundles have converged and their field slots (164 GB at 4097^2) are what keeps the march
            // from holding its units side by side -- at 4097^2 the march's rate IS the number of units in flight.  They go back (the next
            // plan_bundles allocates them again) before the pool is sized.
            auto release = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
            HIP_TRY(this, hipStreamSynchronize(stream));
            if (stream2) HIP_TRY(this, hipStreamSynchronize(stream2));
            // (the candidate lists and the refined boxes' bundle buffers go with them -- GBs at 4097^2 --, and nothing may launch the
            // bundle descriptors that still point there: plan_bundles builds them again for the next chunk or call)
            release(B_pool); release(exc_b); release(lists_b); release(cand_b); release(Br_pool); release(exc_br); release(lists_br); release(cand_br); release(slowIr);
            bundles_a = bundles_b = 0; h_bundles.clear(); h_bundles_r.clear();
            released_bundles_for_march = true;
            HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
            pool = std::min<size_t>(exact_pool_max, std::max<size_t>(1, (size_t)(0.80 * (double)(free_b + have)) / per));
        }
    }
    pool = std::min(pool, xl.size());
    {   // batches of equal size (10 240 + 6 144 units take as long as two full batches)
        const size_t nb = (xl.size() + pool - 1) / pool;
        pool = (xl.size() + nb - 1) / nb;
    }
    // tree slots in LDS per marching unit (odd: the two children of a slot then lie on one side of the LDS / global split): the option, or
    // (0) what lets all the wavefronts of a batch be resident -- a wavefront holds four units' tree tops, 8 bytes per slot
    int lcap = exact_lds_slots;
    if (lcap <= 0) {
        // (every wavefront of a batch must be resident at once -- a second generation would double the batch's time --, so the LDS of a CU is
        // divided with a margin: 148 of its 160 KB; 16 scratch entries per unit on top of the lcap + 1 tree slots)
        const size_t waves_per_cu = std::max<size_t>(1, ((pool + 3) / 4 + 255) / 256);
        const size_t bytes = (size_t)148 * 1024 / waves_per_cu;
        const long fit = (long)(bytes / 32) - 17;
        lcap = (int)std::min<long>(std::min<long>(fit, 4L * (g.nnx + g.nnz) + 1023), 4799);
    }
    lcap = std::max(63, lcap) | 1;
    // (round 5) a batch that fills the chip -- twelve wavefronts or more per CU: the march is then bound by the fabric and by issue together,
    // profiles/r05_pmc_march_full_load.txt -- keeps whole levels in LDS and stores the tree's global part in blocks of three levels
    // (exact_kernel.hip: xg_gi): 16 000 units at 1025^2 1 956 -> 2 137 solves/s.  A smaller batch waits for its own instructions, and the
    // block arithmetic is more of them: 4 096 units 1 244 -> 1 120, 1 536 units at 4097^2 26.8 -> 23.8 (profiles/r05_ab_march_heap.log).
    // Option exact_heap_blocked: 0 never, 1 as above, 2 whenever the LDS part is whole levels (tests).
    int lb = 0;
    const bool full = ((pool + 3) / 4 + 255) / 256 >= 12;
    if (exact_heap_blocked == 1 && full && exact_lds_slots <= 0 && lcap < 4 * (g.nnx + g.nnz) + 1023) { int l = 6; while ((2 << l) - 1 <= lcap) ++l; lcap = (1 << l) - 1; }
    if (((exact_heap_blocked == 1 && full) || exact_heap_blocked == 2) && ((lcap + 1) & lcap) == 0) { while ((1 << lb) < lcap + 1) ++lb; }
    const int gcap = std::min(gcap_max, 65534 - lcap) & ~1;
    const size_t gstride = lb ? exact_heap_blocked_entries(lcap, gcap) : (size_t)gcap;
    if (exact_lds_bytes(lcap) > 156 * 1024) { fail(DSA_ERR_ARGUMENT, "exact_lds_slots %d needs more than 156 KB of LDS (four units per wavefront)", lcap); return DSA_ERR_ARGUMENT; }
    if (ensure(X_heap, pool * gstride) || ensure(x_starts, pool * (exact_start_bytes() / 8)) || ensure(x_nstart, pool)) return status;
    XTiles xt{};
    if (tiles) {
        if (ensure(X_tt, pool * exact_tile_table_entries(ntile)) || ensure(X_tp, pool * (size_t)tcap * 64) || ensure(X_ring, pool * (size_t)tcap) || ensure(X_free, pool * (size_t)tcap) ||
            ensure(X_pins, pool * (((size_t)ntile + 31) / 32)) || ensure(X_pool, 4)) return status;
        xt = XTiles{ X_tt.p, X_tp.p, X_ring.p, X_free.p, X_pins.p, tcap };
    } else if (ensure(X_pool, pool * nrec_c)) return status;
    stats[DSA_STAT_EXACT_POOL] = (double)pool; stats[DSA_STAT_EXACT_TILES] = tiles ? (double)tcap : 0.0;
    HIP_TRY(this, hipMemcpyAsync(x_units.p, xl.data(), xl.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemsetAsync(xinfo.p, 0, (size_t)n * 16, stream));
    for (size_t k = 0; k < xl.size(); k += pool) {
        const int m = (int)std::min(pool, xl.size() - k);
        const XReceivers rc{ rays.p, veln.p, nfield, dpl, out.p, err.p };
        launch_exact(g, batch(), x_units.p + k, m, slow.p, nrec_c, risti_c.p, X_pool.p, nrec_c, X_heap.p, gcap, lcap, x_starts.p, x_nstart.p, xinfo.p, clocks.p,
                     receivers ? &rc : nullptr, compact, stream, tiles ? &xt : nullptr, (int)gstride, lb);
    }
    HIP_TRY(this, hipGetLastError());
    std::vector<int32_t> h_x((size_t)n * 4);
    HIP_TRY(this, hipMemcpyAsync(h_x.data(), xinfo.p, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    for (int u : xl) {
        const int32_t* x = &h_x[(size_t)u * 4];
        if (x[2]) { fail(DSA_ERR_INTERNAL, "unit %d: exact march guard %d (1: tree capacity %d; 2: tile pool of %d tiles per unit too small: option exact_tile_cap, or exact_tiles = -1; 3: a store into a tile without a slot)", first + u, x[2], lcap + gcap, tcap); return DSA_ERR_INTERNAL; }
        stats[DSA_STAT_EXACT_POPS] += (double)x[0] + (double)x[1];
        h_unit_flags[(size_t)(first + u)] |= 2;
    }
    stats[DSA_STAT_EXACT_UNITS] += (double)xl.size();
    if (exact_ties == 1) {
        // the march's pool was sized from what the fixed point's buffers left (or took their place, above): it goes back, so that the next
        // call's bundles find the memory this call's found (exact_ties = 2 keeps its pool: nothing else wants the memory there) -- unless
        // (round 6) the device holds it with room to spare beside the bundles and the unit pool (a third of the memory still free: 1025^2),
        // where giving back and allocating 70 GB again on every call of a tie-prone medium cost seconds (profiles/r06_march_pool.log)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
        march_pool_kept = released_bundles_for_march ? false : free_b * 3 > total_b;
        if (!march_pool_kept) {
            auto release = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
            release(X_pool); release(X_heap); release(X_tt); release(X_tp); release(X_ring); release(X_free); release(X_pins);
        }
    }
    return 0;
}

FimLaunch Engine::launch_shape(int nnx, int nnz) const
{
    // the active band is a few node layers along the front's perimeter (peak ~13 k nodes at 1025^2)
    FimLaunch l;
    l.list_cap = list_cap > 0 ? list_cap : 16 * (nnx + nnz) + 4096;
    l.ready_cap = ready_cap > 0 ? ready_cap : 8 * (nnx + nnz) + 2048;
    // workgroup size of the solve: a front of an N x N grid is a few N nodes long and a round evaluates about a third of
    // it, so the lanes a front can use grow with N (measured: 1025^2 256 threads 7500 solves/s against 6300 with 512;
    // 2049^2 512 threads 1890 against 1430 with 256 and 1580 with 1024; 4097^2 1024 threads 407 against 240 with 256)
    const int longest = std::max(nnx, nnz);
    l.threads = fim_threads > 0 ? fim_threads : (longest > 3000 ? 1024 : longest > 1500 ? 512 : longest > 700 ? 256 : 128);   // (257^2, 513^2 and the 129^2 refined boxes: 128 threads, -10 % kernel time against 256)
    l.lds_pad = fim_lds_pad;
    // the ordered variant keeps a tile bitmap in LDS: up to 32 KB per workgroup (N <= 4097)
    const int ntile = tiles_of(nnx) * tiles_of(nnz);
    l.tile_words = (ntile + 31) / 32;
    l.sorted = (fim_sorted && l.tile_words * 4 <= 36 * 1024) ? 1 : 0;
    l.compact = 0;
    l.tie = 0;
    return l;
}

void Engine::launch_srtimes_chunk(int r0, int nr, int first_unit)
{
    launch_srtimes(g, batch(), first_unit, rays.p + r0, nr, veln.p, nfield, dpl, out.p, err.p, stream);
}

// Depth kernels of the Frechet rows: sen_*(nx*ny, kmax, nz) fp64 and the Vs model vels(nx, ny, nz)
// in the reference's layout (CalSurfG.f90:1005-1016).  on_device: the sen pointers are the engine's
// own device arrays (filled by the dispersion stage); else host arrays to upload.
int Engine::set_sensitivity(int nz, int kmax, const float* vels, const float* depz, const double* svs, const double* svp,
                            const double* srho, bool on_device)
{
    if (!have_maps) { fail(DSA_ERR_STATE, "depth kernels: call dsa_set_maps first"); return DSA_ERR_STATE; }
    if (nz < 2 || kmax < 1 || !vels || !depz) { fail(DSA_ERR_ARGUMENT, "depth kernels: bad arguments"); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    const size_t ncol = (size_t)g.nx * g.ny, n = ncol * kmax * nz;
    if (ensure(Srow, ncol * kmax * (nz - 1)) || ensure(vels_d, ncol * nz)) return status;
    HIP_TRY(this, hipMemcpyAsync(vels_d.p, vels, ncol * nz * 4, hipMemcpyHostToDevice, stream));
    if (!on_device) {
        if (!svs || !svp || !srho) { fail(DSA_ERR_ARGUMENT, "depth kernels: arrays missing"); return DSA_ERR_ARGUMENT; }
        if (ensure(sen_vs, n) || ensure(sen_vp, n) || ensure(sen_rho, n)) return status;
        HIP_TRY(this, hipMemcpyAsync(sen_vs.p, svs, n * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipMemcpyAsync(sen_vp.p, svp, n * 8, hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipMemcpyAsync(sen_rho.p, srho, n * 8, hipMemcpyHostToDevice, stream));
    }
    launch_sen_combine((int)ncol, kmax, nz, vels_d.p, sen_vs.p, sen_vp.p, sen_rho.p, depz[nz - 2] < 35.0f ? 1 : 0, Srow.p, stream);
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipStreamSynchronize(stream));
    sens_nz = nz; sens_kmax = kmax; have_sens = true;
    return 0;
}

// rays and rows of the resident chunk [first_unit, first_unit + n): launches of at most `per`
// rays (slab memory), each: zero slabs -> trace -> list -> count -> scan -> write -> copy out
int Engine::trace_chunk(int first_unit, int n, float* rw, int* iw, int* col, long long cap, long long* nar)
{
    const int r0 = h_src[first_unit].first_ray;
    const int r1 = h_src[first_unit + n - 1].first_ray + h_src[first_unit + n - 1].nrec;
    const size_t t0 = std::lower_bound(h_trace.begin(), h_trace.end(), r0) - h_trace.begin();
    const size_t t1 = std::lower_bound(h_trace.begin(), h_trace.end(), r1) - h_trace.begin();
    if (t1 <= t0) return 0;
    const size_t slab_stride = (size_t)(g.nvx + 2) * (g.nvz + 2), vlist_stride = (size_t)g.nvx * g.nvz;
    const size_t per_ray = (slab_stride + vlist_stride) * 4 + 32;
    size_t budget = ray_budget;
    if (!budget) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
        budget = std::min<size_t>((size_t)40 << 30, free_b / 3 + slabs.cap * 4 + vlist.cap * 4);
    }
    size_t per = std::max<size_t>(budget / per_ray, 64);
    per = std::max<size_t>(std::min(per, t1 - t0), 1);
    if (ensure(slabs, per * slab_stride) || ensure(vlist, per * vlist_stride) || ensure(nvv, per) || ensure(counts, per) ||
        ensure(offsets, per + 1) || ensure(rayinfo, per * 2)) return status;
    std::vector<int32_t> h_info;
    hipEvent_t ea = events[1], eb = events[2], ec = events[3];
    for (size_t t = t0; t < t1; t += per) {
        const int m = (int)std::min(per, t1 - t);
        HIP_TRY(this, hipEventRecord(ea, stream));
        HIP_TRY(this, hipMemsetAsync(slabs.p, 0, (size_t)m * slab_stride * 4, stream));
        float* d_paths = nullptr;
        int* d_path_n = nullptr;
        if (ray_path_cap > 0) {     // the store covers all traced rays of the plan (position in h_trace)
            if (ensure(paths, h_trace.size() * (size_t)ray_path_cap * 2) || ensure(path_n, h_trace.size())) return status;
            d_paths = paths.p + t * (size_t)ray_path_cap * 2; d_path_n = path_n.p + t;
        }
        launch_rays(g, batch(), first_unit, rays.p, trace_ids.p + t, m, veln.p, nfield, dpl, slabs.p, slab_stride, rayinfo.p, err.p,
                    d_paths, ray_path_cap, d_path_n, stream, ray_lanes_opt ? ray_lanes_opt : (m <= kRayGroupMax ? 4 : 1));
        HIP_TRY(this, hipEventRecord(eb, stream));
        RowArgs a{};
        a.rays = rays.p; a.trace_ids = trace_ids.p + t; a.n = m; a.src = src.p; a.unit_base = first_unit;
        a.slabs = slabs.p; a.slab_stride = slab_stride; a.vlist = vlist.p; a.vlist_stride = vlist_stride; a.nv = nvv.p;
        a.S = Srow.p; a.kmax = sens_kmax; a.nz = sens_nz; a.counts = counts.p; a.offsets = offsets.p;
        launch_row_list(g, a, stream);
        launch_row_emit(g, a, false, stream);
        launch_scan(counts.p, m, offsets.p, stream);
        long long total = 0;
        HIP_TRY(this, hipMemcpyAsync(&total, offsets.p + m, 8, hipMemcpyDeviceToHost, stream));
        h_info.resize((size_t)m * 2);
        HIP_TRY(this, hipMemcpyAsync(h_info.data(), rayinfo.p, (size_t)m * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(this, hipStreamSynchronize(stream));
        if (*nar + total > cap) { fail(DSA_ERR_CAPACITY, "Frechet rows need more than the %lld entries provided", cap); return DSA_ERR_CAPACITY; }
        if (total > 0 && grow_rw && grow_iw && grow_col) {
            grow_rw->resize((size_t)(*nar + total)); grow_iw->resize((size_t)(*nar + total)); grow_col->resize((size_t)(*nar + total));
            rw = grow_rw->data(); iw = grow_iw->data(); col = grow_col->data();
        }
        if (total > 0 && rows_on_device) {
            // the launch writes its rows straight behind the ones already resident; the host copy is optional
            const size_t need = (size_t)(*nar + total);
            if (ensure_keep(G_rw, need, (size_t)*nar) || ensure_keep(G_row, need, (size_t)*nar) || ensure_keep(G_col, need, (size_t)*nar)) return status;
            a.rw = G_rw.p + *nar; a.iw = G_row.p + *nar; a.col = G_col.p + *nar;
            launch_row_emit(g, a, true, stream);
            if (rw && iw && col) {
                HIP_TRY(this, hipMemcpyAsync(rw + *nar, a.rw, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(this, hipMemcpyAsync(iw + *nar, a.iw, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
                HIP_TRY(this, hipMemcpyAsync(col + *nar, a.col, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
            }
            G_nar = *nar + total;
        } else if (total > 0) {
            if (ensure(coo_rw, (size_t)total) || ensure(coo_iw, (size_t)total) || ensure(coo_col, (size_t)total)) return status;
            a.rw = coo_rw.p; a.iw = coo_iw.p; a.col = coo_col.p;
            launch_row_emit(g, a, true, stream);
            HIP_TRY(this, hipMemcpyAsync(rw + *nar, coo_rw.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(this, hipMemcpyAsync(iw + *nar, coo_iw.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(this, hipMemcpyAsync(col + *nar, coo_col.p, (size_t)total * 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(this, hipEventRecord(ec, stream));
        HIP_TRY(this, hipStreamSynchronize(stream));
        HIP_TRY(this, hipGetLastError());
        *nar += total;
        float ms = 0;
        HIP_TRY(this, hipEventElapsedTime(&ms, ea, eb)); stats[DSA_STAT_MS_RAYS] += ms;
        HIP_TRY(this, hipEventElapsedTime(&ms, eb, ec)); stats[DSA_STAT_MS_ROWS] += ms;
        stats[DSA_STAT_RAYS] += m;
        for (int q = 0; q < m; ++q) {
            stats[DSA_STAT_RAY_STEPS] += h_info[2 * q + 1]; stats[DSA_STAT_RAYS_CLAMPED] += h_info[2 * q] & 1;
            if (h_info[2 * q] & 1) {
                const int unit = h_rays[(size_t)h_trace[t + (size_t)q]].src;
                rays_clamped += 1;
                if (first_clamped_unit < 0 || unit < first_clamped_unit) first_clamped_unit = unit;
            }
        }
    }
    stats[DSA_STAT_NAR] = (double)*nar;
    return 0;
}

// download tiled records and untile on the host: which = 0 -> T (raw), 1 -> tau (raw)
int Engine::fetch_tiled(const Rec* dev, int nnx, int nnz, int which, float* out)
{
    const int nbx = tiles_of(nnx), nbz = tiles_of(nnz);
    std::vector<Rec> h((size_t)nbx * nbz * kTileRecs);
    HIP_TRY(this, hipMemcpy(h.data(), dev, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    for (int ix = 0; ix < nnx; ++ix)
        for (int iz = 0; iz < nnz; ++iz) {
            const Rec r = h[rec_index(nbz, iz, ix)];
            out[(size_t)ix * nnz + iz] = which ? r.tau : r.T;
        }
    return 0;
}

// coarse field of a resident unit, untiled: which = 0 -> T (sign bit = pinned), 1 -> tau (from the field and its exception table)
int Engine::fetch_compact(int slot, int which, float* out)
{
    std::vector<float> h(nrec_c);
    std::vector<unsigned long long> x((size_t)1 << exc_log2cap);
    HIP_TRY(this, hipMemcpy(h.data(), T_c.p + (size_t)slot * nrec_c, nrec_c * 4, hipMemcpyDeviceToHost));
    HIP_TRY(this, hipMemcpy(x.data(), exc_c.p + ((size_t)slot << exc_log2cap), x.size() * 8, hipMemcpyDeviceToHost));
    for (int ix = 0; ix < g.nnx; ++ix)
        for (int iz = 0; iz < g.nnz; ++iz) {
            const int id = rec_index(g.nbz, iz, ix);
            const float v = h[id];
            float T = v, tau = v;
            if (__builtin_signbit(v)) {
                bool pin = false;
                tau = exc_find(x.data(), exc_log2cap, id, &pin);
                T = pin ? v : -v;
            }
            out[(size_t)ix * g.nnz + iz] = which ? tau : T;
        }
    return 0;
}

int Engine::get_field(int unit, float* ttn)
{
    if (last_chunk_first < 0 || !fields_resident || unit < last_chunk_first || unit >= last_chunk_first + last_chunk_n) { fail(DSA_ERR_STATE, "get_field: unit %d is not resident (last chunk covers %d..%d%s)", unit, last_chunk_first, last_chunk_first + last_chunk_n - 1, fields_resident ? "" : "; its field slots were recycled"); return DSA_ERR_STATE; }
    HIP_TRY(this, hipSetDevice(device));
    if (fetch_compact(unit - last_chunk_first, 0, ttn)) return status;
    for (size_t k = 0; k < nfield; ++k) ttn[k] = fabsf(ttn[k]);
    return 0;
}

int Engine::get_refined(int unit, int* rnx, int* rnz, float* ttnr, int8_t* st)
{
    if (last_chunk_first < 0 || unit < last_chunk_first || unit >= last_chunk_first + last_chunk_n) { fail(DSA_ERR_STATE, "get_refined: unit %d is not resident", unit); return DSA_ERR_STATE; }
    HIP_TRY(this, hipSetDevice(device));
    const SourceDesc& s = h_src[unit];
    const size_t rr = (size_t)kRefMax * kRefMax, n = (size_t)s.rnx * s.rnz;
    *rnx = s.rnx; *rnz = s.rnz;
    HIP_TRY(this, hipMemcpy(ttnr, Tfin_r.p + (size_t)(unit - last_chunk_first) * rr, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(this, hipMemcpy(st, S_r.p + (size_t)(unit - last_chunk_first) * rr, n, hipMemcpyDeviceToHost));
    return 0;
}

int Engine::get_velocity(int map, float* out_v)
{
    if (!have_maps || map < 0 || map >= nmaps) { fail(DSA_ERR_ARGUMENT, "get_velocity: map %d", map); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    HIP_TRY(this, hipMemcpy(out_v, veln.p + (size_t)map * nfield, nfield * 4, hipMemcpyDeviceToHost));
    return 0;
}

}  // namespace dsa

// ---------------------------------------------------------------------------------------------
using dsa::Engine;

extern "C" {

int dsa_create(dsa_engine** out, int device_index)
{
    if (!out) return DSA_ERR_ARGUMENT;
    *out = nullptr;
    Engine* e = new Engine();
    const int rc = e->init(device_index);
    if (rc != 0) { dsa::g_create_error = e->error; delete e; return rc; }
    *out = reinterpret_cast<dsa_engine*>(e);
    return 0;
}

void dsa_destroy(dsa_engine* e) { delete reinterpret_cast<Engine*>(e); }

const char* dsa_error_string(const dsa_engine* e)
{
    if (!e) return dsa::g_create_error.c_str();
    return reinterpret_cast<const Engine*>(e)->error.c_str();
}

int dsa_set_memory_budget(dsa_engine* e, size_t bytes)
{
    if (!e) return DSA_ERR_ARGUMENT;
    reinterpret_cast<Engine*>(e)->mem_budget = bytes;
    return 0;
}

int dsa_set_option(dsa_engine* e, const char* name, double value)
{
    if (!e || !name) return DSA_ERR_ARGUMENT;
    Engine* en = reinterpret_cast<Engine*>(e);
    const std::string n(name);
    if (n == "window_cells" && value > 0) { en->window_cells = (float)value; return 0; }
    if (n == "max_chunk" && value >= 0) { en->max_chunk = (int)value; return 0; }
    if (n == "list_cap" && value >= 0) { en->planned = false; en->list_cap = (int)value; return 0; }
    if (n == "ready_cap" && value >= 0) { en->planned = false; en->ready_cap = (int)value; return 0; }
    if (n == "ray_budget" && value >= 0) { en->ray_budget = (size_t)value; return 0; }
    if (n == "fim_lds_pad" && value >= 0 && value <= 64 * 1024) { en->planned = false; en->fim_lds_pad = (int)value; return 0; }
    if (n == "fim_sorted" && (value == 0 || value == 1)) { en->planned = false; en->fim_sorted = (int)value; return 0; }   // (refined boxes only: the coarse solve exists in the ordered variant only)
    if (n == "ray_path_cap" && value >= 0 && value <= (1 << 24)) { en->ray_path_cap = (int)value; return 0; }
    if (n == "disp_group_shift" && value >= -1 && value <= 3) { en->disp_group_shift = (int)value; return 0; }
    if (n == "disp_layers_lds" && (value == -1 || value == 0 || value == 1)) { en->disp_layers_lds = (int)value; return 0; }
    if (n == "exc_log2cap" && (value == 0 || (value >= 6 && value <= 24))) { en->planned = false; en->exc_log2cap_opt = (int)value; return 0; }
    if (n == "rows_on_device" && (value == 0 || value == 1)) { en->rows_on_device = value != 0; return 0; }
    if (n == "lsmr_device_vectors" && (value == 0 || value == 1)) { en->lsmr_device_vectors = (int)value; return 0; }
    if (n == "field_pool" && value >= -1) { en->planned = false; en->field_pool_opt = (int)value; return 0; }
    if (n == "ray_lanes" && (value == 0 || value == 1 || value == 4)) { en->ray_lanes_opt = (int)value; return 0; }
    if (n == "bundle_window_cells" && value >= 0) { en->bundle_window_opt = (float)value; return 0; }
    if (n == "bundle_threads" && (value == 0 || value == 256 || value == 512 || value == 768)) { en->bundle_threads_opt = (int)value; return 0; }
    if (n == "bundle_max_rounds" && value >= 0) { en->bundle_max_rounds = (int)value; return 0; }
    if (n == "bundle_tail" && (value == 0 || value == 1)) { en->bundle_tail_opt = (int)value; return 0; }
    if (n == "bundle_refined" && (value == 0 || value == 1 || value == 2)) { en->bundle_refined_opt = (int)value; return 0; }
    if (n == "bundle_far_all" && (value == 0 || value == 1)) { en->bundle_far_all = (int)value; return 0; }
    if (n == "bundle_pool" && value >= 0) { en->bundle_pool_opt = (int)value; return 0; }
    if (n == "bundle_members_per_lane" && (value == 0 || value == 4 || value == 2)) { en->bundle_mpl = (int)value; return 0; }
    if (n == "bundle" && (value == 0 || value == 1 || value == 4 || value == 8 || value == 16)) { en->planned = false; en->bundle_opt = (int)value; return 0; }
    if (n == "exact_ties" && (value == 0 || value == 1 || value == 2)) { en->exact_ties = (int)value; return 0; }
    if (n == "tie_threshold" && value >= 0) { en->tie_threshold = (float)value; return 0; }
    if (n == "tie_detect" && (value == 0 || value == 1)) { en->tie_detect = (int)value; return 0; }
    if (n == "tie_sum_threshold" && value >= 0) { en->tie_sum_threshold = (float)value; return 0; }
    if (n == "tie_count_threshold" && value >= 0) { en->tie_count_threshold = (int)value; return 0; }
    if (n == "tie_frozen_bundles" && (value == 0 || value == 1)) { en->tie_frozen_bundles = (int)value; return 0; }
    if (n == "tie_map_strict" && (value == 0 || value == 1)) { en->tie_map_strict = (int)value; return 0; }
    if (n == "tie_scale_guard" && (value == 0 || value == 1)) { en->tie_scale_guard = (int)value; return 0; }
    if (n == "handoff_replay" && (value == 0 || value == 1)) { en->handoff_replay = (int)value; return 0; }
    if (n == "tie_tolerance" && value > 0) { en->tie_tolerance = (float)value; return 0; }
    if (n == "bundle_order" && (value == 0 || value == 1 || value == 2 || value == 3)) { en->bundle_order_opt = (int)value; return 0; }
    if (n == "tie_list" && (value == 0 || value == 1)) { en->tie_list_opt = (int)value; return 0; }
    if (n == "disp_failure_log" && value >= 0 && value <= 65536) { en->disp_failure_log = (int)value; return 0; }
    if (n == "exact_heap_blocked" && (value == 0 || value == 1 || value == 2)) { en->exact_heap_blocked = (int)value; return 0; }
    if (n == "exact_lds_slots" && (value == 0 || (value >= 63 && value <= 4975))) { en->exact_lds_slots = (int)value; return 0; }
    if (n == "exact_tiles" && (value == -1 || value == 0 || value == 1)) { en->exact_tiles_opt = (int)value; return 0; }
    if (n == "exact_tile_cap" && (value == 0 || (value >= 64 && value <= 65000))) { en->exact_tile_cap = (int)value; return 0; }
    if (n == "exact_pool" && value >= 0 && value <= 65535) { en->exact_pool = (int)value; return 0; }
    if (n == "exact_pool_max" && value >= 4 && value <= 32768) { en->exact_pool_max = (size_t)value; return 0; }
    if (n == "fim_threads" && (value == 0 || value == 128 || value == 256 || value == 512 || value == 1024)) { en->planned = false; en->fim_threads = (int)value; return 0; }
    en->fail(DSA_ERR_ARGUMENT, "unknown option or bad value: %s=%g", name, value);
    return DSA_ERR_ARGUMENT;
}

// ray paths of the last dsa_solve_rows (option ray_path_cap > 0): for every traced ray, in data order, its datum, its
// point count and up to ray_path_cap (latitude, longitude) pairs in degrees -- the values the reference's disabled dump
// writes to raypath.out (CalSurfG.f90:2276-2283: rayx = (pi/2 - rgx) 180/pi, rayz = rgz 180/pi, in single precision)
int dsa_ray_paths(dsa_engine* e, int* datum, int* npts, float* latlon)
{
    if (!e || !datum || !npts || !latlon) return DSA_ERR_ARGUMENT;
    Engine* en = reinterpret_cast<Engine*>(e);
    const size_t nr = en->h_trace.size(), cap = (size_t)en->ray_path_cap;
    if (cap == 0 || !en->paths.p || !en->path_n.p || en->paths.cap < nr * cap * 2) { en->fail(DSA_ERR_STATE, "ray_paths: set option ray_path_cap and call dsa_solve_rows first"); return DSA_ERR_STATE; }
    if (hipSetDevice(en->device) != hipSuccess || hipMemcpy(npts, en->path_n.p, nr * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(latlon, en->paths.p, nr * cap * 8, hipMemcpyDeviceToHost) != hipSuccess) { en->fail(DSA_ERR_DEVICE, "ray_paths: copy failed"); return DSA_ERR_DEVICE; }
    const float pi = 3.1415926535898f;                        // CalSurfG.f90:196
    for (size_t r = 0; r < nr; ++r) {
        datum[r] = en->h_rays[(size_t)en->h_trace[r]].data + 1;
        const size_t np = std::min<size_t>((size_t)std::max(npts[r], 0), cap);
        for (size_t k = 0; k < np; ++k) {
            float* q = latlon + (r * cap + k) * 2;
            const float x = q[0], z = q[1];
            q[0] = (pi / 2 - x) * 180.0f / pi;
            q[1] = z * 180.0f / pi;
        }
    }
    return 0;
}

int dsa_set_maps(dsa_engine* e, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int dicing, int nmaps, const double* pv)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->set_maps(nx, ny, goxd, gozd, dvxd, dvzd, dicing, nmaps, pv);
}

int dsa_plan(dsa_engine* e, int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec, const float* rcx, const float* rcz)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->plan(nunits, map_index, scx, scz, nrec, rcx, rcz, nullptr, nullptr, nullptr);
}

int dsa_plan_units(dsa_engine* e, int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec, const float* rcx, const float* rcz,
                   const int* mode, const int* sen_slot, const int* data_first)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->plan(nunits, map_index, scx, scz, nrec, rcx, rcz, mode, sen_slot, data_first);
}

int dsa_set_depth_kernels(dsa_engine* e, int nz, int kmax, const float* vels, const float* depz, const double* sen_vs, const double* sen_vp, const double* sen_rho)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->set_sensitivity(nz, kmax, vels, depz, sen_vs, sen_vp, sen_rho, false);
}

int dsa_dispersion_begin(dsa_engine* e, int nx, int ny, int nz, const float* vels, const float* depz, float minthk, int kmax_total, int nmaps_total)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->dispersion_begin(nx, ny, nz, vels, depz, minthk, kmax_total, nmaps_total);
}

int dsa_dispersion_run(dsa_engine* e, int iwave, int igr, int nper, const double* t, int with_kernels, int sen_slot, int map_first)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->dispersion_run(iwave, igr, nper, t, with_kernels, sen_slot, map_first);
}

int dsa_dispersion_copy_maps(dsa_engine* e, int from, int to, int n)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->dispersion_copy_map(from, to, n);
}

int dsa_dispersion_fetch(dsa_engine* e, int map_first, int nper, double* pv, int with_kernels, int sen_slot, double* sen_vs, double* sen_vp, double* sen_rho)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->dispersion_fetch(map_first, nper, pv, with_kernels, sen_slot, sen_vs, sen_vp, sen_rho);
}

int dsa_maps_from_dispersion(dsa_engine* e, float goxd, float gozd, float dvxd, float dvzd, int dicing)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->maps_from_dispersion(goxd, gozd, dvxd, dvzd, dicing);
}

int dsa_kernels_from_dispersion(dsa_engine* e)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->kernels_from_dispersion();
}

int dsa_solve(dsa_engine* e, float* dsurf)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->solve(dsurf, nullptr, nullptr, nullptr, 0, nullptr);
}

// the receiver times left on the device: d_dsurf is a device pointer (on the engine's device) to ndata floats; what the N-rank path
// hands to the RCCL all-gather without a detour through the host (north_star: all-gather of the travel times over xGMI)
int dsa_solve_device(dsa_engine* e, void* d_dsurf)
{
    if (!e || !d_dsurf) return DSA_ERR_ARGUMENT;
    Engine* en = reinterpret_cast<Engine*>(e);
    const int rc = en->solve(nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    if (rc != 0) return rc;
    if (en->ndata == 0) return 0;
    if (hipSetDevice(en->device) != hipSuccess ||
        hipMemcpyAsync(d_dsurf, en->out.p, en->ndata * sizeof(float), hipMemcpyDeviceToDevice, en->stream) != hipSuccess ||
        hipStreamSynchronize(en->stream) != hipSuccess) { en->fail(DSA_ERR_DEVICE, "solve_device: copy into the caller's device buffer failed"); return DSA_ERR_DEVICE; }
    return 0;
}

int dsa_solve_rows(dsa_engine* e, float* dsurf, float* rw, int* iw, int* col, long long capacity, long long* nar)
{
    if (!e || !nar) return DSA_ERR_ARGUMENT;
    Engine* en = reinterpret_cast<Engine*>(e);
    // with option rows_on_device the three arrays may be null: the rows stay on the device (dsa_iteration_system_device)
    if ((!rw || !iw || !col) && !en->rows_on_device) return DSA_ERR_ARGUMENT;
    if (!rw || !iw || !col) { rw = nullptr; iw = nullptr; col = nullptr; }
    return en->solve(dsurf, rw, iw, col, capacity, nar);
}

int dsa_get_dims(const dsa_engine* e, int* nnx, int* nnz)
{
    if (!e || !nnx || !nnz) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    *nnx = en->g.nnx; *nnz = en->g.nnz;
    return 0;
}

int dsa_keep_fields(dsa_engine* e, int on)
{
    if (!e) return DSA_ERR_ARGUMENT;
    // with keep_fields every planned unit must stay resident (dsa_get_field / dsa_get_refined of any unit): dsa_plan fails
    // with DSA_ERR_CAPACITY when the units do not fit one chunk
    reinterpret_cast<Engine*>(e)->keep_fields = on != 0;
    reinterpret_cast<Engine*>(e)->planned = false;
    return 0;
}

int dsa_get_field(dsa_engine* e, int unit, float* ttn)
{
    if (!e || !ttn) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->get_field(unit, ttn);
}

int dsa_get_velocity(dsa_engine* e, int map, float* veln)
{
    if (!e || !veln) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->get_velocity(map, veln);
}

int dsa_get_refined(dsa_engine* e, int unit, int* rnx, int* rnz, float* ttnr, int8_t* status)
{
    if (!e || !rnx || !rnz || !ttnr || !status) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->get_refined(unit, rnx, rnz, ttnr, status);
}

int dsa_debug_field(dsa_engine* e, int unit, int which, float* out)
{
    if (!e || !out) return DSA_ERR_ARGUMENT;
    Engine* en = reinterpret_cast<Engine*>(e);
    if (en->last_chunk_first < 0 || unit < en->last_chunk_first || unit >= en->last_chunk_first + en->last_chunk_n) return DSA_ERR_STATE;
    if (which < 2 && !en->fields_resident) return DSA_ERR_STATE;
    const size_t slot = (size_t)(unit - en->last_chunk_first);
    if (which < 2) return en->fetch_compact((int)slot, which, out);
    const dsa::SourceDesc& s = en->h_src[unit];
    return en->fetch_tiled(en->F_r.p + slot * dsa::kRefRecs, s.rnx, s.rnz, which - 2, out);
}

// probe builds: the 24 trip counters of a DSA_LEDGER build (tools/isa_ledger.py), summed over the units of the last solve
int dsa_debug_counters(const dsa_engine* e, double* out24)
{
    if (!e || !out24) return DSA_ERR_ARGUMENT;
    for (int q = 0; q < 24; ++q) out24[q] = reinterpret_cast<const Engine*>(e)->phase_ticks[8 + q];
    return 0;
}

int dsa_dispersion_failure(const dsa_engine* e, int index, int* info, double* vals, float* table, double* c)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<const Engine*>(e)->dispersion_failure(index, info, vals, table, c);
}

int dsa_dispersion_diagnostics(const dsa_engine* e, long long* count, int* first, double* period)
{
    if (!e) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    if (count) *count = en->disp_fail_count;
    if (first) for (int q = 0; q < 5; ++q) first[q] = en->disp_fail_count ? en->disp_fail_first[q] : 0;
    if (period) *period = en->disp_fail_count ? en->disp_fail_period : 0.0;
    return 0;
}

int dsa_ray_diagnostics(const dsa_engine* e, long long* clamped, int* first_unit)
{
    if (!e) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    if (clamped) *clamped = en->rays_clamped;
    if (first_unit) *first_unit = en->first_clamped_unit;
    return 0;
}

int dsa_unit_tie_sums(const dsa_engine* e, int nunits, int* count, float* sum, int* frozen)
{
    if (!e) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    if (nunits < 0 || (size_t)nunits > en->h_unit_tie_sum.size()) return DSA_ERR_ARGUMENT;
    for (int u = 0; u < nunits; ++u) {
        if (count) count[u] = en->h_unit_tie_count[(size_t)u];
        if (sum) sum[u] = en->h_unit_tie_sum[(size_t)u];
        if (frozen) frozen[u] = en->h_unit_froze[(size_t)u];
    }
    return 0;
}

int dsa_unit_ties(const dsa_engine* e, int nunits, int* flags, float* influence)
{
    if (!e) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    if (nunits < 0 || (size_t)nunits > en->h_unit_flags.size()) return DSA_ERR_ARGUMENT;
    for (int u = 0; u < nunits; ++u) { if (flags) flags[u] = en->h_unit_flags[(size_t)u]; if (influence) influence[u] = en->h_unit_tie[(size_t)u]; }
    return 0;
}

int dsa_unit_rounds(const dsa_engine* e, int nunits, int* rounds)
{
    if (!e || !rounds) return DSA_ERR_ARGUMENT;
    const Engine* en = reinterpret_cast<const Engine*>(e);
    if (nunits < 0 || (size_t)nunits > en->h_unit_rounds.size()) return DSA_ERR_ARGUMENT;
    for (int u = 0; u < nunits; ++u) rounds[u] = en->h_unit_rounds[(size_t)u];
    return 0;
}

int dsa_get_stats(const dsa_engine* e, double* out)
{
    if (!e || !out) return DSA_ERR_ARGUMENT;
    std::memcpy(out, reinterpret_cast<const Engine*>(e)->stats, sizeof(double) * DSA_STAT_COUNT);
    // phase clocks of the coarse solve, summed over units (100 MHz wall clock ticks): pass A, even half,
    // odd half, round end; then the summed list lengths and ready counts
    for (int q = 0; q < 8; ++q) out[DSA_STAT_COUNT + q] = reinterpret_cast<const Engine*>(e)->phase_ticks[q];
    return 0;
}

}  // extern "C"
