// Host side of the engine-level C ABI (include/dsurftomo_amd.h): owns device memory, turns the
// caller's (map, source, receivers) units into descriptors, and sequences the kernels per chunk
// of sources on one HIP stream.  No numerical work happens here except exact fp32 geometry and
// the libm sine tables (host_geometry.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
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

Engine::~Engine()
{
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    rel(velv); rel(veln); rel(slow); rel(risti_c); rel(cbasis); rel(rbasis);
    rel(src); rel(rays); rel(out); rel(err);
    rel(slow_r); rel(F_r); rel(Tfin_r); rel(S_r); rel(risti_r); rel(vcorner); rel(seed_r); rel(nseed_r);
    rel(rst); rel(cst); rel(cinit); rel(heap); rel(flags); rel(F_c); rel(seed_c); rel(nseed_c);
    rel(prob_r); rel(prob_c); rel(info); rel(clocks); rel(lists);
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
    if (g.nnx > 32767 || g.nnz > 32767) { fail(DSA_ERR_ARGUMENT, "grid %dx%d exceeds the 32767-node index range", g.nnx, g.nnz); return DSA_ERR_ARGUMENT; }
    nmaps = nm;
    nfield = (size_t)g.nnx * g.nnz;
    nrec_c = (size_t)g.nbx * g.nbz * kTileRecs;
    const size_t nv = (size_t)nx * ny;
    std::vector<float> hv(nv * nm);
    for (size_t k = 0; k < nv * nm; ++k) hv[k] = (float)pv[k];       // velv = real(pv), CalSurfG.f90:1492
    hmin_slow = 1e30f;
    for (float v : hv) if (v > 0.0f && 1.0f / v < hmin_slow) hmin_slow = 1.0f / v;
    std::vector<float> cb(4 * (dicing + 1)), rb(4 * (dicing * kSgdl + 1)), rc(g.nnx);
    basis_table(dicing, cb.data());
    basis_table(dicing * kSgdl, rb.data());
    risti_table(g.gox, g.dnx, g.earth, g.nnx, rc.data());
    dpl = min_cell_km(g);
    if (ensure(velv, hv.size()) || ensure(veln, nfield * nm) || ensure(slow, nrec_c * nm) || ensure(risti_c, rc.size()) ||
        ensure(cbasis, cb.size()) || ensure(rbasis, rb.size())) return status;
    HIP_TRY(this, hipMemcpyAsync(velv.p, hv.data(), hv.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(cbasis.p, cb.data(), cb.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(rbasis.p, rb.data(), rb.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipMemcpyAsync(risti_c.p, rc.data(), rc.size() * 4, hipMemcpyHostToDevice, stream));
    for (int m = 0; m < nm; ++m)
        launch_gridder(g, velv.p + nv * m, cbasis.p, veln.p + nfield * m, slow.p + nrec_c * m, stream);
    HIP_TRY(this, hipGetLastError());
    HIP_TRY(this, hipStreamSynchronize(stream));
    planned = false;
    have_maps = true;
    return 0;
}

int Engine::plan(int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec,
                 const float* rcx, const float* rcz)
{
    if (!have_maps) { fail(DSA_ERR_STATE, "plan: call dsa_set_maps first"); return DSA_ERR_STATE; }
    if (nunits < 0 || (nunits > 0 && (!map_index || !scx || !scz || !nrec))) { fail(DSA_ERR_ARGUMENT, "plan: bad arguments"); return DSA_ERR_ARGUMENT; }
    HIP_TRY(this, hipSetDevice(device));
    h_src.resize(nunits);
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
        if (nrec[u] < 0) { fail(DSA_ERR_ARGUMENT, "plan: negative receiver count"); return DSA_ERR_ARGUMENT; }
        nr += (size_t)nrec[u];
        risti_table(s.rgox, s.rdnx, g.earth, s.rnx, &h_risti_r[(size_t)u * kRefMax]);
    }
    if (nr > 0 && (!rcx || !rcz)) { fail(DSA_ERR_ARGUMENT, "plan: receivers missing"); return DSA_ERR_ARGUMENT; }
    h_rays.resize(nr);
    for (int u = 0; u < nunits; ++u)
        for (int k = 0; k < h_src[u].nrec; ++k) {
            const size_t r = (size_t)h_src[u].first_ray + k;
            const float rx = rcx[r], rz = rcz[r];
            const int irx = (int)((rx - g.gox) / g.dnx) + 1, irz = (int)((rz - g.goz) / g.dnz) + 1;
            if (irx < 1 || irx > g.nnx || irz < 1 || irz > g.nnz) {
                fail(DSA_ERR_OUTSIDE, "Receiver lies outside model (lat,long)= %g %g", 90.0 - rx * 180.0 / kPi, rz * 180.0 / kPi);
                return DSA_ERR_OUTSIDE;
            }
            h_rays[r] = RayDesc{ u, rx, rz, sinf(rx) };
        }
    // chunk size from the memory budget
    size_t free_b = 0, total_b = 0;
    HIP_TRY(this, hipMemGetInfo(&free_b, &total_b));
    size_t budget = mem_budget ? mem_budget : (size_t)(0.6 * (double)free_b);
    const size_t rr = (size_t)kRefMax * kRefMax;
    per_unit_bytes = nrec_c * 8 + (size_t)(40 * (g.nnx + g.nnz) + 10240) * 4 + (size_t)kRefRecs * 12 + rr * 5 + kRefMax * 4 + (size_t)kSeedR * 4 + (size_t)kSeedC * 4 + kRWin * kRWin * 2 +
                     (size_t)kCWinMax * kCWinMax * 3 + kHeapCap * 4 + 256 + sizeof(FimProblem) * 2 + sizeof(SourceDesc);
    size_t c = budget / per_unit_bytes;
    if (c < 1) { fail(DSA_ERR_DEVICE, "memory budget %zu B cannot hold one source (%zu B)", budget, per_unit_bytes); return DSA_ERR_DEVICE; }
    chunk = (int)std::min<size_t>(c, (size_t)std::max(nunits, 1));
    if (max_chunk > 0) chunk = std::min(chunk, max_chunk);
    const size_t C = (size_t)chunk;
    { const FimLaunch lc = launch_shape(g.nnx, g.nnz); lists_stride = (size_t)2 * lc.list_cap + lc.ready_cap; }
    if (ensure(lists, C * lists_stride) || ensure(src, C) || ensure(rays, std::max<size_t>(nr, 1)) || ensure(out, std::max<size_t>(nr, 1)) || ensure(err, 4) ||
        ensure(slow_r, C * kRefRecs) || ensure(F_r, C * kRefRecs) || ensure(Tfin_r, C * rr) || ensure(S_r, C * rr) ||
        ensure(risti_r, C * kRefMax) || ensure(vcorner, C * 4) || ensure(seed_r, C * kSeedR) || ensure(nseed_r, C) ||
        ensure(rst, C * kRWin * kRWin) || ensure(cst, C * kCWinMax * kCWinMax) || ensure(cinit, C * kCWinMax * kCWinMax) ||
        ensure(heap, C * kHeapCap) || ensure(flags, C * 4) || ensure(F_c, C * nrec_c) ||
        ensure(seed_c, C * kSeedC) || ensure(nseed_c, C) || ensure(prob_r, C) || ensure(prob_c, C) || ensure(info, C * 16) || ensure(clocks, C * 8)) return status;
    if (nr) HIP_TRY(this, hipMemcpyAsync(rays.p, h_rays.data(), nr * sizeof(RayDesc), hipMemcpyHostToDevice, stream));
    HIP_TRY(this, hipStreamSynchronize(stream));
    planned = true;
    last_chunk_first = -1;
    return 0;
}

BatchPtrs Engine::batch() const
{
    BatchPtrs b;
    b.src = src.p; b.slow_r = slow_r.p; b.F_r = F_r.p; b.Tfin_r = Tfin_r.p; b.S_r = S_r.p; b.risti_r = risti_r.p;
    b.vcorner = vcorner.p; b.seed_r = seed_r.p; b.nseed_r = nseed_r.p; b.rst = rst.p; b.cst = cst.p; b.cinit = cinit.p;
    b.heap = heap.p; b.flags = flags.p; b.F_c = F_c.p; b.seed_c = seed_c.p; b.nseed_c = nseed_c.p;
    b.lists = lists.p; b.lists_stride = lists_stride;
    return b;
}

int Engine::solve(float* dsurf)
{
    if (!planned) { fail(DSA_ERR_STATE, "solve: call dsa_plan first"); return DSA_ERR_STATE; }
    HIP_TRY(this, hipSetDevice(device));
    const int nunits = (int)h_src.size();
    std::fill(stats, stats + DSA_STAT_COUNT, 0.0);
    std::fill(phase_ticks, phase_ticks + 8, 0.0);
    stats[DSA_STAT_UNITS] = nunits;
    stats[DSA_STAT_CHUNK] = chunk;
    // causal window: a few cells' worth of travel time (narrowest cell, fastest velocity of the model)
    const float cell_c = dpl * hmin_slow;
    const float window_c = window_cells * cell_c;
    const float window_r = window_cells * cell_c / (float)kSgdl;
    HIP_TRY(this, hipMemsetAsync(err.p, 0, 4 * sizeof(int32_t), stream));
    HIP_TRY(this, hipEventRecord(events[0], stream));
    std::vector<int32_t> h_info, h_flags;
    for (int first = 0; first < nunits; first += chunk) {
        const int n = std::min(chunk, nunits - first);
        const BatchPtrs b = batch();
        HIP_TRY(this, hipMemcpyAsync(src.p, h_src.data() + first, (size_t)n * sizeof(SourceDesc), hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipMemcpyAsync(risti_r.p, h_risti_r.data() + (size_t)first * kRefMax, (size_t)n * kRefMax * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(this, hipEventRecord(events[1], stream));
        launch_fill(reinterpret_cast<float*>(F_c.p), (size_t)n * nrec_c * 2, kInf, stream);      // (T, tau) = (+inf, +inf)
        launch_make_problems(g, b, n, slow.p, nrec_c, risti_c.p, window_r, window_c, prob_r.p, prob_c.p, info.p, clocks.p, stream);
        launch_refine(g, b, n, velv.p, (size_t)g.nx * g.ny, rbasis.p, stream);
        launch_refined_startup(g, b, n, stream);
        HIP_TRY(this, hipEventRecord(events[2], stream));
        launch_fim(prob_r.p, n, launch_shape(kRefMax, kRefMax), stream);
        HIP_TRY(this, hipEventRecord(events[3], stream));
        launch_handoff(g, b, n, stream);
        launch_coarse_march(g, b, n, slow.p, nrec_c, risti_c.p, stream);
        HIP_TRY(this, hipEventRecord(events[4], stream));
        launch_fim(prob_c.p, n, launch_shape(g.nnx, g.nnz), stream);
        HIP_TRY(this, hipEventRecord(events[5], stream));
        // receivers of this chunk
        const int r0 = h_src[first].first_ray;
        const int r1 = h_src[first + n - 1].first_ray + h_src[first + n - 1].nrec;
        if (r1 > r0) launch_srtimes_chunk(r0, r1 - r0, first);
        HIP_TRY(this, hipEventRecord(events[6], stream));
        HIP_TRY(this, hipGetLastError());
        h_info.resize((size_t)n * 16);
        h_flags.resize((size_t)n * 4);
        HIP_TRY(this, hipMemcpyAsync(h_info.data(), info.p, (size_t)n * 16 * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(this, hipMemcpyAsync(h_flags.data(), flags.p, (size_t)n * 4 * 4, hipMemcpyDeviceToHost, stream));
        if (dsurf && r1 > r0)
            HIP_TRY(this, hipMemcpyAsync(dsurf + r0, out.p + r0, (size_t)(r1 - r0) * 4, hipMemcpyDeviceToHost, stream));
        std::vector<unsigned long long> h_clk((size_t)n * 8);
        HIP_TRY(this, hipMemcpyAsync(h_clk.data(), clocks.p, (size_t)n * 64, hipMemcpyDeviceToHost, stream));
        HIP_TRY(this, hipStreamSynchronize(stream));
        for (int u = 0; u < n; ++u) {
            for (int q = 0; q < 6; ++q) phase_ticks[q] += (double)h_clk[(size_t)u * 8 + q];
            phase_ticks[6] = std::max(phase_ticks[6], (double)h_clk[(size_t)u * 8 + 6]);
        }
        float ms = 0;
        HIP_TRY(this, hipEventElapsedTime(&ms, events[2], events[3])); stats[DSA_STAT_MS_FIM_REFINED] += ms;
        HIP_TRY(this, hipEventElapsedTime(&ms, events[4], events[5])); stats[DSA_STAT_MS_FIM_COARSE] += ms;
        float a = 0, c2 = 0, d = 0;
        HIP_TRY(this, hipEventElapsedTime(&a, events[1], events[2]));
        HIP_TRY(this, hipEventElapsedTime(&c2, events[3], events[4]));
        HIP_TRY(this, hipEventElapsedTime(&d, events[5], events[6]));
        stats[DSA_STAT_MS_STAGES] += a + c2 + d;
        stats[DSA_STAT_LAUNCHES_FIM_COARSE] += 1;
        last_chunk_first = first;
        last_chunk_n = n;
        for (int u = 0; u < n; ++u) {
            const int32_t* fi = &h_info[(size_t)u * 16];
            stats[DSA_STAT_ROUNDS_MAX] = std::max(stats[DSA_STAT_ROUNDS_MAX], (double)fi[8]);
            unsigned long long ev;
            std::memcpy(&ev, fi + 12, 8);
            stats[DSA_STAT_EVALS_TOTAL] += (double)ev;
            stats[DSA_STAT_RESCANS] += fi[1] + fi[9];
            stats[DSA_STAT_FREEZES] += fi[3] + fi[11];
            if (fi[2] < 0 || fi[10] < 0) { fail(DSA_ERR_INTERNAL, "unit %d: fixed-point solve did not converge (rounds %d/%d)", first + u, fi[0], fi[8]); return DSA_ERR_INTERNAL; }
            if (h_flags[(size_t)u * 4 + 1]) { fail(DSA_ERR_INTERNAL, "unit %d: serial march guard %d (1/17 window, 2/18 tree)", first + u, h_flags[(size_t)u * 4 + 1]); return DSA_ERR_INTERNAL; }
        }
        last_chunk_first = first;
        last_chunk_n = n;
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

FimLaunch Engine::launch_shape(int nnx, int nnz) const
{
    // the active band is a few node layers along the front's perimeter (peak ~13 k nodes at 1025^2)
    FimLaunch l;
    l.list_cap = list_cap > 0 ? list_cap : 16 * (nnx + nnz) + 4096;
    l.ready_cap = ready_cap > 0 ? ready_cap : 8 * (nnx + nnz) + 2048;
    l.threads = fim_threads;
    return l;
}

void Engine::launch_srtimes_chunk(int r0, int nr, int first_unit)
{
    launch_srtimes(g, batch(), first_unit, rays.p + r0, nr, veln.p, nfield, dpl, out.p + r0, err.p, stream);
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

int Engine::get_field(int unit, float* ttn)
{
    if (last_chunk_first < 0 || unit < last_chunk_first || unit >= last_chunk_first + last_chunk_n) { fail(DSA_ERR_STATE, "get_field: unit %d is not resident (last chunk covers %d..%d)", unit, last_chunk_first, last_chunk_first + last_chunk_n - 1); return DSA_ERR_STATE; }
    HIP_TRY(this, hipSetDevice(device));
    if (fetch_tiled(F_c.p + (size_t)(unit - last_chunk_first) * nrec_c, g.nnx, g.nnz, 0, ttn)) return status;
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
    if (n == "list_cap" && value >= 0) { en->list_cap = (int)value; return 0; }
    if (n == "ready_cap" && value >= 0) { en->ready_cap = (int)value; return 0; }
    if (n == "fim_threads" && (value == 256 || value == 512 || value == 1024)) { en->fim_threads = (int)value; return 0; }
    en->fail(DSA_ERR_ARGUMENT, "unknown option or bad value: %s=%g", name, value);
    return DSA_ERR_ARGUMENT;
}

int dsa_set_maps(dsa_engine* e, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int dicing, int nmaps, const double* pv)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->set_maps(nx, ny, goxd, gozd, dvxd, dvzd, dicing, nmaps, pv);
}

int dsa_plan(dsa_engine* e, int nunits, const int* map_index, const float* scx, const float* scz, const int* nrec, const float* rcx, const float* rcz)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->plan(nunits, map_index, scx, scz, nrec, rcx, rcz);
}

int dsa_solve(dsa_engine* e, float* dsurf)
{
    if (!e) return DSA_ERR_ARGUMENT;
    return reinterpret_cast<Engine*>(e)->solve(dsurf);
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
    // with keep_fields the chunk is capped so that every planned unit stays resident
    reinterpret_cast<Engine*>(e)->keep_fields = on != 0;
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
    const size_t slot = (size_t)(unit - en->last_chunk_first);
    if (which < 2) return en->fetch_tiled(en->F_c.p + slot * en->nrec_c, en->g.nnx, en->g.nnz, which, out);
    const dsa::SourceDesc& s = en->h_src[unit];
    return en->fetch_tiled(en->F_r.p + slot * dsa::kRefRecs, s.rnx, s.rnz, which - 2, out);
}

int dsa_get_stats(const dsa_engine* e, double* out)
{
    if (!e || !out) return DSA_ERR_ARGUMENT;
    std::memcpy(out, reinterpret_cast<const Engine*>(e)->stats, sizeof(double) * DSA_STAT_COUNT);
    // phase clocks of the coarse solve, summed over units (100 MHz wall clock ticks): pass A, even half,
    // odd half, round end; then the summed list lengths and ready counts
    for (int q = 0; q < 7; ++q) out[DSA_STAT_COUNT + q] = reinterpret_cast<const Engine*>(e)->phase_ticks[q];
    return 0;
}

}  // extern "C"
