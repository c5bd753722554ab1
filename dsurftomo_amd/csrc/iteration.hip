// Host side of one outer iteration around the device calls: what the reference's main program
// does between CalSurfG and LSMR (main.f90:361-466) and after LSMR (main.f90:520-535).  Plain host code with the
// reference's fp32 arithmetic, so that an iteration driven through this library (dsurftomo_amd/invert.py) produces the
// reference's numbers; O(nar) work, no device involved.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/dsurftomo_amd.h"
#include "engine.h"
#include "spmv_state.h"

namespace {

// cbst = obst - dsyn (main.f90:361-363); data outside [q25, q75] * threshold0 get weight 0 and residual 0 (:366-372);
// q25 / q75: elements int(0.25 N) and int(0.75 N) of the sorted residuals (getpercentile.f90:27-30)
int residual_weights(int dall, const float* obst, const float* dsyn, float threshold0, float* cbst, float* datweight)
{
    for (int i = 0; i < dall; ++i) cbst[i] = obst[i] - dsyn[i];
    float q25, q75;
    {
        std::vector<float> ra(cbst, cbst + dall);
        const int i25 = (int)(0.25f * (float)dall), i75 = (int)(0.75f * (float)dall);          // 1-based ranks
        if (i25 < 1 || i75 < 1) return DSA_ERR_ARGUMENT;
        std::nth_element(ra.begin(), ra.begin() + (i75 - 1), ra.end());
        q75 = ra[i75 - 1];
        if (i25 < i75) std::nth_element(ra.begin(), ra.begin() + (i25 - 1), ra.begin() + (i75 - 1));
        q25 = ra[i25 - 1];
    }
    const float lo = q25 * threshold0, hi = q75 * threshold0;
    for (int i = 0; i < dall; ++i) {
        const bool out = cbst[i] < lo || cbst[i] > hi;
        datweight[i] = out ? 0.0f : 1.0f;
        if (out) cbst[i] = 0.0f;
    }
    return 0;
}

long long regularisation_entries(int nvx, int nvz, int nl)
{
    const long long maxvp = (long long)nvx * nvz * nl;
    long long interior = 0;
    if (nvx > 2 && nvz > 2 && nl > 2) interior = (long long)(nvx - 2) * (nvz - 2) * (nl - 2);
    return 7 * interior + (maxvp - interior);
}

// first-difference Laplacian rows, one per model parameter in (k, j, i) order: 2 w on the model's faces, 6 w and six -w
// inside (main.f90:420-457); row numbers continue behind the dall data rows; returns the entries written
long long regularisation_rows(int nvx, int nvz, int nl, float weight0, int dall, float* rw, int* row_out, int* col)
{
    long long nar = 0;
    int row = dall;
    const int plane = nvz * nvx;
    for (int k = 1; k <= nl; ++k)
        for (int j = 1; j <= nvz; ++j)
            for (int i = 1; i <= nvx; ++i) {
                ++row;
                const int here = (k - 1) * plane + (j - 1) * nvx + i;
                const bool face = i == 1 || i == nvx || j == 1 || j == nvz || k == 1 || k == nl;
                if (face) {
                    col[nar] = here; rw[nar] = 2.0f * weight0; row_out[nar] = row;
                    nar += 1;
                } else {
                    const int nb[7] = { here, here - 1, here + 1, here - nvx, here + nvx, here - plane, here + plane };
                    for (int q = 0; q < 7; ++q) {
                        col[nar + q] = nb[q];
                        rw[nar + q] = q == 0 ? 6.0f * weight0 : -1.0f * weight0;
                        row_out[nar + q] = row;
                    }
                    nar += 7;
                }
            }
    return nar;
}

// rw[k] *= w[row[k] - 1] (main.f90:379)
__global__ void k_scale_rows(long long nar, float* __restrict__ rw, const int* __restrict__ row, const float* __restrict__ w)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nar) return;
    rw[k] = rw[k] * w[row[k] - 1];
}

}  // namespace

namespace dsa {
int spmv_load_from_device(Engine* e, int m, int n, long long nar, const float* d_rw, const int* d_row, const int* d_col, long long nar_data);
void spmv_abs_column_sums(Engine* e, float* d_out);
}

extern "C" {

// cbst = obst - dsyn; data outside [q25, q75] * threshold0 get weight 0 (q25 / q75: elements int(0.25 N) and int(0.75 N)
// of the sorted residuals, getpercentile.f90:27-30); rows scaled by their weights; DWS = column sums of |G|;
// first-difference Laplacian rows appended below the data rows (2 w on the model's faces, 6 w and six -w inside).
int dsa_iteration_system(int nx, int ny, int nz, int dall, long long nar_in, long long capacity, float* rw, int* iw, int* col,
                         const float* obst, const float* dsyn, float threshold0, float weight0, float* cbst,
                         float* datweight, float* norm, int* m_out, long long* nar_out, float* dws)
{
    if (nx < 3 || ny < 3 || nz < 2 || dall < 1 || nar_in < 0 || !rw || !iw || !col || !obst || !dsyn || !cbst || !datweight || !norm ||
        !m_out || !nar_out || !dws) return DSA_ERR_ARGUMENT;
    const int nvx = nx - 2, nvz = ny - 2, nl = nz - 1;
    const long long maxvp = (long long)nvx * nvz * nl;
    { const int rc = residual_weights(dall, obst, dsyn, threshold0, cbst, datweight); if (rc != 0) return rc; }   // :361-372
    std::fill(norm, norm + maxvp, 0.0f);
    for (long long k = 0; k < nar_in; ++k) {                                                   // :378-385
        rw[k] = rw[k] * datweight[iw[1 + k] - 1];
        norm[col[k] - 1] = norm[col[k] - 1] + std::fabs(rw[k]);
    }
    float total = 0.0f, top = 0.0f;                                                            // :386-392
    for (long long i = 0; i < maxvp; ++i) { total = total + norm[i]; if (norm[i] > top) top = norm[i]; }
    dws[0] = top; dws[1] = total / (float)maxvp;

    // regularisation rows, one per model parameter in (k, j, i) order (:420-457)
    const long long nar_out_ = nar_in + regularisation_entries(nvx, nvz, nl);
    if (nar_out_ > capacity) return DSA_ERR_CAPACITY;                                          // the reference: stop 'increase sparsity fraction'
    for (long long i = 0; i < maxvp; ++i) cbst[dall + i] = 0.0f;
    const long long nar = nar_in + regularisation_rows(nvx, nvz, nl, weight0, dall, rw + nar_in, iw + 1 + nar_in, col + nar_in);
    const int row = dall + (int)maxvp;
    *m_out = row;
    iw[0] = (int)nar;                                                                          // :461-464
    for (long long k = 0; k < nar; ++k) iw[1 + nar + k] = col[k];
    *nar_out = nar;
    return 0;
}

// The same step on the rows that dsa_calsurfg / dsa_solve_rows left on the device (option rows_on_device, or dsa_calsurfg
// called with null rw / iw / col): the weights are applied, the regularisation rows appended and both orderings of the
// matrix built where the rows are, so the 12 bytes per entry never cross PCIe (main.f90:349-359 hands them to the host,
// :361-466 rebuilds them there, :487-489 gives them to LSMR).  Afterwards dsa_lsmr solves on that matrix.  Same bits as
// dsa_iteration_system followed by dsa_spmv_load: the scaling is one fp32 multiply per entry, the DWS column sums add
// |entry| in storage order.
int dsa_iteration_system_device(dsa_engine* h, int nx, int ny, int nz, int dall, const float* obst, const float* dsyn, float threshold0,
                                float weight0, float* cbst, float* datweight, float* norm, int* m_out, long long* nar_out, float* dws)
{
    if (!h) return DSA_ERR_ARGUMENT;
    dsa::Engine* e = reinterpret_cast<dsa::Engine*>(h);
    if (nx < 3 || ny < 3 || nz < 2 || dall < 1 || !obst || !dsyn || !cbst || !datweight || !norm || !m_out || !nar_out || !dws) { e->fail(DSA_ERR_ARGUMENT, "iteration_system_device: bad arguments"); return DSA_ERR_ARGUMENT; }
    if (!e->rows_on_device || (e->G_nar > 0 && !e->G_rw.p)) { e->fail(DSA_ERR_STATE, "iteration_system_device: no rows on the device (option rows_on_device + dsa_solve_rows, or dsa_calsurfg with null arrays)"); return DSA_ERR_STATE; }
    const int nvx = nx - 2, nvz = ny - 2, nl = nz - 1;
    const long long maxvp = (long long)nvx * nvz * nl, nar_in = e->G_nar;
    { const int rc = residual_weights(dall, obst, dsyn, threshold0, cbst, datweight); if (rc != 0) { e->fail(rc, "iteration_system_device: too few data"); return rc; } }
    for (long long i = 0; i < maxvp; ++i) cbst[dall + i] = 0.0f;
    if (hipSetDevice(e->device) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "iteration_system_device: hipSetDevice"); return DSA_ERR_DEVICE; }
    const long long nreg = regularisation_entries(nvx, nvz, nl), nar = nar_in + nreg;
    if (nar > 0x7fffffffll) { e->fail(DSA_ERR_CAPACITY, "iteration_system_device: more than 2^31-1 matrix entries"); return DSA_ERR_CAPACITY; }
    std::vector<float> hrw((size_t)nreg);
    std::vector<int> hrow((size_t)nreg), hcol((size_t)nreg);
    regularisation_rows(nvx, nvz, nl, weight0, dall, hrw.data(), hrow.data(), hcol.data());
    dsa::DevBuf<float> d_w, d_norm;
    auto rel = [](auto& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; };
    auto done = [&](int rc) { rel(d_w); rel(d_norm); return rc; };
#define IT_TRY(call) do { if ((call) != hipSuccess) { e->fail(DSA_ERR_DEVICE, "iteration_system_device: %s failed", #call); return done(DSA_ERR_DEVICE); } } while (0)
    if (e->ensure(d_w, (size_t)dall) || e->ensure(d_norm, (size_t)maxvp) ||
        e->ensure_keep(e->G_rw, (size_t)nar, (size_t)nar_in) || e->ensure_keep(e->G_row, (size_t)nar, (size_t)nar_in) || e->ensure_keep(e->G_col, (size_t)nar, (size_t)nar_in)) return done(e->status);
    IT_TRY(hipMemcpyAsync(d_w.p, datweight, (size_t)dall * 4, hipMemcpyHostToDevice, e->stream));
    IT_TRY(hipMemsetAsync(d_norm.p, 0, (size_t)maxvp * 4, e->stream));
    if (nar_in > 0)
        hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)((nar_in + 255) / 256)), dim3(256), 0, e->stream, nar_in, e->G_rw.p, e->G_row.p, d_w.p);
    IT_TRY(hipMemcpyAsync(e->G_rw.p + nar_in, hrw.data(), (size_t)nreg * 4, hipMemcpyHostToDevice, e->stream));
    IT_TRY(hipMemcpyAsync(e->G_row.p + nar_in, hrow.data(), (size_t)nreg * 4, hipMemcpyHostToDevice, e->stream));
    IT_TRY(hipMemcpyAsync(e->G_col.p + nar_in, hcol.data(), (size_t)nreg * 4, hipMemcpyHostToDevice, e->stream));
    IT_TRY(hipStreamSynchronize(e->stream));
    const int m = dall + (int)maxvp;
    { const int rc = dsa::spmv_load_from_device(e, m, (int)maxvp, nar, e->G_rw.p, e->G_row.p, e->G_col.p, nar_in); if (rc != 0) return done(rc); }
    // DWS (main.f90:378-392): per column the sum of |entry| over the DATA entries, which lead every column in storage order
    dsa::spmv_abs_column_sums(e, d_norm.p);
    IT_TRY(hipMemcpyAsync(norm, d_norm.p, (size_t)maxvp * 4, hipMemcpyDeviceToHost, e->stream));
    IT_TRY(hipStreamSynchronize(e->stream));
    IT_TRY(hipGetLastError());
#undef IT_TRY
    float total = 0.0f, top = 0.0f;                                                            // :386-392
    for (long long i = 0; i < maxvp; ++i) { total = total + norm[i]; if (norm[i] > top) top = norm[i]; }
    dws[0] = top; dws[1] = total / (float)maxvp;
    e->G_nar = nar;
    *m_out = m;
    *nar_out = nar;
    return done(0);
}

// main.f90:520-535: the update is clipped to +-0.5 km/s, the model to [minvel, maxvel]; vsf(nx, ny, nz) column-major
int dsa_model_update(int nx, int ny, int nz, float* dv, float* vsf, float minvel, float maxvel)
{
    if (nx < 3 || ny < 3 || nz < 2 || !dv || !vsf) return DSA_ERR_ARGUMENT;
    const int nvx = nx - 2, nvz = ny - 2;
    for (int k = 0; k < nz - 1; ++k)
        for (int j = 0; j < nvz; ++j)
            for (int i = 0; i < nvx; ++i) {
                float& d = dv[((size_t)k * nvz + j) * nvx + i];
                if (d >= 0.500f) d = 0.500f;
                if (d <= -0.500f) d = -0.500f;
                float& v = vsf[((size_t)k * ny + (j + 1)) * nx + (i + 1)];
                v = v + d;
                if (v < minvel) v = minvel;
                if (v > maxvel) v = maxvel;
            }
    return 0;
}

}  // extern "C"
