// Host side of one outer iteration around the device calls: what the reference's main program
// does between CalSurfG and LSMR (main.f90:361-466) and after LSMR (main.f90:520-535).  Plain host code with the
// reference's fp32 arithmetic, so that an iteration driven through this library (dsurftomo_amd/invert.py) produces the
// reference's numbers; O(nar) work, no device involved.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/dsurftomo_amd.h"

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
    for (int i = 0; i < dall; ++i) cbst[i] = obst[i] - dsyn[i];                                // :361-363
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
    for (int i = 0; i < dall; ++i) {                                                           // :366-372
        const bool out = cbst[i] < lo || cbst[i] > hi;
        datweight[i] = out ? 0.0f : 1.0f;
        if (out) cbst[i] = 0.0f;
    }
    std::fill(norm, norm + maxvp, 0.0f);
    for (long long k = 0; k < nar_in; ++k) {                                                   // :378-385
        rw[k] = rw[k] * datweight[iw[1 + k] - 1];
        norm[col[k] - 1] = norm[col[k] - 1] + std::fabs(rw[k]);
    }
    float total = 0.0f, top = 0.0f;                                                            // :386-392
    for (long long i = 0; i < maxvp; ++i) { total = total + norm[i]; if (norm[i] > top) top = norm[i]; }
    dws[0] = top; dws[1] = total / (float)maxvp;

    // regularisation rows, one per model parameter in (k, j, i) order (:420-457)
    long long interior = 0;
    if (nvx > 2 && nvz > 2 && nl > 2) interior = (long long)(nvx - 2) * (nvz - 2) * (nl - 2);
    const long long nar_out_ = nar_in + 7 * interior + (maxvp - interior);
    if (nar_out_ > capacity) return DSA_ERR_CAPACITY;                                          // the reference: stop 'increase sparsity fraction'
    long long nar = nar_in;
    int row = dall;
    const int plane = nvz * nvx;
    for (int k = 1; k <= nl; ++k)
        for (int j = 1; j <= nvz; ++j)
            for (int i = 1; i <= nvx; ++i) {
                ++row;
                cbst[row - 1] = 0.0f;
                const int here = (k - 1) * plane + (j - 1) * nvx + i;
                const bool face = i == 1 || i == nvx || j == 1 || j == nvz || k == 1 || k == nl;
                if (face) {
                    col[nar] = here; rw[nar] = 2.0f * weight0; iw[1 + nar] = row;
                    nar += 1;
                } else {
                    const int nb[7] = { here, here - 1, here + 1, here - nvx, here + nvx, here - plane, here + plane };
                    for (int q = 0; q < 7; ++q) {
                        col[nar + q] = nb[q];
                        rw[nar + q] = q == 0 ? 6.0f * weight0 : -1.0f * weight0;
                        iw[1 + nar + q] = row;
                    }
                    nar += 7;
                }
            }
    *m_out = row;
    iw[0] = (int)nar;                                                                          // :461-464
    for (long long k = 0; k < nar; ++k) iw[1 + nar + k] = col[k];
    *nar_out = nar;
    return 0;
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
