// Host-side geometry of a CalSurfG call: grid set-up and per-source descriptors.
// Plain fp32 arithmetic in the reference's order (CalSurfG.f90:1044-1065, :1209-1246); the only
// transcendental is libm sinf for the per-column `risti` tables, which is why they are made here
// and uploaded instead of being computed by device code.
#pragma once

#include <math.h>

#include "dispersion_core.h"
#include "source_stage.h"

namespace dsa {

constexpr float kPi = 3.1415926535898f;   // reference globalp::pi, rounded to fp32

inline void make_grid(GridDesc& g, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd)
{
    g.nx = nx; g.ny = ny; g.nvx = nx - 2; g.nvz = ny - 2; g.gdx = gd; g.gdz = gd;
    g.earth = 6371.0f;
    g.dvx = dvxd * kPi / 180.0f;
    g.dvz = dvzd * kPi / 180.0f;
    g.gox = (90.0f - goxd) * kPi / 180.0f;
    g.goz = gozd * kPi / 180.0f;
    g.nnx = (g.nvx - 1) * gd + 1;
    g.nnz = (g.nvz - 1) * gd + 1;
    g.dnx = g.dvx / (float)gd;
    g.dnz = g.dvz / (float)gd;
    g.nbx = (g.nnx + 7) / 8;
    g.nbz = (g.nnz + 7) / 8;
}

// (n+1) x 4 cubic B-spline basis values at u = i/n
inline void basis_table(int n, float* out)
{
    for (int i = 1; i <= n + 1; ++i) {
        float u = (float)n;
        u = (float)(i - 1) / u;
        bspline4(u, out + 4 * (i - 1));
    }
}

inline void risti_table(float gox, float dnx, float earth, int n, float* out)
{
    for (int ix = 1; ix <= n; ++ix) out[ix - 1] = earth * sinf(gox + (float)(ix - 1) * dnx);
}

// minimum cell width in km used by the receiver / ray routines (reference :1705-1709, :1864-1868)
inline float min_cell_km(const GridDesc& g)
{
    float dpl = g.dnx * g.earth;
    float rd1 = g.dnz * g.earth * sinf(g.gox);
    if (rd1 < dpl) dpl = rd1;
    rd1 = g.dnz * g.earth * sinf(g.gox + (float)(g.nnx - 1) * g.dnx);
    if (rd1 < dpl) dpl = rd1;
    return dpl;
}

// returns 0, or -1 when the source lies outside the grid (the reference STOPs there)
inline int make_source(const GridDesc& g, float x, float z, SourceDesc& s)
{
    s.scx = x; s.scz = z;
    int isx = (int)((x - g.gox) / g.dnx) + 1;
    int isz = (int)((z - g.goz) / g.dnz) + 1;
    if (isx < 1 || isx > g.nnx || isz < 1 || isz > g.nnz) return -1;
    if (isx == g.nnx) isx -= 1;
    if (isz == g.nnz) isz -= 1;
    s.vnl = isx - kSgs; if (s.vnl < 1) s.vnl = 1;
    s.vnr = isx + kSgs; if (s.vnr > g.nnx) s.vnr = g.nnx;
    s.vnt = isz - kSgs; if (s.vnt < 1) s.vnt = 1;
    s.vnb = isz + kSgs; if (s.vnb > g.nnz) s.vnb = g.nnz;
    s.rnx = (s.vnr - s.vnl) * kSgdl + 1;
    s.rnz = (s.vnb - s.vnt) * kSgdl + 1;
    s.rdnx = g.dvx / (float)(g.gdx * kSgdl);
    s.rdnz = g.dvz / (float)(g.gdz * kSgdl);
    s.rgox = g.gox + g.dnx * (float)(s.vnl - 1);
    s.rgoz = g.goz + g.dnz * (float)(s.vnt - 1);
    // source cell in the refined grid (travel, :312-325 with the refined geometry)
    int rx = (int)((x - s.rgox) / s.rdnx) + 1;
    int rz = (int)((z - s.rgoz) / s.rdnz) + 1;
    if (rx < 1 || rx > s.rnx || rz < 1 || rz > s.rnz) return -1;
    if (rx == s.rnx) rx -= 1;
    if (rz == s.rnz) rz -= 1;
    s.isx_r = rx; s.isz_r = rz;
    s.dsx_r = (x - s.rgox) - (float)(rx - 1) * s.rdnx;
    s.dsz_r = (z - s.rgoz) - (float)(rz - 1) * s.rdnz;
    // literal edge test of the refined stage: coarse bounds against refined extents
    s.open_xlo = (s.vnl != 1);
    s.open_xhi = (s.vnr != s.rnx);
    s.open_zlo = (s.vnt != 1);
    s.open_zhi = (s.vnb != s.rnz);
    s.rwx0 = rx - kRWin / 2;
    s.rwz0 = rz - kRWin / 2;
    s.cwx0 = s.vnl - 1 - kCMargin; if (s.cwx0 < 0) s.cwx0 = 0;
    s.cwz0 = s.vnt - 1 - kCMargin; if (s.cwz0 < 0) s.cwz0 = 0;
    int ex = s.vnr + kCMargin; if (ex > g.nnx) ex = g.nnx;
    int ez = s.vnb + kCMargin; if (ez > g.nnz) ez = g.nnz;
    s.cwnx = ex - s.cwx0;
    s.cwnz = ez - s.cwz0;
    s.nbx_r = (s.rnx + 7) / 8;
    s.nbz_r = (s.rnz + 7) / 8;
    s.period = 0; s.first_ray = 0; s.nrec = 0; s.sen_slot = 0;
    return 0;
}

// Velocity-independent layer geometry of a call: sublayer counts and thicknesses of
// refineGrid2LayerMdl (CalSurfG.f90:2352-2411) pushed through the earth-flattening transform of
// `sphere` (surfdisp96.f:480-547) with libm log / powf.  Returns 0, or -1 when the refined model
// exceeds the reference's NL = 200 layers.
inline int make_layer_geom(int nz, const float* depz, float minthk0, LayerGeom& G)
{
    G.nz = nz;
    float thk[kMaxLayers];
    int k = 0;
    for (int i = 1; i <= nz - 1; ++i) {
        const float t = depz[i] - depz[i - 1];
        const float minthk = t / minthk0;
        const int nsub = (int)((t + 1.0e-4f) / minthk) + 1;
        if (nsub < 1 || k + nsub + 1 > kMaxLayers) return -1;
        G.nsub[i - 1] = nsub;
        const float newthk = t / (float)nsub;
        for (int j = 0; j < nsub; ++j) thk[k++] = newthk;
    }
    thk[k++] = 1.0f;              // sphere() gives the half space unit thickness before flattening
    G.rmax = k;
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    for (int i = 0; i < k; ++i) {
        dr = dr + (double)thk[i];
        const double r1 = ar - dr;
        const double z0 = ar * log(ar / r0);
        const double z1 = ar * log(ar / r1);
        G.dflat[i] = (float)(z1 - z0);
        const double tmp = (ar + ar) / (r0 + r1);
        G.tmp[i] = tmp;
        const float x = (float)tmp, x2 = x * x;
        G.rhofac_love[i] = 1.0f / (x * (x2 * x2));        // btp**(-5): binary powering, then the reciprocal
        G.rhofac_rayl[i] = powf(x, -2.275f);
        r0 = r1;
    }
    return 0;
}

}  // namespace dsa
