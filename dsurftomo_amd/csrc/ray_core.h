// Ray back-trace from a receiver to the source and the per-vertex kernel it leaves behind
// (reference `rpaths`, CalSurfG.f90:1771-2318), plus the Brocher chain-rule coefficients of the
// Frechet row assembly (CalSurfG.f90:1383-1423).
//
// Written once for device code; compiles for the host so that tests/hostcheck.cpp can run the same
// arithmetic on a CPU next to the oracle.  fp32, one rounding per operation, no FMA contraction.
#pragma once

#include "source_stage.h"

namespace dsa {

// sin(x) for fp32 x, rounded from an fp64 polynomial: libm's sinf (glibc >= 2.28: reduction by
// multiples of pi/2 in fp64, degree-7 / degree-8 minimax polynomials in fp64, one rounding to fp32).
// The reference calls sinf along every ray step; the device's own sinf differs from libm's in the
// last bit, so the ray tracer carries this one.  Checked bitwise against libm over every fp32 in
// [1e-3, 3.2] by tests/test_hostcheck.py; valid for |x| < 120.
DSA_HD float sinf_libm(float y)
{
    double x = (double)y;
    int n = 0;
    double sgn = 1.0;
    if (fabsf(y) >= 0x1.921fb6p-1f) {                       // pi/4
        const double r = x * 0x1.45F306DC9C883p+23;         // 2/pi * 2^24
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * 0x1.921FB54442D18p0;            // pi/2
        if (n & 2) sgn = -1.0;
    }
    const double x2 = x * x;
    if ((n & 1) == 0) {
        const double xs = x * sgn;
        const double x3 = xs * x2, s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13, x7 = x3 * x2;
        const double s = xs + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2, c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16;
    const double c1 = 1.0 + x2 * -0x1.ffffffd0c621cp-2, x6 = x4 * x2, c = c1 + x4 * 0x1.55553e1068f19p-5;
    return (float)(sgn * (c + x6 * c2));
}

// fp32 division by a denominator that stays the same for the whole ray (round 5).  The compiler expands `x / d` into v_div_scale (x2), v_rcp, one
// Newton step on the reciprocal, the quotient refined twice, v_div_fmas and v_div_fixup -- eleven instructions, and a step of a ray divides
// some fifty times by the cell sizes of the three grids.  recipf_of / divf_by are that expansion split at the denominator: its reciprocal once
// per ray, six instructions per quotient; the same instructions on the same operands, minus the rescaling of v_div_scale / v_div_fmas, which
// acts only when the denominator is denormal or beyond 2^126, the exponents lie 96 apart, the quotient is denormal or the numerator lies
// below 2^-104.  It is used only where that cannot happen: denominators that are cell sizes in radians (1e-6 .. 1) or twice a cell size in
// km, numerators that are differences of coordinates in radians or of travel times in seconds -- zero, or no smaller than an ulp of such a
// number (> 1e-12); zeros and their signs come out of v_div_fixup as the full division has them.  The host build divides.
struct RecipF { float d, r; };
DSA_HD RecipF recipf_of(float d)
{
    RecipF R;
    R.d = d;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSA_RAY_PLAIN_DIV)
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    R.r = __builtin_fmaf(e, r, r);
#else
    R.r = 0.0f;
#endif
    return R;
}
DSA_HD float divf_by(float x, const RecipF& R)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSA_RAY_PLAIN_DIV)
    float q = x * R.r;
    float e = __builtin_fmaf(-R.d, q, x);
    q = __builtin_fmaf(e, R.r, q);
    e = __builtin_fmaf(-R.d, q, x);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, R.r, q), R.d, x);
#else
    return x / R.d;
#endif
}

// bspline4 (eikonal_core.h) with its four divisions by 6 through a shared reciprocal: u = a coordinate difference over a cell size, so u, 1 - u
// and their cubes are zero or no smaller than 1e-22
DSA_HD void bspline4_by(float u, float w[4], const RecipF& by_6)
{
    const float u2 = u * u, u3 = u * (u * u);
    const float m = 1.0f - u;
    w[0] = divf_by(m * (m * m), by_6);
    w[1] = divf_by(4.0f - 6.0f * u2 + 3.0f * u3, by_6);
    w[2] = divf_by(1.0f + 3.0f * u + 3.0f * u2 - 3.0f * u3, by_6);
    w[3] = divf_by(u3, by_6);
}

// four B-spline weights, each zero or at least 2^-52 (see trace_ray's vertex sums); the host build, which divides, does not ask
DSA_HD bool weights_plain(const float w[4])
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSA_RAY_PLAIN_DIV)
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) ok = ok && (w[q] == 0.0f || fabsf(w[q]) >= 0x1p-52f);
    return ok;
#else
    return false;
#endif
}

// What a ray reads: the coarse field (tiled records), the diced velocity (x-major rows) and the
// refined snapshot of its source (x-major, leading dimension rnz; status 0 = alive at hand-off)
struct RayFields {
    const float* F;       // compact coarse field (one float per node, sign bit = exceptional node: eikonal_core.h)
    const float* veln;
    const float* Tr;
    const int8_t* Sr;
};

// The 4x4 patch of B-spline vertices a ray segment contributes to moves only when the ray crosses
// a vertex cell, so its 16 running sums live in registers and the per-ray slab in HBM is touched
// once per crossing.  Sums stay sequential per vertex (load, add, add, ..., store), which keeps the
// fp32 result identical to the reference's read-modify-write per sub-segment.
// Slab layout: (nvz+2) rows of (nvx+2), x fastest: slab[vz * (nvx+2) + vx], vertex indices 0-based
// including the border ring (reference fdm(0:nvz+1, 0:nvx+1)).
// LPR (lanes per ray) = 4, round 5: four neighbouring lanes trace the same ray -- the same steps, every lane for itself -- and split the patch by
// SLAB row: lane `sub` keeps the one patch row whose slab row is congruent to sub modulo 4 (patch row (sub - pz) & 3), so a slab row has one
// owner for the whole ray and the lanes never read what another lane wrote.  Every vertex sum sees the same operations in the same order as
// with one lane per ray: the same bits.
template <int LPR = 1>
struct PatchAcc {
    float* slab;
    int ldx;
    int px, pz;        // vertex index of element (m=1, l=1); -1 = nothing loaded
    int sub;
    float a[LPR == 1 ? 4 : 1][4];     // [l-1][m-1]; LPR = 4: [0][m-1] of my row

    DSA_HDM void init(float* s, int ld, int sub_) { slab = s; ldx = ld; px = -1; pz = -1; sub = sub_; }
    DSA_HDM int my_row() const { return (sub - pz) & 3; }
    DSA_HDM void store()
    {
        if (px < 0) return;
        if (LPR == 1) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int m = 0; m < 4; ++m) slab[(size_t)(pz + l) * ldx + (px + m)] = a[l][m];
        } else {
            const int l = my_row();
#pragma unroll
            for (int m = 0; m < 4; ++m) slab[(size_t)(pz + l) * ldx + (px + m)] = a[0][m];
        }
    }
    DSA_HDM void move(int nx0, int nz0)
    {
        if (nx0 == px && nz0 == pz) return;
        store();
        px = nx0; pz = nz0;
        if (LPR == 1) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int m = 0; m < 4; ++m) a[l][m] = slab[(size_t)(pz + l) * ldx + (px + m)];
        } else {
            const int l = my_row();
#pragma unroll
            for (int m = 0; m < 4; ++m) a[0][m] = slab[(size_t)(pz + l) * ldx + (px + m)];
        }
    }
};

DSA_HD bool refined_cell_alive(const SourceDesc& s, const int8_t* Sr, int ipxr, int ipzr)
{
    if (ipxr < 1 || ipxr >= s.rnx) return false;
    if (ipzr < 1 || ipzr >= s.rnz) return false;
    const size_t o = (size_t)(ipxr - 1) * s.rnz + (ipzr - 1);
    return Sr[o] == 0 && Sr[o + 1] == 0 && Sr[o + s.rnz] == 0 && Sr[o + s.rnz + 1] == 0;
}

// Optional record of the ray's points (the reference's rgx / rgz arrays, CalSurfG.f90:1910-1911, :2043-2076: the receiver,
// the end point of every step after clamping, finally the source), which its disabled dump would write to raypath.out
// (:2276-2283): up to `cap` (colatitude, longitude) pairs in radians; n counts all points of the ray.
struct RayPath {
    float* pts = nullptr;
    int cap = 0, n = 0;
    DSA_HDM void push(float x, float z) { if (pts && n < cap) { pts[2 * n] = x; pts[2 * n + 1] = z; } n += 1; }
};

// returns 0, or -1 when the receiver lies outside the grid; *flags bit 0 = the ray was clamped at
// the model edge (reference rbint), *nsteps = gradient steps taken
template <int LPR = 1>
DSA_HD int trace_ray(const GridDesc& g, const SourceDesc& s, const RayFields& f, float rcx, float rcz,
                     float dpl_cell, float* slab, int* flags, int* nsteps, RayPath* path = nullptr, int sub = 0)
{
    const int nnx = g.nnx, nnz = g.nnz;
    const float gox = g.gox, goz = g.goz, dnx = g.dnx, dnz = g.dnz, earth = g.earth;
    const float goxr = s.rgox, gozr = s.rgoz, dnxr = s.rdnx, dnzr = s.rdnz;
    const float dvx = g.dvx, dvz = g.dvz, scx = s.scx, scz = s.scz;
    const int gdx = g.gdx, gdz = g.gdz, ldr = s.rnz;
    const long maxrp = (long)nnx * (long)nnz;
    const float dpl = 0.5f * dpl_cell;
    const int isx = (int)((scx - goxr) / dnxr) + 1;
    const int isz = (int)((scz - gozr) / dnzr) + 1;
    int steps = 0;

    int ipx = (int)((rcx - gox) / dnx) + 1;
    int ipz = (int)((rcz - goz) / dnz) + 1;
    if (ipx < 1 || ipx >= nnx || ipz < 1 || ipz >= nnz) { *nsteps = 0; return -1; }

    float rgx = rcx, rgz = rcz;
    float sred = sq((scx - rgx) * earth);
    float sin_at = sinf_libm(rgx);           // sin of the ray point's colatitude, carried from the end of a step to the start of the next
    sred = sred + sq((scz - rgz) * earth * sin_at);
    sred = sqrtf(sred);
    bool sw = sred < 2.0f * dpl;
    int ipxr = (int)((rcx - goxr) / dnxr) + 1;
    int ipzr = (int)((rcz - gozr) / dnzr) + 1;
    bool igref = refined_cell_alive(s, f.Sr, ipxr, ipzr);
    if (!sw && igref && ipxr == isx && ipzr == isz) sw = true;
    if (path) { path->push(rgx, rgz); if (sw) path->push(scx, scz); }

    PatchAcc<LPR> acc;
    acc.init(slab, g.nvx + 2, sub);
    const RecipF by_dnx = recipf_of(dnx), by_dnz = recipf_of(dnz), by_dnxr = recipf_of(dnxr), by_dnzr = recipf_of(dnzr);
    const RecipF by_dvx = recipf_of(dvx), by_dvz = recipf_of(dvz);
    const RecipF by_gx = recipf_of(2.0f * earth * dnx), by_gxr = recipf_of(2.0f * earth * dnxr), by_6 = recipf_of(6.0f);
    int ivx_at = (ipx - 1) / gdx + 1, ivz_at = (ipz - 1) / gdz + 1;          // the vertex cell of (ipx, ipz), carried from step to step
    for (long j = 1; j <= maxrp && !sw; ++j) {
        float dtx, dtz;
        const float sin_rgx = sin_at;
        if (igref) {
            const float* q = f.Tr + (size_t)(ipxr - 1) * ldr + (ipzr - 1);
            const float t00 = q[0], t01 = q[1], t10 = q[ldr], t11 = q[ldr + 1];   // t[x][z]
            dtx = t10 - t00;
            dtx = dtx + t11 - t01;
            dtx = divf_by(dtx, by_gxr);
            dtz = t01 - t00;
            dtz = dtz + t11 - t10;
            dtz = dtz / (2.0f * earth * sin_rgx * dnzr);
        } else {
            const float t00 = t_value(f.F[rec_index(g.nbz, ipz - 1, ipx - 1)]), t01 = t_value(f.F[rec_index(g.nbz, ipz, ipx - 1)]);
            const float t10 = t_value(f.F[rec_index(g.nbz, ipz - 1, ipx)]), t11 = t_value(f.F[rec_index(g.nbz, ipz, ipx)]);
            dtx = t10 - t00;
            dtx = dtx + t11 - t01;
            dtx = divf_by(dtx, by_gx);
            dtz = t01 - t00;
            dtz = dtz + t11 - t10;
            dtz = dtz / (2.0f * earth * sin_rgx * dnz);
        }
        const float rd1 = sqrtf(sq(dtx) + sq(dtz));
        float rgx1 = rgx - dpl * dtx / (earth * rd1);
        float rgz1 = rgz - dpl * dtz / (earth * sin_rgx * rd1);
        steps += 1;

        const int ipxo = ipx, ipzo = ipz;
        ipxr = (int)divf_by(rgx1 - goxr, by_dnxr) + 1;
        ipzr = (int)divf_by(rgz1 - gozr, by_dnzr) + 1;
        igref = refined_cell_alive(s, f.Sr, ipxr, ipzr);
        ipx = (int)divf_by(rgx1 - gox, by_dnx) + 1;
        ipz = (int)divf_by(rgz1 - goz, by_dnz) + 1;

        sred = sq((scx - rgx1) * earth);
        sin_at = sinf_libm(rgx1);
        sred = sred + sq((scz - rgz1) * earth * sin_at);
        sred = sqrtf(sred);
        sw = sred < 2.0f * dpl;
        if (!sw && igref && ipxr == isx && ipzr == isz) sw = true;

        if (ipx < 1) { rgx1 = gox; ipx = 1; *flags |= 1; sin_at = sinf_libm(rgx1); }
        if (ipx >= nnx) { rgx1 = gox + (float)(nnx - 1) * dnx; ipx = nnx - 1; *flags |= 1; sin_at = sinf_libm(rgx1); }
        if (ipz < 1) { rgz1 = goz; ipz = 1; *flags |= 1; }
        if (ipz >= nnz) { rgz1 = goz + (float)(nnz - 1) * dnz; ipz = nnz - 1; *flags |= 1; }
        if (path) { path->push(rgx1, rgz1); if (sw) path->push(scx, scz); }

        // split the segment where it crosses a vertex-cell face (reference :2112-2156)
        const int ivx = (ipx - 1) / gdx + 1, ivz = (ipz - 1) / gdz + 1;
        const int ivxo = ivx_at, ivzo = ivz_at;            // = (ipxo - 1) / gdx + 1, (ipzo - 1) / gdz + 1
        ivx_at = ivx; ivz_at = ivz;
        int nhp = 0;
        int chp0 = 0, chp1 = 0;
        float vr0 = 0.0f, vr1 = 0.0f;
        if (ivx != ivxo) {
            nhp = 1;
            const float xi = (ivx > ivxo) ? gox + (float)(ivx - 1) * dvx : gox + (float)ivx * dvx;
            vr0 = (xi - rgx) / (rgx1 - rgx);
            chp0 = 1;
        }
        if (ivz != ivzo) {
            const float zi = (ivz > ivzo) ? goz + (float)(ivz - 1) * dvz : goz + (float)ivz * dvz;
            const float r = (zi - rgz) / (rgz1 - rgz);
            if (nhp == 0) { vr0 = r; chp0 = 2; }
            else if (r >= vr0) { vr1 = r; chp1 = 2; }
            else { vr1 = vr0; chp1 = chp0; vr0 = r; chp0 = 2; }
            nhp += 1;
        }
        nhp += 1;     // the end point, ratio 1

        float drx = (rgx - gox) - (float)(ipxo - 1) * dnx;
        float drz = (rgz - goz) - (float)(ipzo - 1) * dnz;
        float vel = 0.0f;
        for (int l = 1; l <= 2; ++l)
            for (int m = 1; m <= 2; ++m) {
                float produ = (1.0f - fabsf(divf_by((float)(m - 1) * dnz - drz, by_dnz)));
                produ = produ * (1.0f - fabsf(divf_by((float)(l - 1) * dnx - drx, by_dnx)));
                if (ipzo - 1 + m <= nnz && ipxo - 1 + l <= nnx)
                    vel = vel + f.veln[(size_t)(ipxo - 1 + l - 1) * nnz + (ipzo - 1 + m - 1)] * produ;
            }
        drx = (rgx - gox) - (float)(ivxo - 1) * dvx;
        drz = (rgz - goz) - (float)(ivzo - 1) * dvz;
        float vi[4], wi[4];
        bspline4_by(divf_by(drx, by_dvx), vi, by_6);
        bspline4_by(divf_by(drz, by_dvz), wi, by_6);
        int ivxt = ivxo, ivzt = ivzo;
        float vprev = 0.0f;
        bool wok = LPR == 1 && weights_plain(vi) && weights_plain(wi);
        for (int k = 1; k <= nhp; ++k) {
            const float velo = vel;
            float vio[4], wio[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { vio[q] = vi[q]; wio[q] = wi[q]; }
            if (k > 1) {
                const int c = (k == 2) ? chp0 : chp1;
                if (c == 1) ivxt = ivx;
                else if (c == 2) ivzt = ivz;
            }
            const float vrat = (k == nhp) ? 1.0f : (k == 1 ? vr0 : vr1);
            const float rigz = rgz + vrat * (rgz1 - rgz);
            const float rigx = rgx + vrat * (rgx1 - rgx);
            const int ipxt = (int)divf_by(rigx - gox, by_dnx) + 1;
            const int ipzt = (int)divf_by(rigz - goz, by_dnz) + 1;
            drx = (rigx - gox) - (float)(ipxt - 1) * dnx;
            drz = (rigz - goz) - (float)(ipzt - 1) * dnz;
            vel = 0.0f;
            for (int m = 1; m <= 2; ++m)
                for (int n = 1; n <= 2; ++n) {
                    float produ = (1.0f - fabsf(divf_by((float)(n - 1) * dnz - drz, by_dnz)));
                    produ = produ * (1.0f - fabsf(divf_by((float)(m - 1) * dnx - drx, by_dnx)));
                    if (ipzt - 1 + n <= nnz && ipxt - 1 + m <= nnx && ipzt - 1 + n >= 1 && ipxt - 1 + m >= 1)
                        vel = vel + f.veln[(size_t)(ipxt - 1 + m - 1) * nnz + (ipzt - 1 + n - 1)] * produ;
                }
            drx = (rigx - gox) - (float)(ivxt - 1) * dvx;
            drz = (rigz - goz) - (float)(ivzt - 1) * dvz;
            bspline4_by(divf_by(drx, by_dvx), vi, by_6);
            bspline4_by(divf_by(drz, by_dvz), wi, by_6);
            const float dinc = (k == 1) ? vrat * dpl : (vrat - vprev) * dpl;
            vprev = vrat;
            acc.move(ivxt - 1, ivzt - 1);
            const float v2 = sq(vel), vo2 = sq(velo);
            const bool wok_old = wok;
            wok = weights_plain(vi) && weights_plain(wi);
            if (LPR == 1 && wok && wok_old) {
                // (the thirty-two quotients of a sub-segment through two reciprocals: every weight is zero or at least 2^-52, so a product of two is
                // zero or at least 2^-104 and the squared velocities are of order ten -- outside v_div_scale's reach, see divf_by)
                const RecipF by_v2 = recipf_of(v2), by_vo2 = recipf_of(vo2);
#pragma unroll
                for (int l = 0; l < 4; ++l)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        float r1 = divf_by(vi[m] * wi[l], by_v2);
                        const float r2 = divf_by(vio[m] * wio[l], by_vo2);
                        r1 = -(r1 + r2) * dinc / 2.0f;
                        acc.a[l][m] = r1 + acc.a[l][m];
                    }
            } else if (LPR == 1) {
#pragma unroll
                for (int l = 0; l < 4; ++l)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        float r1 = vi[m] * wi[l] / v2;
                        const float r2 = vio[m] * wio[l] / vo2;
                        r1 = -(r1 + r2) * dinc / 2.0f;
                        acc.a[l][m] = r1 + acc.a[l][m];
                    }
            } else {
                const int l = acc.my_row();
                const float wl = l == 0 ? wi[0] : l == 1 ? wi[1] : l == 2 ? wi[2] : wi[3];
                const float wol = l == 0 ? wio[0] : l == 1 ? wio[1] : l == 2 ? wio[2] : wio[3];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float r1 = vi[m] * wl / v2;
                    const float r2 = vio[m] * wol / vo2;
                    r1 = -(r1 + r2) * dinc / 2.0f;
                    acc.a[0][m] = r1 + acc.a[0][m];
                }
            }
        }
        rgx = rgx1; rgz = rgz1;
    }
    acc.store();
    *nsteps = steps;
    return 0;
}

// d(vp)/d(vs) and d(rho)/d(vs) of the Brocher relations the reference hard-wires
// (CalSurfG.f90:1385-1423); `shallow` = depz(nz-1) < 35 km picks the coefficient set
DSA_HD void brocher_chain(float v, bool shallow, float* coe_a, float* coe_rho)
{
    const float v2 = v * v, v3 = (v * v) * v, v4 = ((v * v) * v) * v;
    float a, vpft;
    if (shallow) {
        a = 2.0947f - (0.8206f * 2.0f) * v + (0.2683f * 3.0f) * v2 - (0.0251f * 4.0f) * v3;
        vpft = 0.9409f + 2.0947f * v - 0.8206f * v2 + 0.2683f * v3 - 0.0251f * v4;
    } else {
        a = 2.2110f - (0.8984f * 2.0f) * v + (0.2786f * 3.0f) * v2 - (0.02412f * 4.0f) * v3;
        vpft = 0.9098f + 2.2110f * v - 0.8984f * v2 + 0.2786f * v3 - 0.02412f * v4;
    }
    const float p2 = vpft * vpft, p3 = (vpft * vpft) * vpft, p4 = ((vpft * vpft) * vpft) * vpft;
    *coe_a = a;
    *coe_rho = a * (1.6612f - (0.4721f * 2.0f) * vpft + (0.0671f * 3.0f) * p2 - (0.0043f * 4.0f) * p3 + (0.000106f * 5.0f) * p4);
}

}  // namespace dsa
