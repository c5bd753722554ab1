// Surface-wave dispersion of one layered column: fundamental-mode phase / group velocity at a list
// of periods (reference surfdisp96.f:52-1062 as driven by CalSurfG.f90 depthkernel :1-169 and
// caldespersion :2866-2927: spherical earth, mode 1, one wave type per call).
//
// One curve is a serial chain (the root search at period k starts from the root at k-1), so the
// device runs one curve per lane and gets its parallelism from columns x perturbations.  The layer
// arrays of a lane live in a strided workspace (element l of lane t at [l * stride + t]) so that
// the lanes of a wavefront read consecutive addresses.
//
// Arithmetic contract: fp64 secular functions with the reference's operation order; the
// single-precision temporaries of the F77 original (implicit REAL*4: cc1, betmx, t1a, t1b, cc0,
// gvel, ...) are kept single.  Everything that does not depend on the velocities -- the flattened
// layer thicknesses (logs), the flattening factors and the density exponent (powf) -- comes from
// the host in LayerGeom, computed with libm, so the only device transcendentals are the
// sin / cos / exp inside the secular functions.
#pragma once

#include "eikonal_core.h"

namespace dsa {

// sin and cos of one argument: on the device one call shares the argument reduction (same bits as the two calls, fewer
// instructions); the host build keeps libm's two calls.
DSA_HD void sin_cos(double x, double* s, double* c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(x, s, c);
#else
    *s = sin(x); *c = cos(x);
#endif
}

// IEEE division by a denominator that serves several quotients (round 5).  The compiler expands `x / d` on this hardware into v_div_scale (x2),
// v_rcp, two Newton steps on the reciprocal, the quotient, its remainder, v_div_fmas and v_div_fixup: eleven instructions, sixteen times per
// layer of the Rayleigh secular function -- a fifth of its instruction stream.  recip_of / div_by are that expansion split where it splits: the
// reciprocal's four FMAs depend on the denominator alone (five quotients share the layer product's norm, three the density, every layer
// 1 / omega), the quotient takes three more and the fix-up of the special cases (zeros and their signs, infinities, NaN: as the full division
// has them).  What is left out is the rescaling of v_div_scale / v_div_fmas, which acts only on operands it would otherwise lose bits of: a
// denominator that is denormal or beyond 2^1021, exponents 768 apart, a quotient in the denormal range, a numerator below 2^-970 -- such a
// quotient (a component 290 orders of magnitude under the layer product's norm) may round differently in its last bit here.  Everywhere else
// the value is the full division's, bit for bit: the same instructions on the same operands.  The host build divides.
struct Recip { double d, r; };
DSA_HD Recip recip_of(double d)
{
    Recip R;
    R.d = d;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSA_DISP_PLAIN_DIV)          // (-DDSA_DISP_PLAIN_DIV: the compiler's division, for the A/B of profiles/r05_ab_dispersion.log)
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    R.r = __builtin_fma(r, e, r);
#else
    R.r = 0.0;
#endif
    return R;
}
DSA_HD double div_by(double x, const Recip& R)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DSA_DISP_PLAIN_DIV)
    const double q = x * R.r;
    const double rem = __builtin_fma(-R.d, q, x);
    return __builtin_amdgcn_div_fixup(__builtin_fma(rem, R.r, q), R.d, x);
#else
    return x / R.d;
#endif
}

DSA_HD double larger_abs(double t, double v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmax(t, __builtin_fabs(v));
#else
    return fabs(v) > t ? fabs(v) : t;
#endif
}

constexpr int kMaxLayers = 200;     // reference NL
constexpr int kMaxPeriods = 60;     // reference NP

// velocity-independent part of the layered model of a call (host-made, refineGrid2LayerMdl
// CalSurfG.f90:2352-2411 + sphere surfdisp96.f:480-547)
struct LayerGeom {
    int nz;                 // grid depths
    int rmax;               // layers after refinement, incl. the half space
    int nsub[kMaxLayers];   // sublayers of grid interval i (0-based, nz-1 of them)
    float dflat[kMaxLayers];    // flattened thickness of layer l
    double tmp[kMaxLayers];     // velocity flattening factor (ar+ar)/(r0+r1)
    float rhofac_love[kMaxLayers], rhofac_rayl[kMaxLayers];   // btp**(-5), btp**(-2.275)
};

// strided view of one lane's layers
struct Layers {
    float* d; float* a; float* b; float* rho;
    size_t stride;
    int mmax, llw;
    // Optional lane group (device, Rayleigh): `gsize` lanes of one wavefront work on the same curve, all running the same
    // root search; inside the secular function lane `gsub` forms the layer matrix of every gsize-th layer and the lanes
    // exchange the matrices through `xch` (15 doubles per lane, LDS).  gsize = 1: one lane per curve.
    double* xch = nullptr;
    int gsize = 1, gsub = 0;
    DSA_HDM float D(int k) const { return d[(size_t)k * stride]; }      // k 0-based
    DSA_HDM float A(int k) const { return a[(size_t)k * stride]; }
    DSA_HDM float B(int k) const { return b[(size_t)k * stride]; }
    DSA_HDM float R(int k) const { return rho[(size_t)k * stride]; }
};

DSA_HD double sign1(double x) { return __builtin_copysign(1.0, x); }

// Love: Thomson-Haskell, surfdisp96.f:704-763
DSA_HD double dltar1(const Layers& m, double wvno, double omega)
{
    const int mmax = m.mmax;
    double beta1 = (double)m.B(mmax - 1);
    double rho1 = (double)m.R(mmax - 1);
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    // propagator terms of layer k: xmu, y, z, cosq
    auto layer_terms1 = [&](int k, double* c) {
        const double beta = (double)m.B(k - 1);
        const double rho = (double)m.R(k - 1);
        const double dk = (double)m.D(k - 1);
        const double xmu = rho * beta * beta;
        const double xkb_ = omega / beta;
        const double wp = wvno + xkb_;
        const double wm = fabs(wvno - xkb_);
        const double rb_ = sqrt(wp * wm);
        const double q = dk * rb_;
        double sinq, y, z, cosq;
        if (wvno < xkb_) {
            sin_cos(q, &sinq, &cosq); y = sinq / rb_; z = -rb_ * sinq;
        } else if (wvno == xkb_) {
            cosq = 1.0; y = dk; z = 0.0;
        } else {
            double fac = 0.0;
            if (q < 16) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            sinq = (1.0 - fac) * 0.5;
            y = sinq / rb_; z = rb_ * sinq;
        }
        c[0] = xmu; c[1] = y; c[2] = z; c[3] = cosq;
    };
    auto product_step1 = [&](const double* c) {
        const double xmu = c[0], y = c[1], z = c[2], cosq = c[3];
        const double e10 = e1 * cosq + e2 * xmu * z;
        const double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10);
        const double ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.e-40) xnor = 1.0;
        const Recip by_xnor = recip_of(xnor);
        e1 = div_by(e10, by_xnor);
        e2 = div_by(e20, by_xnor);
    };
    if (m.gsize <= 1) {
        for (int k = mmax - 1; k >= m.llw; --k) {
            double c[4];
            layer_terms1(k, c);
            product_step1(c);
        }
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        const int K = m.gsize;                       // lane groups: see dltar4
        for (int kb = mmax - 1; kb >= m.llw; kb -= K) {
            const int k = kb - m.gsub;
            if (k >= m.llw) {
                double c[4];
                layer_terms1(k, c);
                double* mine = m.xch + m.gsub * 15;
                for (int i = 0; i < 4; ++i) mine[i] = c[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int nb = (kb - m.llw + 1) < K ? (kb - m.llw + 1) : K;
            for (int j = 0; j < nb; ++j) {
                double c[4];
                const double* theirs = m.xch + j * 15;
                for (int i = 0; i < 4; ++i) c[i] = theirs[i];
                product_step1(c);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
#endif
    }
    return e1;
}

// products of the P / SV eigenfunctions over one layer, surfdisp96.f:868-985
struct LayerTerms { double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz, w, cosp; };

DSA_HD void layer_terms(double p, double q, double ra, double rb, double wvno, double xka, double xkb, double dpth, LayerTerms& o)
{
    double pex = 0.0, sex = 0.0, sinp = 0.0, x, sinq = 0.0, y, z, cosq, fac, w, cosp;
    // (w = sinp / ra and y = sinq / rb of the oscillating and of the evanescent branch: one division behind the branches, so that a wavefront
    // whose lanes sit on both sides of a layer's velocity divides once)
    if (wvno < xka) {
        sin_cos(p, &sinp, &cosp); x = -ra * sinp;
    } else if (wvno == xka) {
        cosp = 1.0; x = 0.0;
    } else {
        pex = p; fac = 0.0;
        if (p < 16) fac = exp(-2.0 * p);
        cosp = (1.0 + fac) * 0.5;
        sinp = (1.0 - fac) * 0.5;
        x = ra * sinp;
    }
    w = div_by(sinp, recip_of(ra));
    if (wvno == xka) w = dpth;
    if (wvno < xkb) {
        sin_cos(q, &sinq, &cosq); z = -rb * sinq;
    } else if (wvno == xkb) {
        cosq = 1.0; z = 0.0;
    } else {
        sex = q; fac = 0.0;
        if (q < 16) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        sinq = (1.0 - fac) * 0.5;
        z = rb * sinq;
    }
    y = div_by(sinq, recip_of(rb));
    if (wvno == xkb) y = dpth;
    const double exa = pex + sex;
    o.a0 = 0.0;
    if (exa < 60.0) o.a0 = exp(-exa);
    o.cpcq = cosp * cosq;
    o.cpy = cosp * y;
    o.cpz = cosp * z;
    o.cqw = cosq * w;
    o.cqx = cosq * x;
    o.xy = x * y;
    o.xz = x * z;
    o.wy = w * y;
    o.wz = w * z;
    o.w = w;
    o.cosp = cosp;
}

// Rayleigh: Dunkin's compound matrix, surfdisp96.f:767-865 with dnka :1018-1062 and normc :989-1014
DSA_HD double dltar4(const Layers& m, double wvno, double omga)
{
    const int mmax = m.mmax;
    double e0, e1, e2, e3, e4;
    LayerTerms o;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    const double wvno2 = wvno * wvno;
    double xka = omega / (double)m.A(mmax - 1);
    double xkb = omega / (double)m.B(mmax - 1);
    double wvnop = wvno + xka;
    double wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)m.B(mmax - 1) / omega;
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    {
        const double gamm1 = gam - 1.0;
        const double rho1 = (double)m.R(mmax - 1);
        e0 = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
        e1 = -rho1 * ra;
        e2 = rho1 * (gamm1 - gammk * ra * rb);
        e3 = rho1 * rb;
        e4 = wvno2 - ra * rb;
    }
    const double tt = -2.0 * wvno2;
    const Recip by_omega = recip_of(omega);
    // layer matrix of layer k (compound matrix ca(i, j), named cIJ; the entries that are copies or tt-multiples of others
    // are formed in the product step)
    auto layer_matrix = [&](int k, double* c) {
        const double ak = (double)m.A(k - 1), bk = (double)m.B(k - 1);
        const double xka_ = div_by(omega, recip_of(ak));
        const double xkb_ = div_by(omega, recip_of(bk));
        const double t_ = div_by(bk, by_omega);
        const double gammk_ = 2.0 * t_ * t_;
        const double gam_ = gammk_ * wvno2;
        double wp = wvno + xka_;
        double wm = fabs(wvno - xka_);
        const double ra_ = sqrt(wp * wm);
        wp = wvno + xkb_;
        wm = fabs(wvno - xkb_);
        const double rb_ = sqrt(wp * wm);
        const double dpth = (double)m.D(k - 1);
        const double rho = (double)m.R(k - 1);
        LayerTerms o;
        layer_terms(ra_ * dpth, rb_ * dpth, ra_, rb_, wvno, xka_, xkb_, dpth, o);
        const double one = 1.0, two = 2.0;
        const double gamm1 = gam_ - one;
        const double twgm1 = gam_ + gamm1;
        const double gmgmk = gam_ * gammk_;
        const double gmgm1 = gam_ * gamm1;
        const double gm1sq = gamm1 * gamm1;
        const double rho2 = rho * rho;
        const Recip by_rho = recip_of(rho), by_rho2 = recip_of(rho2);
        const double a0pq = o.a0 - o.cpcq;
        c[0] = o.cpcq - two * gmgm1 * a0pq - gmgmk * o.xz - wvno2 * gm1sq * o.wy;                      // c11
        c[1] = div_by(wvno2 * o.cpy - o.cqx, by_rho);                                                  // c12
        c[2] = div_by(-(twgm1 * a0pq + gammk_ * o.xz + wvno2 * gamm1 * o.wy), by_rho);                 // c13
        c[3] = div_by(o.cpz - wvno2 * o.cqw, by_rho);                                                  // c14
        c[4] = div_by(-(two * wvno2 * a0pq + o.xz + wvno2 * wvno2 * o.wy), by_rho2);                   // c15
        c[5] = (gmgmk * o.cpz - gm1sq * o.cqw) * rho;                                                  // c21
        c[6] = o.cpcq;                                                                                 // c22
        c[7] = gammk_ * o.cpz - gamm1 * o.cqw;                                                         // c23
        c[8] = -o.wz;                                                                                  // c24
        c[9] = (gm1sq * o.cpy - gmgmk * o.cqx) * rho;                                                  // c41
        c[10] = -o.xy;                                                                                 // c42
        c[11] = gamm1 * o.cpy - gammk_ * o.cqx;                                                        // c43
        c[12] = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * o.xz + gm1sq * gm1sq * o.wy) * rho2;    // c51
        c[13] = -(gammk_ * gamm1 * twgm1 * a0pq + gam_ * gammk_ * gammk_ * o.xz + gamm1 * gm1sq * o.wy) * rho;   // c53
        c[14] = o.a0 + two * (o.cpcq - c[0]);                                                          // c33
    };
    // e <- normalised e * ca
    auto product_step = [&](const double* c) {
        const double c11 = c[0], c12 = c[1], c13 = c[2], c14 = c[3], c15 = c[4], c21 = c[5], c22 = c[6], c23 = c[7], c24 = c[8];
        const double c41 = c[9], c42 = c[10], c43 = c[11], c51 = c[12], c53 = c[13], c33 = c[14];
        const double c25 = c14, c44 = c22, c45 = c12, c52 = c41, c54 = c21, c55 = c11;
        const double c31 = tt * c53;
        const double c32 = tt * c43;
        const double c34 = tt * c23;
        const double c35 = tt * c13;
        // ee(i) = sum_j e(j) ca(j, i), accumulated from zero in j order
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0, n4 = 0.0;
        n0 = n0 + e0 * c11; n0 = n0 + e1 * c21; n0 = n0 + e2 * c31; n0 = n0 + e3 * c41; n0 = n0 + e4 * c51;
        n1 = n1 + e0 * c12; n1 = n1 + e1 * c22; n1 = n1 + e2 * c32; n1 = n1 + e3 * c42; n1 = n1 + e4 * c52;
        n2 = n2 + e0 * c13; n2 = n2 + e1 * c23; n2 = n2 + e2 * c33; n2 = n2 + e3 * c43; n2 = n2 + e4 * c53;
        n3 = n3 + e0 * c14; n3 = n3 + e1 * c24; n3 = n3 + e2 * c34; n3 = n3 + e3 * c44; n3 = n3 + e4 * c54;
        n4 = n4 + e0 * c15; n4 = n4 + e1 * c25; n4 = n4 + e2 * c35; n4 = n4 + e3 * c45; n4 = n4 + e4 * c55;
        // (normc :989-1014: `if (dabs(ee(i)) .gt. t1) t1 = dabs(ee(i))` -- the larger of the two, t1 when ee(i) is NaN: v_max_f64 on the device)
        double t1 = 0.0;
        t1 = larger_abs(t1, n0); t1 = larger_abs(t1, n1); t1 = larger_abs(t1, n2); t1 = larger_abs(t1, n3); t1 = larger_abs(t1, n4);
        if (t1 < 1.e-40) t1 = 1.0;
        const Recip by_t1 = recip_of(t1);
        e0 = div_by(n0, by_t1); e1 = div_by(n1, by_t1); e2 = div_by(n2, by_t1); e3 = div_by(n3, by_t1); e4 = div_by(n4, by_t1);
    };
    if (m.gsize <= 1) {
        for (int k = mmax - 1; k >= m.llw; --k) {
            double c[15];
            layer_matrix(k, c);
            product_step(c);
        }
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        // the lanes of a group take one layer each, park its matrix in LDS, then every lane multiplies the group's
        // matrices in layer order (DS operations of a wavefront execute in order)
        const int K = m.gsize;
        for (int kb = mmax - 1; kb >= m.llw; kb -= K) {
            const int k = kb - m.gsub;
            if (k >= m.llw) {
                double c[15];
                layer_matrix(k, c);
                double* mine = m.xch + m.gsub * 15;
                for (int i = 0; i < 15; ++i) mine[i] = c[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const int nb = (kb - m.llw + 1) < K ? (kb - m.llw + 1) : K;
            for (int j = 0; j < nb; ++j) {
                double c[15];
                const double* theirs = m.xch + j * 15;
                for (int i = 0; i < 15; ++i) c[i] = theirs[i];
                product_step(c);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
#endif
    }
    if (m.llw != 1) {
        // water layer on top
        xka = omega / (double)m.A(0);
        wvnop = wvno + xka;
        wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        const double dpth = (double)m.D(0);
        const double rho1 = (double)m.R(0);
        const double znul = 1.0e-05;
        layer_terms(ra * dpth, znul, ra, znul, wvno, xka, znul, dpth, o);
        const double w0 = -rho1 * o.w;
        return o.cosp * e0 + w0 * e1;
    }
    return e0;
}

template <int IFUNC>
DSA_HD double secular(const Layers& m, double wvno, double omega)
{
    return IFUNC == 1 ? dltar1(m, wvno, omega) : dltar4(m, wvno, omega);
}

// hybrid interval halving / Neville iteration, surfdisp96.f:551-668
template <int IFUNC>
DSA_HD double nevill(const Layers& m, double t, double c1, double c2, double del1, double del2, double twopi)
{
    double x[12], y[12], c3, del3;
    const double omega = twopi / t;
    int nev, nctrl = 1, mm = 1;
    c3 = 0.5 * (c1 + c2);
    del3 = secular<IFUNC>(m, omega / c3, omega);
    nev = 1;
    for (;;) {
        nctrl = nctrl + 1;
        if (nctrl >= 100) break;
        if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
            nev = 0;
            c3 = 0.5 * (c1 + c2);
            del3 = secular<IFUNC>(m, omega / c3, omega);
        }
        const double s13 = del1 - del3;
        const double s32 = del3 - del2;
        if (sign1(del3) * sign1(del1) < 0.0) { c2 = c3; del2 = del3; }
        else { c1 = c3; del1 = del3; }
        if (fabs(c1 - c2) <= 1.e-6 * c1) break;
        if (sign1(s13) != sign1(s32)) nev = 0;
        const double ss1 = fabs(del1);
        const double s1 = (double)0.01f * ss1;          // single-precision literal in the reference
        const double ss2 = fabs(del2);
        const double s2 = (double)0.01f * ss2;
        bool halve = (s1 > ss2 || s2 > ss1 || nev == 0);
        if (!halve) {
            if (nev == 2) { x[mm + 1] = c3; y[mm + 1] = del3; }
            else { x[1] = c1; y[1] = del1; x[2] = c2; y[2] = del2; mm = 1; }
            for (int kk = 1; kk <= mm; ++kk) {
                const int j = mm - kk + 1;
                const double denom = y[mm + 1] - y[j];
                if (fabs(denom) < 1.0e-10 * fabs(y[mm + 1])) { halve = true; break; }
                x[j] = (-y[j] * x[j + 1] + y[mm + 1] * x[j]) / denom;
            }
            if (!halve) {
                c3 = x[1];
                del3 = secular<IFUNC>(m, omega / c3, omega);
                nev = 2;
                mm = mm + 1;
                if (mm > 10) mm = 10;
            }
        }
        if (halve) {
            c3 = 0.5 * (c1 + c2);
            del3 = secular<IFUNC>(m, omega / c3, omega);
            nev = 1;
            mm = 1;
        }
    }
    return c3;
}

// bracket a sign change in steps of dc, then refine; surfdisp96.f:384-476.  returns iret
template <int IFUNC>
DSA_HD int getsol(const Layers& m, double t1, double* c1io, double clow, double dc, double cm, float betmx, int ifirst, double* del1st)
{
    const double twopi = 2.0 * 3.141592653589793;
    double c1 = *c1io, c2, del1, del2;
    double omega = twopi / t1;
    double wvno = omega / c1;
    del1 = secular<IFUNC>(m, wvno, omega);
    if (ifirst == 1) *del1st = del1;
    const double plmn = sign1(*del1st) * sign1(del1);
    int idir = +1;
    if (ifirst != 1 && plmn < 0.0) idir = -1;
    // (bounded: with NaN in the model no comparison ever ends the reference's loop; a device lane must not spin)
    for (int guard = 0; ; ++guard) {
        if (guard > 100000) { *c1io = c1; return -1; }
        if (idir > 0) c2 = c1 + dc; else c2 = c1 - dc;
        if (c2 <= clow) { idir = +1; c1 = clow; continue; }
        omega = twopi / t1;
        wvno = omega / c2;
        del2 = secular<IFUNC>(m, wvno, omega);
        if (sign1(del1) != sign1(del2)) break;
        c1 = c2;
        del1 = del2;
        if (c1 < cm) { *c1io = c1; return -1; }
        if (c1 >= ((double)betmx + dc)) { *c1io = c1; return -1; }
    }
    c1 = nevill<IFUNC>(m, t1, c1, c2, del1, del2, twopi);
    *c1io = c1;
    if (c1 > (double)betmx) return -1;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The root search as a STATE MACHINE (round 3).  The reference nests its loops: per period the bracketing walk `getsol`
// (surfdisp96.f:384-476), inside it the hybrid halving / Neville refinement `nevill` (:551-668), and the secular function is called
// from seven places.  Sixty-four lanes running that nest re-converge at every loop exit, so a wavefront takes, per period, the
// LONGEST search among its lanes: measured lane fill 0.76 (profiles/r02_pmc_dispersion.txt).  Here the nest is unrolled into states:
// the only loop is "evaluate the secular function at my trial wavenumber, then advance my own search to its next trial", so every
// trip of that loop does useful work on every lane that still has a period to do, whatever stage its search is in.  The sequence of
// evaluations of a curve -- and therefore every bit of its result -- is the reference's.
struct RootSearch {
    // curve level (dispersion_curve)
    int k, kmax, igr, failed_k;
    int pass;                 // 1: the root at t1 (phase) / t1a (group), 2: the second root at t1b
    float t1a, t1b;
    double cc, cm, dc, c1, cprev, ck, del1st, clow;
    float betmx;
    // getsol
    int st;                   // what the pending evaluation is for (RS_*)
    int ifirst, idir, guard;
    double t1, omega, c2, del1, del2;
    // nevill
    int nev, nctrl, mm;
    double c3, del3, x[12], y[12];
    // the pending evaluation
    double wvno;
};
enum { RS_DONE = 0, RS_G1, RS_G2, RS_NA, RS_NB, RS_NCD };

// what follows a finished getsol (iret, root in r.c1): the curve's bookkeeping, then the next search; false when the curve is done
template <int IFUNC>
DSA_HD bool rs_after_getsol(RootSearch& r, int iret, const double* t, double* cg, size_t cstride);

// start a getsol at period r.t1 from r.c1 (surfdisp96.f:384-400)
DSA_HD void rs_getsol_start(RootSearch& r)
{
    const double twopi = 2.0 * 3.141592653589793;
    r.omega = twopi / r.t1;
    r.wvno = r.omega / r.c1;
    r.guard = 0;
    r.st = RS_G1;
}
// the bracketing walk up to its next evaluation; returns false when the search fails here (iret = -1)
DSA_HD bool rs_bracket_step(RootSearch& r)
{
    const double twopi = 2.0 * 3.141592653589793;
    for (;;) {
        // (bounded: with NaN in the model no comparison ever ends the reference's loop; a device lane must not spin)
        if (r.guard > 100000) return false;
        r.guard += 1;
        if (r.idir > 0) r.c2 = r.c1 + r.dc; else r.c2 = r.c1 - r.dc;
        if (r.c2 <= r.clow) { r.idir = +1; r.c1 = r.clow; continue; }
        r.omega = twopi / r.t1;
        r.wvno = r.omega / r.c2;
        r.st = RS_G2;
        return true;
    }
}
// nevill's loop body (surfdisp96.f:575-668) from the bracket update to its next evaluation; false: the refinement has ended (root r.c3)
DSA_HD bool rs_nevill_mid(RootSearch& r)
{
    const double s13 = r.del1 - r.del3;
    const double s32 = r.del3 - r.del2;
    if (sign1(r.del3) * sign1(r.del1) < 0.0) { r.c2 = r.c3; r.del2 = r.del3; }
    else { r.c1 = r.c3; r.del1 = r.del3; }
    if (fabs(r.c1 - r.c2) <= 1.e-6 * r.c1) return false;
    if (sign1(s13) != sign1(s32)) r.nev = 0;
    const double ss1 = fabs(r.del1);
    const double s1 = (double)0.01f * ss1;          // single-precision literal in the reference
    const double ss2 = fabs(r.del2);
    const double s2 = (double)0.01f * ss2;
    bool halve = (s1 > ss2 || s2 > ss1 || r.nev == 0);
    if (!halve) {
        if (r.nev == 2) { r.x[r.mm + 1] = r.c3; r.y[r.mm + 1] = r.del3; }
        else { r.x[1] = r.c1; r.y[1] = r.del1; r.x[2] = r.c2; r.y[2] = r.del2; r.mm = 1; }
        for (int kk = 1; kk <= r.mm; ++kk) {
            const int j = r.mm - kk + 1;
            const double denom = r.y[r.mm + 1] - r.y[j];
            if (fabs(denom) < 1.0e-10 * fabs(r.y[r.mm + 1])) { halve = true; break; }
            r.x[j] = (-r.y[j] * r.x[j + 1] + r.y[r.mm + 1] * r.x[j]) / denom;
        }
        if (!halve) {
            r.c3 = r.x[1];
            r.nev = 2;
            r.mm = r.mm + 1;
            if (r.mm > 10) r.mm = 10;
        }
    }
    if (halve) {
        r.c3 = 0.5 * (r.c1 + r.c2);
        r.nev = 1;
        r.mm = 1;
    }
    r.wvno = r.omega / r.c3;
    r.st = RS_NCD;
    return true;
}

// `del` = the secular function at the trial the search asked for: advance to the next trial.  Returns false when the curve is done.
template <int IFUNC>
DSA_HD bool rs_advance(RootSearch& r, double del, const double* t, double* cg, size_t cstride)
{
    bool in_nevill_top = false, in_nevill_mid = false;
    switch (r.st) {
    case RS_G1: {
        r.del1 = del;
        if (r.ifirst == 1) r.del1st = r.del1;
        const double plmn = sign1(r.del1st) * sign1(r.del1);
        r.idir = +1;
        if (r.ifirst != 1 && plmn < 0.0) r.idir = -1;
        if (!rs_bracket_step(r)) return rs_after_getsol<IFUNC>(r, -1, t, cg, cstride);
        return true;
    }
    case RS_G2: {
        r.del2 = del;
        if (sign1(r.del1) != sign1(r.del2)) {
            // nevill (surfdisp96.f:551-574): the midpoint first
            r.nctrl = 1; r.mm = 1;
            r.c3 = 0.5 * (r.c1 + r.c2);
            r.wvno = r.omega / r.c3;
            r.st = RS_NA;
            return true;
        }
        r.c1 = r.c2;
        r.del1 = r.del2;
        if (r.c1 < r.cm) return rs_after_getsol<IFUNC>(r, -1, t, cg, cstride);
        if (r.c1 >= ((double)r.betmx + r.dc)) return rs_after_getsol<IFUNC>(r, -1, t, cg, cstride);
        if (!rs_bracket_step(r)) return rs_after_getsol<IFUNC>(r, -1, t, cg, cstride);
        return true;
    }
    case RS_NA: r.del3 = del; r.nev = 1; in_nevill_top = true; break;
    case RS_NB: r.del3 = del; in_nevill_mid = true; break;
    case RS_NCD: r.del3 = del; in_nevill_top = true; break;
    default: return false;
    }
    for (;;) {
        if (in_nevill_top) {
            r.nctrl = r.nctrl + 1;
            if (r.nctrl >= 100) break;
            if (r.c3 < fmin(r.c1, r.c2) || r.c3 > fmax(r.c1, r.c2)) {
                r.nev = 0;
                r.c3 = 0.5 * (r.c1 + r.c2);
                r.wvno = r.omega / r.c3;
                r.st = RS_NB;
                return true;
            }
            in_nevill_mid = true;
        }
        if (in_nevill_mid) {
            if (!rs_nevill_mid(r)) break;
            return true;
        }
    }
    // the refinement has ended: getsol's tail (surfdisp96.f:470-476)
    r.c1 = r.c3;
    return rs_after_getsol<IFUNC>(r, r.c1 > (double)r.betmx ? -1 : 1, t, cg, cstride);
}

// set up period r.k (surfdisp96.f:223-262) and start its first search
DSA_HD void rs_period_start(RootSearch& r, const double* t)
{
    const float h = 0.005f;
    const double onea = (double)1.500f;
    double t1 = t[r.k - 1];
    r.t1b = 0.0f;
    if (r.igr > 0) {
        r.t1a = (float)(t1 / (double)(1.f + h));
        r.t1b = (float)(t1 / (double)(1.f - h));
        t1 = (double)r.t1a;
    } else {
        r.t1a = (float)t1;
    }
    if (r.k == 1) { r.c1 = r.cc; r.clow = r.cc; r.ifirst = 1; }
    else { r.ifirst = 0; r.c1 = r.cprev - onea * r.dc; r.clow = r.cm; }
    r.t1 = t1;
    r.pass = 1;
    rs_getsol_start(r);
}

template <int IFUNC>
DSA_HD bool rs_after_getsol(RootSearch& r, int iret, const double* t, double* cg, size_t cstride)
{
    const double one = 1.0e-2;
    const double onea = (double)1.500f;
    if (r.pass == 1) {
        if (iret == -1) {
            // the reference logs "improper initial value in disper - no zero found" and zero-fills the rest of the curve (:308-348)
            r.failed_k = r.k;
            for (int i = r.k; i <= r.kmax; ++i) cg[(size_t)(i - 1) * cstride] = 0.0;
            r.st = RS_DONE;
            return false;
        }
        r.ck = r.c1;
        r.cprev = r.ck;
        if (r.igr > 0) {
            r.t1 = (double)r.t1b;
            r.clow = 0.0 + one * r.dc;               // cb(k) is still zero here
            r.c1 = r.c1 - onea * r.dc;
            r.ifirst = 0;
            r.pass = 2;
            rs_getsol_start(r);
            return true;
        }
        r.c1 = 0.0;
    } else {
        if (iret == -1) r.c1 = r.ck;
    }
    const float cc0 = (float)r.ck;
    const float cc1b = (float)r.c1;
    if (r.igr == 0) cg[(size_t)(r.k - 1) * cstride] = (double)cc0;
    else {
        const float gvel = (1 / r.t1a - 1 / r.t1b) / (1 / (r.t1a * cc0) - 1 / (r.t1b * cc1b));
        cg[(size_t)(r.k - 1) * cstride] = (double)gvel;
    }
    r.k += 1;
    if (r.k > r.kmax) { r.st = RS_DONE; return false; }
    rs_period_start(r, t);
    return true;
}

// starting phase velocity from the half-space Rayleigh equation, all REAL*4; surfdisp96.f:361-382
DSA_HD float gtsolh(float a, float b)
{
    float c = 0.95f * b;
    for (int i = 0; i < 5; ++i) {
        const float gamma = b / a;
        const float kappa = c / b;
        const float k2 = kappa * kappa;
        const float gk2 = (gamma * kappa) * (gamma * kappa);
        const float fac1 = sqrtf(1.0f - gk2);
        const float fac2 = sqrtf(1.0f - k2);
        const float fr = (2.0f - k2) * (2.0f - k2) - 4.0f * fac1 * fac2;
        float frp = -(4.0f * (2.0f - k2) * kappa) + 4.0f * fac2 * gamma * gamma * kappa / fac1 + 4.0f * fac1 * kappa / fac2;
        frp = frp / b;
        c = c - fr / frp;
    }
    return c;
}

// Brocher relations vs -> vp -> rho as the column drivers apply them (CalSurfG.f90:49-53)
DSA_HD void brocher_vp_rho(float vs, float* vp, float* rho)
{
    const float v2 = vs * vs, v3 = vs * (vs * vs), v4 = ((vs * vs) * vs) * vs;
    const float p = 0.9409f + 2.0947f * vs - 0.8206f * v2 + 0.2683f * v3 - 0.0251f * v4;
    const float p2 = p * p, p3 = p * (p * p), p4 = ((p * p) * p) * p, p5 = (((p * p) * p) * p) * p;
    *vp = p;
    *rho = 1.6612f * p - 0.4721f * p2 + 0.0671f * p3 - 0.0043f * p4 + 0.000106f * p5;
}

// Fill a lane's layers from the column's grid values (vs, vp, rho at the nz depths): linear-gradient
// sublayers (refineGrid2LayerMdl) and earth flattening (sphere) with the host's geometry factors.
template <int IFUNC>
DSA_HD void build_layers(const LayerGeom& G, const float* vs, const float* vp, const float* rho, Layers& m)
{
    int k = 0;
    for (int i = 1; i <= G.nz - 1; ++i) {
        const int nsub = G.nsub[i - 1];
        for (int j = 1; j <= nsub; ++j) {
            const float rvp = vp[i - 1] + (float)(2 * j - 1) * (vp[i] - vp[i - 1]) / (float)(2 * nsub);
            const float rvs = vs[i - 1] + (float)(2 * j - 1) * (vs[i] - vs[i - 1]) / (float)(2 * nsub);
            const float rrho = rho[i - 1] + (float)(2 * j - 1) * (rho[i] - rho[i - 1]) / (float)(2 * nsub);
            const size_t o = (size_t)k * m.stride;
            m.d[o] = G.dflat[k];
            m.a[o] = (float)((double)rvp * G.tmp[k]);
            m.b[o] = (float)((double)rvs * G.tmp[k]);
            m.rho[o] = rrho * (IFUNC == 1 ? G.rhofac_love[k] : G.rhofac_rayl[k]);
            ++k;
        }
    }
    const size_t o = (size_t)k * m.stride;
    m.d[o] = 0.0f;
    m.a[o] = (float)((double)vp[G.nz - 1] * G.tmp[k]);
    m.b[o] = (float)((double)vs[G.nz - 1] * G.tmp[k]);
    m.rho[o] = rho[G.nz - 1] * (IFUNC == 1 ? G.rhofac_love[k] : G.rhofac_rayl[k]);
    m.mmax = k + 1;
    m.llw = (m.B(0) <= 0.0f) ? 2 : 1;
}

// One dispersion curve (surfdisp96.f:52-350 with mode = 1): cg[k] for k < kmax, written with `cstride`.
// igr = 0 phase velocity, 1 group velocity from two roots at T/(1 +- h).
// One dispersion curve (surfdisp96.f:52-350 with mode = 1): cg[k] for k < kmax, written with `cstride`.
// igr = 0 phase velocity, 1 group velocity from two roots at T/(1 +- h).
// Returns 0, or -- when no zero of the secular function was found for a period (the reference's "improper initial value in disper -
// no zero found" block on unit 66, surfdisp96.f:308-339, after which it zero-fills the rest of the curve, :342-348) -- the 1-based
// index k of that period.
template <int IFUNC>
DSA_HD int dispersion_curve_states(const Layers& m, int igr, int kmax, const double* t, double* cg, size_t cstride)
{
    const int mmax = m.mmax;
    const float ddc = 0.005f;
    int jmn = 1, jsol = 1;
    float betmx = -1.e20f, betmn = 1.e20f;
    for (int i = 0; i < mmax; ++i) {
        const float bi = m.B(i), ai = m.A(i);
        if (bi > 0.01f && bi < betmn) { betmn = bi; jmn = i + 1; jsol = 1; }
        else if (bi <= 0.01f && ai < betmn) { betmn = ai; jmn = i + 1; jsol = 0; }
        if (bi > betmx) betmx = bi;
    }
    float cc1;
    if (jsol == 0) cc1 = betmn;
    else cc1 = gtsolh(m.A(jmn - 1), m.B(jmn - 1));
    cc1 = .95f * cc1;
    cc1 = .90f * cc1;
    RootSearch r;
    r.k = 1; r.kmax = kmax; r.igr = igr; r.failed_k = 0;
    r.cc = (double)cc1;
    r.dc = fabs((double)ddc);
    r.cm = r.cc;
    r.c1 = r.cc; r.cprev = 0.0; r.del1st = 0.0; r.ck = 0.0; r.clow = r.cc;
    r.betmx = betmx;
    r.c2 = 0.0; r.del1 = 0.0; r.del2 = 0.0; r.c3 = 0.0; r.del3 = 0.0; r.nev = 0; r.nctrl = 0; r.mm = 1; r.idir = 1; r.guard = 0; r.ifirst = 1;
    for (int i = 0; i < 12; ++i) { r.x[i] = 0.0; r.y[i] = 0.0; }
    if (kmax < 1) return 0;
    rs_period_start(r, t);
    // the one loop: every trip evaluates the secular function once per lane, wherever the lane's search stands
    bool live = true;
    while (live) {
        const double del = secular<IFUNC>(m, r.wvno, r.omega);
        live = rs_advance<IFUNC>(r, del, t, cg, cstride);
    }
    return r.failed_k;
}

// The reference's own loop nest (kept for Love waves, see dispersion_curve).  Returns 0, or -- when no zero of the secular function was found for a period (the reference's "improper initial value in disper -
// no zero found" block on unit 66, surfdisp96.f:308-339, after which it zero-fills the rest of the curve, :342-348) -- the 1-based
// index k of that period.
// What the reference's unit-66 block prints about a curve that ended without a root (surfdisp96.f:327-337): the starting phase velocity
// and floor (cc, cm), the phase velocity the failed search stopped at (c1), and the roots of the periods before (c(1..k-1)).  Filled by
// the loop nest below when asked (the host's replay of a failing curve, Engine::dispersion_failure); the device passes none.
struct DispTrace { double cc, cm, c1; double c[kMaxPeriods]; };

template <int IFUNC>
DSA_HD int dispersion_curve_nested(const Layers& m, int igr, int kmax, const double* t, double* cg, size_t cstride, DispTrace* tr = nullptr)
{
    const int mmax = m.mmax;
    const float ddc = 0.005f, h = 0.005f;
    const float sone = 1.500f;
    const double one = 1.0e-2;
    int jmn = 1, jsol = 1;
    float betmx = -1.e20f, betmn = 1.e20f;
    for (int i = 0; i < mmax; ++i) {
        const float bi = m.B(i), ai = m.A(i);
        if (bi > 0.01f && bi < betmn) { betmn = bi; jmn = i + 1; jsol = 1; }
        else if (bi <= 0.01f && ai < betmn) { betmn = ai; jmn = i + 1; jsol = 0; }
        if (bi > betmx) betmx = bi;
    }
    const double onea = (double)sone;
    float cc1;
    if (jsol == 0) cc1 = betmn;
    else cc1 = gtsolh(m.A(jmn - 1), m.B(jmn - 1));
    cc1 = .95f * cc1;
    cc1 = .90f * cc1;
    const double cc = (double)cc1;
    const double dc = fabs((double)ddc);
    const double cm = cc;
    double c1 = cc, cprev = 0.0, del1st = 0.0;
    int k;
    bool failed = false;
    for (k = 1; k <= kmax; ++k) {
        double t1 = t[k - 1];
        float t1a, t1b = 0.0f;
        if (igr > 0) {
            t1a = (float)(t1 / (double)(1.f + h));
            t1b = (float)(t1 / (double)(1.f - h));
            t1 = (double)t1a;
        } else {
            t1a = (float)t1;
        }
        double clow;
        int ifirst;
        if (k == 1) { c1 = cc; clow = cc; ifirst = 1; }
        else { ifirst = 0; c1 = cprev - onea * dc; clow = cm; }
        int iret = getsol<IFUNC>(m, t1, &c1, clow, dc, cm, betmx, ifirst, &del1st);
        if (iret == -1) { failed = true; if (tr) { tr->cc = cc; tr->cm = cm; tr->c1 = c1; } break; }
        const double ck = c1;
        cprev = ck;
        if (tr) tr->c[k - 1] = ck;
        if (igr > 0) {
            t1 = (double)t1b;
            clow = 0.0 + one * dc;               // cb(k) is still zero here
            c1 = c1 - onea * dc;
            iret = getsol<IFUNC>(m, t1, &c1, clow, dc, cm, betmx, 0, &del1st);
            if (iret == -1) c1 = ck;
        } else c1 = 0.0;
        const float cc0 = (float)ck;
        const float cc1b = (float)c1;
        if (igr == 0) cg[(size_t)(k - 1) * cstride] = (double)cc0;
        else {
            const float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1b));
            cg[(size_t)(k - 1) * cstride] = (double)gvel;
        }
    }
    if (failed)
        for (int i = k; i <= kmax; ++i) cg[(size_t)(i - 1) * cstride] = 0.0;   // the reference logs a warning and zero-fills
    return failed ? k : 0;
}

// Which form runs where (measured, profiles/r03_ab_dispersion_states.txt; both give the reference's bits): the state machine for Rayleigh
// waves, whose secular function (Dunkin's compound matrices: ~700 instructions per layer) is what the lanes should spend their time in
// -- +16 % roots/s at the headline size (lane fill 0.76 -> 0.9), +18 % for group velocities; the loop nest for Love waves, whose
// Thomson-Haskell function is so cheap that the state bookkeeping costs more than the idle lanes did (-9 %).
template <int IFUNC>
DSA_HD int dispersion_curve(const Layers& m, int igr, int kmax, const double* t, double* cg, size_t cstride)
{
    return IFUNC == 2 ? dispersion_curve_states<IFUNC>(m, igr, kmax, t, cg, cstride) : dispersion_curve_nested<IFUNC>(m, igr, kmax, t, cg, cstride);
}

}  // namespace dsa
