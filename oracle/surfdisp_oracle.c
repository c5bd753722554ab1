/* TEST INFRASTRUCTURE ONLY -- dispersion side of the oracle (see dsurf_oracle.h).
 *
 * C restatement of the reference's surfdisp96 family (surfdisp96.f:52-1062: Thomson-Haskell for Love,
 * Dunkin's compound matrix for Rayleigh, bracketing in 0.005 km/s steps, Neville / interval-halving
 * refinement, earth flattening) and of the column drivers in CalSurfG.f90 (refineGrid2LayerMdl
 * :2352-2411, caldespersion :2866-2927, depthkernel :1-169).
 *
 * The reference is F77 with implicit typing: names starting with i-n are integers, everything else
 * undeclared is REAL*4.  That matters: several intermediates are single precision on purpose here
 * (cc1, betmx, betmn, t1a, t1b, cc0, gvel, dhalf), and single-precision literals inside double
 * expressions stay single (0.01*ss1 is dble(0.01f)*ss1).  Pinned bitwise against the reference's
 * own symbols in tests/test_oracle_vs_ref.py.
 */
#include "dsurf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NLMAX 200

typedef struct {
    int mmax, llw;
    float d[NLMAX], a[NLMAX], b[NLMAX], rho[NLMAX], rtp[NLMAX], dtp[NLMAX], btp[NLMAX];
    float dhalf;        /* `save dhalf` in sphere */
    double del1st;      /* `save del1st` in getsol */
} model;

static double dsign1(double x) { return copysign(1.0, x); }

/* surfdisp96.f:704-763 */
static double dltar1(const model *m, double wvno, double omega)
{
    const int mmax = m->mmax;
    double beta1 = (double)m->b[mmax - 1];
    double rho1 = (double)m->rho[mmax - 1];
    double xkb = omega / beta1;
    double wvnop = wvno + xkb;
    double wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double e1 = rho1 * rb;
    double e2 = 1.0 / (beta1 * beta1);
    for (int k = mmax - 1; k >= m->llw; --k) {
        beta1 = (double)m->b[k - 1];
        rho1 = (double)m->rho[k - 1];
        const double xmu = rho1 * beta1 * beta1;
        xkb = omega / beta1;
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        const double q = (double)m->d[k - 1] * rb;
        double sinq, y, z, cosq;
        if (wvno < xkb) {
            sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
        } else if (wvno == xkb) {
            cosq = 1.0; y = (double)m->d[k - 1]; z = 0.0;
        } else {
            double fac = 0.0;
            if (q < 16) fac = exp(-2.0 * q);
            cosq = (1.0 + fac) * 0.5;
            sinq = (1.0 - fac) * 0.5;
            y = sinq / rb; z = rb * sinq;
        }
        const double e10 = e1 * cosq + e2 * xmu * z;
        const double e20 = e1 * y / xmu + e2 * cosq;
        double xnor = fabs(e10);
        const double ynor = fabs(e20);
        if (ynor > xnor) xnor = ynor;
        if (xnor < 1.e-40) xnor = 1.0;
        e1 = e10 / xnor;
        e2 = e20 / xnor;
    }
    return e1;
}

typedef struct { double a0, cpcq, cpy, cpz, cqw, cqx, xy, xz, wy, wz; } ovr;

/* surfdisp96.f:868-985 */
static void var(double p, double q, double ra, double rb, double wvno, double xka, double xkb, double dpth,
                double *w, double *cosp, double *exa, ovr *o)
{
    double pex = 0.0, sex = 0.0, sinp, x = 0.0, sinq, y = 0.0, z = 0.0, cosq = 0.0, fac;
    *exa = 0.0;
    o->a0 = 1.0;
    if (wvno < xka) {
        sinp = sin(p); *w = sinp / ra; x = -ra * sinp; *cosp = cos(p);
    } else if (wvno == xka) {
        *cosp = 1.0; *w = dpth; x = 0.0;
    } else {
        pex = p; fac = 0.0;
        if (p < 16) fac = exp(-2.0 * p);
        *cosp = (1.0 + fac) * 0.5;
        sinp = (1.0 - fac) * 0.5;
        *w = sinp / ra; x = ra * sinp;
    }
    if (wvno < xkb) {
        sinq = sin(q); y = sinq / rb; z = -rb * sinq; cosq = cos(q);
    } else if (wvno == xkb) {
        cosq = 1.0; y = dpth; z = 0.0;
    } else {
        sex = q; fac = 0.0;
        if (q < 16) fac = exp(-2.0 * q);
        cosq = (1.0 + fac) * 0.5;
        sinq = (1.0 - fac) * 0.5;
        y = sinq / rb; z = rb * sinq;
    }
    *exa = pex + sex;
    o->a0 = 0.0;
    if (*exa < 60.0) o->a0 = exp(-*exa);
    o->cpcq = *cosp * cosq;
    o->cpy = *cosp * y;
    o->cpz = *cosp * z;
    o->cqw = cosq * *w;
    o->cqx = cosq * x;
    o->xy = x * y;
    o->xz = x * z;
    o->wy = *w * y;
    o->wz = *w * z;
}

/* surfdisp96.f:1018-1062 */
static void dnka(double ca[5][5], double wvno2, double gam, double gammk, double rho, const ovr *o)
{
    const double one = 1.0, two = 2.0;
    const double gamm1 = gam - one;
    const double twgm1 = gam + gamm1;
    const double gmgmk = gam * gammk;
    const double gmgm1 = gam * gamm1;
    const double gm1sq = gamm1 * gamm1;
    const double rho2 = rho * rho;
    const double a0pq = o->a0 - o->cpcq;
#define CA(i, j) ca[(i) - 1][(j) - 1]
    CA(1, 1) = o->cpcq - two * gmgm1 * a0pq - gmgmk * o->xz - wvno2 * gm1sq * o->wy;
    CA(1, 2) = (wvno2 * o->cpy - o->cqx) / rho;
    CA(1, 3) = -(twgm1 * a0pq + gammk * o->xz + wvno2 * gamm1 * o->wy) / rho;
    CA(1, 4) = (o->cpz - wvno2 * o->cqw) / rho;
    CA(1, 5) = -(two * wvno2 * a0pq + o->xz + wvno2 * wvno2 * o->wy) / rho2;
    CA(2, 1) = (gmgmk * o->cpz - gm1sq * o->cqw) * rho;
    CA(2, 2) = o->cpcq;
    CA(2, 3) = gammk * o->cpz - gamm1 * o->cqw;
    CA(2, 4) = -o->wz;
    CA(2, 5) = CA(1, 4);
    CA(4, 1) = (gm1sq * o->cpy - gmgmk * o->cqx) * rho;
    CA(4, 2) = -o->xy;
    CA(4, 3) = gamm1 * o->cpy - gammk * o->cqx;
    CA(4, 4) = CA(2, 2);
    CA(4, 5) = CA(1, 2);
    CA(5, 1) = -(two * gmgmk * gm1sq * a0pq + gmgmk * gmgmk * o->xz + gm1sq * gm1sq * o->wy) * rho2;
    CA(5, 2) = CA(4, 1);
    CA(5, 3) = -(gammk * gamm1 * twgm1 * a0pq + gam * gammk * gammk * o->xz + gamm1 * gm1sq * o->wy) * rho;
    CA(5, 4) = CA(2, 1);
    CA(5, 5) = CA(1, 1);
    const double t = -two * wvno2;
    CA(3, 1) = t * CA(5, 3);
    CA(3, 2) = t * CA(4, 3);
    CA(3, 3) = o->a0 + two * (o->cpcq - CA(1, 1));
    CA(3, 4) = t * CA(2, 3);
    CA(3, 5) = t * CA(1, 3);
#undef CA
}

/* surfdisp96.f:989-1014 */
static void normc(double ee[5], double *ex)
{
    double t1 = 0.0;
    for (int i = 0; i < 5; ++i) if (fabs(ee[i]) > t1) t1 = fabs(ee[i]);
    if (t1 < 1.e-40) t1 = 1.0;
    for (int i = 0; i < 5; ++i) { double t2 = ee[i]; t2 = t2 / t1; ee[i] = t2; }
    *ex = log(t1);
}

/* surfdisp96.f:767-865 */
static double dltar4(const model *m, double wvno, double omga)
{
    const int mmax = m->mmax;
    double e[5], ee[5], ca[5][5];
    ovr o;
    double omega = omga;
    if (omega < 1.0e-4) omega = 1.0e-4;
    const double wvno2 = wvno * wvno;
    double xka = omega / (double)m->a[mmax - 1];
    double xkb = omega / (double)m->b[mmax - 1];
    double wvnop = wvno + xka;
    double wvnom = fabs(wvno - xka);
    double ra = sqrt(wvnop * wvnom);
    wvnop = wvno + xkb;
    wvnom = fabs(wvno - xkb);
    double rb = sqrt(wvnop * wvnom);
    double t = (double)m->b[mmax - 1] / omega;
    double gammk = 2.0 * t * t;
    double gam = gammk * wvno2;
    const double gamm1 = gam - 1.0;
    double rho1 = (double)m->rho[mmax - 1];
    e[0] = rho1 * rho1 * (gamm1 * gamm1 - gam * gammk * ra * rb);
    e[1] = -rho1 * ra;
    e[2] = rho1 * (gamm1 - gammk * ra * rb);
    e[3] = rho1 * rb;
    e[4] = wvno2 - ra * rb;
    for (int k = mmax - 1; k >= m->llw; --k) {
        xka = omega / (double)m->a[k - 1];
        xkb = omega / (double)m->b[k - 1];
        t = (double)m->b[k - 1] / omega;
        gammk = 2.0 * t * t;
        gam = gammk * wvno2;
        wvnop = wvno + xka;
        wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        wvnop = wvno + xkb;
        wvnom = fabs(wvno - xkb);
        rb = sqrt(wvnop * wvnom);
        const double dpth = (double)m->d[k - 1];
        rho1 = (double)m->rho[k - 1];
        const double p = ra * dpth;
        const double q = rb * dpth;
        double w, cosp, exa;
        var(p, q, ra, rb, wvno, xka, xkb, dpth, &w, &cosp, &exa, &o);
        dnka(ca, wvno2, gam, gammk, rho1, &o);
        for (int i = 0; i < 5; ++i) {
            double cr = 0.0;
            for (int j = 0; j < 5; ++j) cr = cr + e[j] * ca[j][i];
            ee[i] = cr;
        }
        normc(ee, &exa);
        for (int i = 0; i < 5; ++i) e[i] = ee[i];
    }
    if (m->llw != 1) {
        xka = omega / (double)m->a[0];
        wvnop = wvno + xka;
        wvnom = fabs(wvno - xka);
        ra = sqrt(wvnop * wvnom);
        const double dpth = (double)m->d[0];
        rho1 = (double)m->rho[0];
        const double p = ra * dpth;
        const double znul = 1.0e-05;
        double w, cosp, exa;
        var(p, znul, ra, znul, wvno, xka, znul, dpth, &w, &cosp, &exa, &o);
        const double w0 = -rho1 * w;
        return cosp * e[0] + w0 * e[1];
    }
    return e[0];
}

static double dltar(const model *m, double wvno, double omega, int kk)
{
    return kk == 1 ? dltar1(m, wvno, omega) : dltar4(m, wvno, omega);
}

/* surfdisp96.f:670-680 */
static void half(const model *m, double c1, double c2, double *c3, double *del3, double omega, int ifunc)
{
    *c3 = 0.5 * (c1 + c2);
    const double wvno = omega / *c3;
    *del3 = dltar(m, wvno, omega, ifunc);
}

/* surfdisp96.f:551-668 */
static double nevill(const model *m, double t, double c1, double c2, double del1, double del2, int ifunc, double twopi)
{
    double x[21], y[21], c3, del3;
    const double omega = twopi / t;
    int nev, nctrl = 1, mm = 1;
    half(m, c1, c2, &c3, &del3, omega, ifunc);
    nev = 1;
    for (;;) {
        nctrl = nctrl + 1;
        if (nctrl >= 100) break;
        if (c3 < fmin(c1, c2) || c3 > fmax(c1, c2)) {
            nev = 0;
            half(m, c1, c2, &c3, &del3, omega, ifunc);
        }
        const double s13 = del1 - del3;
        const double s32 = del3 - del2;
        if (dsign1(del3) * dsign1(del1) < 0.0) { c2 = c3; del2 = del3; }
        else { c1 = c3; del1 = del3; }
        if (fabs(c1 - c2) <= 1.e-6 * c1) break;
        if (dsign1(s13) != dsign1(s32)) nev = 0;
        const double ss1 = fabs(del1);
        const double s1 = (double)0.01f * ss1;          /* single-precision literal in the reference */
        const double ss2 = fabs(del2);
        const double s2 = (double)0.01f * ss2;
        if (s1 > ss2 || s2 > ss1 || nev == 0) {
            half(m, c1, c2, &c3, &del3, omega, ifunc);
            nev = 1;
            mm = 1;
        } else {
            if (nev == 2) { x[mm + 1] = c3; y[mm + 1] = del3; }
            else { x[1] = c1; y[1] = del1; x[2] = c2; y[2] = del2; mm = 1; }
            int bad = 0;
            for (int kk = 1; kk <= mm; ++kk) {
                const int j = mm - kk + 1;
                const double denom = y[mm + 1] - y[j];
                if (fabs(denom) < 1.0e-10 * fabs(y[mm + 1])) { bad = 1; break; }
                x[j] = (-y[j] * x[j + 1] + y[mm + 1] * x[j]) / denom;
            }
            if (!bad) {
                c3 = x[1];
                const double wvno = omega / c3;
                del3 = dltar(m, wvno, omega, ifunc);
                nev = 2;
                mm = mm + 1;
                if (mm > 10) mm = 10;
            } else {
                half(m, c1, c2, &c3, &del3, omega, ifunc);
                nev = 1;
                mm = 1;
            }
        }
    }
    return c3;
}

/* surfdisp96.f:384-476; returns iret */
static int getsol(model *m, double t1, double *c1io, double clow, double dc, double cm, float betmx, int ifunc, int ifirst)
{
    const double twopi = 2.0 * 3.141592653589793;
    double c1 = *c1io, c2, del1, del2;
    double omega = twopi / t1;
    double wvno = omega / c1;
    del1 = dltar(m, wvno, omega, ifunc);
    if (ifirst == 1) m->del1st = del1;
    const double plmn = dsign1(m->del1st) * dsign1(del1);
    int idir = +1;
    if (ifirst == 1) idir = +1;
    else if (plmn >= 0.0) idir = +1;
    else idir = -1;
    for (;;) {
        if (idir > 0) c2 = c1 + dc; else c2 = c1 - dc;
        if (c2 <= clow) { idir = +1; c1 = clow; continue; }
        omega = twopi / t1;
        wvno = omega / c2;
        del2 = dltar(m, wvno, omega, ifunc);
        if (dsign1(del1) != dsign1(del2)) break;
        c1 = c2;
        del1 = del2;
        if (c1 < cm) { *c1io = c1; return -1; }
        if (c1 >= ((double)betmx + dc)) { *c1io = c1; return -1; }
    }
    const double cn = nevill(m, t1, c1, c2, del1, del2, ifunc, twopi);
    c1 = cn;
    *c1io = c1;
    if (c1 > (double)betmx) return -1;
    return 1;
}

/* surfdisp96.f:480-547 */
static void sphere(model *m, int ifunc, int iflag)
{
    const int mmax = m->mmax;
    const double ar = 6370.0;
    double dr = 0.0, r0 = ar;
    m->d[mmax - 1] = 1.0f;
    if (iflag == 0) {
        for (int i = 0; i < mmax; ++i) { m->dtp[i] = m->d[i]; m->rtp[i] = m->rho[i]; }
        for (int i = 0; i < mmax; ++i) {
            dr = dr + (double)m->d[i];
            const double r1 = ar - dr;
            const double z0 = ar * log(ar / r0);
            const double z1 = ar * log(ar / r1);
            m->d[i] = (float)(z1 - z0);
            const double tmp = (ar + ar) / (r0 + r1);
            m->a[i] = (float)((double)m->a[i] * tmp);
            m->b[i] = (float)((double)m->b[i] * tmp);
            m->btp[i] = (float)tmp;
            r0 = r1;
        }
        m->dhalf = m->d[mmax - 1];
    } else {
        m->d[mmax - 1] = m->dhalf;
        for (int i = 0; i < mmax; ++i) {
            if (ifunc == 1) {
                /* btp**(-5): integer power, binary-exponentiation order, then the reciprocal */
                const float x = m->btp[i];
                const float x2 = x * x;
                m->rho[i] = m->rtp[i] * (1.0f / (x * (x2 * x2)));
            } else if (ifunc == 2) {
                m->rho[i] = m->rtp[i] * powf(m->btp[i], -2.275f);
            }
        }
    }
    m->d[mmax - 1] = 0.0f;
}

/* surfdisp96.f:361-382, all REAL*4 */
static float gtsolh(float a, float b)
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

void dso_surfdisp96(const float *thkm, const float *vpm, const float *vsm, const float *rhom,
                    int nlayer, int iflsph, int iwave, int mode, int igr, int kmax,
                    const double *t, double *cg)
{
    model M;
    model *m = &M;
    memset(m, 0, sizeof *m);
    double c[80], cb[80];
    const int mmax = nlayer;
    m->mmax = mmax;
    for (int i = 0; i < mmax; ++i) { m->b[i] = vsm[i]; m->a[i] = vpm[i]; m->d[i] = thkm[i]; m->rho[i] = rhom[i]; }
    int idispl = 0, idispr = 0;
    if (iwave == 1) { idispl = kmax; idispr = 0; } else if (iwave == 2) { idispl = 0; idispr = kmax; }
    const float sone0 = 1.500f, ddc0 = 0.005f, h0 = 0.005f;
    m->llw = 1;
    if (m->b[0] <= 0.0f) m->llw = 2;
    const double one = 1.0e-2;
    if (iflsph == 1) sphere(m, 0, 0);
    int jmn = 1, jsol = 1;
    float betmx = -1.e20f, betmn = 1.e20f;
    for (int i = 0; i < mmax; ++i) {
        if (m->b[i] > 0.01f && m->b[i] < betmn) { betmn = m->b[i]; jmn = i + 1; jsol = 1; }
        else if (m->b[i] <= 0.01f && m->a[i] < betmn) { betmn = m->a[i]; jmn = i + 1; jsol = 0; }
        if (m->b[i] > betmx) betmx = m->b[i];
    }
    for (int ifunc = 1; ifunc <= 2; ++ifunc) {
        if (ifunc == 1 && idispl <= 0) continue;
        if (ifunc == 2 && idispr <= 0) continue;
        if (iflsph == 1) sphere(m, ifunc, 1);
        const float ddc = ddc0;
        float sone = sone0;
        const float h = h0;
        if (sone < 0.01f) sone = 2.0f;
        const double onea = (double)sone;
        float cc1;
        if (jsol == 0) cc1 = betmn;
        else cc1 = gtsolh(m->a[jmn - 1], m->b[jmn - 1]);
        cc1 = .95f * cc1;
        cc1 = .90f * cc1;
        const double cc = (double)cc1;
        double dc = (double)ddc;
        dc = fabs(dc);
        double c1 = cc;
        const double cm = cc;
        for (int i = 0; i < kmax; ++i) { cb[i] = 0.0; c[i] = 0.0; }
        int ift = 999;
        for (int iq = 1; iq <= mode; ++iq) {
            const int is = 1, ie = kmax;
            int k, failed = 0;
            for (k = is; k <= ie; ++k) {
                if (k >= ift) { failed = 1; break; }
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
                if (k == is && iq == 1) { c1 = cc; clow = cc; ifirst = 1; }
                else if (k == is && iq > 1) { c1 = c[is - 1] + one * dc; clow = c1; ifirst = 1; }
                else if (k > is && iq > 1) {
                    ifirst = 0;
                    clow = c[k - 1] + one * dc;
                    c1 = c[k - 2];
                    if (c1 < clow) c1 = clow;
                } else { ifirst = 0; c1 = c[k - 2] - onea * dc; clow = cm; }
                int iret = getsol(m, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
                if (iret == -1) { failed = 1; break; }
                c[k - 1] = c1;
                if (igr > 0) {
                    t1 = (double)t1b;
                    ifirst = 0;
                    clow = cb[k - 1] + one * dc;
                    c1 = c1 - onea * dc;
                    iret = getsol(m, t1, &c1, clow, dc, cm, betmx, ifunc, ifirst);
                    if (iret == -1) c1 = c[k - 1];
                    cb[k - 1] = c1;
                } else c1 = 0.0;
                const float cc0 = (float)c[k - 1];
                const float cc1b = (float)c1;
                if (igr == 0) cg[k - 1] = (double)cc0;
                else {
                    const float gvel = (1 / t1a - 1 / t1b) / (1 / (t1a * cc0) - 1 / (t1b * cc1b));
                    cg[k - 1] = (double)gvel;
                }
            }
            if (failed) {
                /* label 1700/1750: the reference logs a warning to unit 66 and zero-fills the rest */
                ift = k;
                for (int i = k; i <= ie; ++i) cg[i - 1] = 0.0;
            }
        }
    }
}

/* CalSurfG.f90:2352-2411 */
void dso_refine_layers(float minthk0, int mmax, const float *dep, const float *vp, const float *vs,
                       const float *rho, int *rmax, float *rdep, float *rvp, float *rvs,
                       float *rrho, float *rthk)
{
    int k = 0;
    float initdep = 0.0f;
    for (int i = 1; i <= mmax - 1; ++i) {
        const float thk = dep[i] - dep[i - 1];
        const float minthk = thk / minthk0;
        const int nsub = (int)((thk + 1.0e-4f) / minthk) + 1;
        const float newthk = thk / (float)nsub;
        for (int j = 1; j <= nsub; ++j) {
            k = k + 1;
            rthk[k - 1] = newthk;
            rdep[k - 1] = initdep + rthk[k - 1];
            initdep = rdep[k - 1];
            rvp[k - 1] = vp[i - 1] + (float)(2 * j - 1) * (vp[i] - vp[i - 1]) / (float)(2 * nsub);
            rvs[k - 1] = vs[i - 1] + (float)(2 * j - 1) * (vs[i] - vs[i - 1]) / (float)(2 * nsub);
            rrho[k - 1] = rho[i - 1] + (float)(2 * j - 1) * (rho[i] - rho[i - 1]) / (float)(2 * nsub);
        }
    }
    k = k + 1;
    rthk[k - 1] = 0.0f;
    rvp[k - 1] = vp[mmax - 1];
    rvs[k - 1] = vs[mmax - 1];
    rrho[k - 1] = rho[mmax - 1];
    rdep[k - 1] = dep[mmax - 1];
    *rmax = k;
}

static inline float q2(float x) { return x * x; }
static inline float q3(float x) { return x * (x * x); }
/* positive integer powers are plain left-to-right chains in the reference's build (checked against
 * flang: x**4 = ((x*x)*x)*x); only negative ones go through the binary method (sphere, btp**(-5)) */
static inline float q4(float x) { return ((x * x) * x) * x; }
static inline float q5(float x) { return (((x * x) * x) * x) * x; }

/* Brocher relations, CalSurfG.f90:49-53 */
static void brocher(float vs, float *vp, float *rho)
{
    const float p = 0.9409f + 2.0947f * vs - 0.8206f * q2(vs) + 0.2683f * q3(vs) - 0.0251f * q4(vs);
    *vp = p;
    *rho = 1.6612f * p - 0.4721f * q2(p) + 0.0671f * q3(p) - 0.0043f * q4(p) + 0.000106f * q5(p);
}

void dso_caldespersion(int nx, int ny, int nz, const float *vel, double *pv, int iwave, int igr,
                       int kmax, const double *t, const float *depz, float minthk)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int jj = 1; jj <= ny; ++jj) {
        float vsz[NLMAX], vpz[NLMAX], rhoz[NLMAX];
        float rdep[NLMAX], rvp[NLMAX], rvs[NLMAX], rrho[NLMAX], rthk[NLMAX];
        double cg[80];
        for (int ii = 1; ii <= nx; ++ii) {
            for (int k = 0; k < nz; ++k) {
                vsz[k] = vel[(size_t)k * nx * ny + (size_t)(jj - 1) * nx + (ii - 1)];
                brocher(vsz[k], &vpz[k], &rhoz[k]);
            }
            int rmax;
            dso_refine_layers(minthk, nz, depz, vpz, vsz, rhoz, &rmax, rdep, rvp, rvs, rrho, rthk);
            dso_surfdisp96(rthk, rvp, rvs, rrho, rmax, 1, iwave, 1, igr, kmax, t, cg);
            for (int k = 0; k < kmax; ++k) pv[(size_t)k * nx * ny + (size_t)(jj - 1) * nx + (ii - 1)] = cg[k];
        }
    }
}

void dso_depthkernel(int nx, int ny, int nz, const float *vel, double *pv, double *sen_vs,
                     double *sen_vp, double *sen_rho, int iwave, int igr, int kmax,
                     const double *t, const float *depz, float minthk)
{
    const float dln = 0.01f;
    const size_t ncol = (size_t)nx * ny;
#pragma omp parallel for schedule(dynamic, 2)
    for (int jj = 1; jj <= ny; ++jj) {
        float vsz[NLMAX], vpz[NLMAX], rhoz[NLMAX], vsm[NLMAX], vpm[NLMAX], rhom[NLMAX];
        float rdep[NLMAX], rvp[NLMAX], rvs[NLMAX], rrho[NLMAX], rthk[NLMAX];
        double cg[80], cg1[80], cg2[80];
        for (int ii = 1; ii <= nx; ++ii) {
            const size_t colidx = (size_t)(jj - 1) * nx + (ii - 1);
            for (int k = 0; k < nz; ++k) {
                vsz[k] = vel[(size_t)k * ncol + colidx];
                brocher(vsz[k], &vpz[k], &rhoz[k]);
            }
            int rmax;
            dso_refine_layers(minthk, nz, depz, vpz, vsz, rhoz, &rmax, rdep, rvp, rvs, rrho, rthk);
            dso_surfdisp96(rthk, rvp, rvs, rrho, rmax, 1, iwave, 1, igr, kmax, t, cg);
            for (int k = 0; k < kmax; ++k) pv[(size_t)k * ncol + colidx] = cg[k];
            for (int k = 0; k < nz; ++k) { vsm[k] = vsz[k]; vpm[k] = vpz[k]; rhom[k] = rhoz[k]; }
            for (int i = 0; i < nz; ++i) {
                float *arr[3] = { vsm, vpm, rhom };
                const float *base[3] = { vsz, vpz, rhoz };
                double *out[3] = { sen_vs, sen_vp, sen_rho };
                for (int q = 0; q < 3; ++q) {
                    arr[q][i] = base[q][i] - 0.5f * dln * base[q][i];
                    dso_refine_layers(minthk, nz, depz, vpm, vsm, rhom, &rmax, rdep, rvp, rvs, rrho, rthk);
                    dso_surfdisp96(rthk, rvp, rvs, rrho, rmax, 1, iwave, 1, igr, kmax, t, cg1);
                    arr[q][i] = base[q][i] + 0.5f * dln * base[q][i];
                    dso_refine_layers(minthk, nz, depz, vpm, vsm, rhom, &rmax, rdep, rvp, rvs, rrho, rthk);
                    dso_surfdisp96(rthk, rvp, rvs, rrho, rmax, 1, iwave, 1, igr, kmax, t, cg2);
                    arr[q][i] = base[q][i];
                    for (int nn = 0; nn < kmax; ++nn)
                        out[q][((size_t)i * kmax + nn) * ncol + colidx] = (cg2[nn] - cg1[nn]) / (double)(dln * base[q][i]);
                }
            }
        }
    }
}
