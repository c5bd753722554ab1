/* TEST INFRASTRUCTURE ONLY -- see dsurf_oracle.h.
 *
 * The inversion step next to the CalSurfG path, restated in C with the
 * reference's fp32 operation order: LSMR as shipped with DSurfTomo (lsmrModule.f90:36-750, single
 * precision: lsmrDataModule.f90 sets dp = selected_real_kind(4)), its BLAS subset (lsmrblas.f90) and
 * the per-iteration glue of main.f90:361-466 (residual, percentile weights, DWS, Laplacian rows).
 * Pinned bit-exactly against the reference's own objects (oracle/_ref, tests/test_oracle_vs_ref.py).
 * Compile with -ffp-contract=off and SSE math.
 */
#include "dsurf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lsmrblas.f90:247-277 (unit stride) */
float dso_dnrm2(int n, const float *x)
{
    if (n < 1) return 0.0f;
    if (n == 1) return fabsf(x[0]);
    float scale = 0.0f, ssq = 1.0f;
    for (int i = 0; i < n; ++i) {
        if (x[i] != 0.0f) {
            const float absxi = fabsf(x[i]);
            if (scale < absxi) {
                const float q = scale / absxi;
                ssq = 1.0f + ssq * (q * q);
                scale = absxi;
            } else {
                const float q = absxi / scale;
                ssq = ssq + q * q;
            }
        }
    }
    return scale * sqrtf(ssq);
}

/* lsmrblas.f90:317-359 (unit stride; element-wise, so the unrolling does not matter) */
void dso_dscal(int n, float sa, float *x)
{
    for (int i = 0; i < n; ++i) x[i] = sa * x[i];
}

/* lsmrModule.f90:686-711 */
static float d2norm(float a, float b)
{
    const float scale = fabsf(a) + fabsf(b);
    if (scale == 0.0f) return 0.0f;
    const float p = a / scale, q = b / scale;
    return scale * sqrtf(p * p + q * q);
}

/* lsmrModule.f90:36-750 without the printing (nout <= 0 in main.f90:47,107: the unit is never opened).
 * b is not modified.  Outputs like the reference's; when A'b = 0 the reference returns before setting itn,
 * here itn = 0. */
void dso_lsmr(const int *m_, const int *n_, const int *leniw, const int *lenrw, const int *iw, const float *rw,
              const float *b, const float *damp_, const float *atol_, const float *btol_, const float *conlim_,
              const int *itnlim_, const int *localSize_, float *x, int *istop, int *itn, float *normA,
              float *condA, float *normr, float *normAr, float *normx)
{
    const int m = *m_, n = *n_, itnlim = *itnlim_;
    const float damp = *damp_, atol = *atol_, btol = *btol_, conlim = *conlim_;
    int localVecs = *localSize_;
    if (m < localVecs) localVecs = m;
    if (n < localVecs) localVecs = n;                                                 /* :365 */
    float *h = calloc((size_t)n, 4), *hbar = calloc((size_t)n, 4), *u = malloc((size_t)m * 4), *v = calloc((size_t)n, 4);
    float *localV = localVecs > 0 ? malloc((size_t)n * (size_t)localVecs * 4) : NULL;
    const int one_i = 1, two_i = 2;
    const int damped = damp > 0.0f;

    memcpy(u, b, (size_t)m * 4);                                                      /* :383-385 */
    for (int i = 0; i < n; ++i) x[i] = 0.0f;
    float alpha = 0.0f;
    float beta = dso_dnrm2(m, u);
    if (beta > 0.0f) {
        dso_dscal(m, 1.0f / beta, u);
        dso_aprod(&two_i, m_, n_, v, u, leniw, lenrw, iw, rw);                        /* v = A'u */
        alpha = dso_dnrm2(n, v);
    }
    if (alpha > 0.0f) dso_dscal(n, 1.0f / alpha, v);
    *itn = 0; *istop = 0; *normA = 0.0f; *condA = 0.0f; *normx = 0.0f;
    *normr = beta;
    *normAr = alpha * beta;
    if (*normAr == 0.0f) goto done;                                                   /* :404-405 -> 800 */

    int localOrtho = 0, localPointer = 0, localVQueueFull = 0;
    if (localVecs > 0) {                                                              /* :408-413 */
        localPointer = 1; localOrtho = 1;
        memcpy(localV, v, (size_t)n * 4);
    }
    float zetabar = alpha * beta, alphabar = alpha, rho = 1.0f, rhobar = 1.0f, cbar = 1.0f, sbar = 0.0f;
    memcpy(h, v, (size_t)n * 4);
    float betadd = beta, betad = 0.0f, rhodold = 1.0f, tautildeold = 0.0f, thetatilde = 0.0f, zeta = 0.0f, d = 0.0f;
    float normA2 = alpha * alpha, maxrbar = 0.0f, minrbar = 1e+30f;
    const float normb = beta;
    float ctol = 0.0f;
    if (conlim > 0.0f) ctol = 1.0f / conlim;

    for (;;) {                                                                        /* :480 */
        *itn += 1;
        dso_dscal(m, -alpha, u);
        dso_aprod(&one_i, m_, n_, v, u, leniw, lenrw, iw, rw);                        /* u = A v - alpha u */
        beta = dso_dnrm2(m, u);
        if (beta > 0.0f) {
            dso_dscal(m, 1.0f / beta, u);
            if (localOrtho) {                                                         /* localVEnqueue, :715-727 */
                if (localPointer < localVecs) localPointer += 1;
                else { localPointer = 1; localVQueueFull = 1; }
                memcpy(localV + (size_t)(localPointer - 1) * (size_t)n, v, (size_t)n * 4);
            }
            dso_dscal(n, -beta, v);
            dso_aprod(&two_i, m_, n_, v, u, leniw, lenrw, iw, rw);                    /* v = A'u - beta v */
            if (localOrtho) {                                                         /* localVOrtho, :731-748 */
                const int lim = localVQueueFull ? localVecs : localPointer;
                for (int k = 0; k < lim; ++k) {
                    const float *lv = localV + (size_t)k * (size_t)n;
                    float dd = 0.0f;
                    for (int i = 0; i < n; ++i) dd = dd + v[i] * lv[i];               /* dot_product: in order, fp32 */
                    for (int i = 0; i < n; ++i) v[i] = v[i] - dd * lv[i];
                }
            }
            alpha = dso_dnrm2(n, v);
            if (alpha > 0.0f) dso_dscal(n, 1.0f / alpha, v);
        }
        /* rotations, :516-560 */
        const float alphahat = d2norm(alphabar, damp);
        const float chat = alphabar / alphahat, shat = damp / alphahat;
        const float rhoold = rho;
        rho = d2norm(alphahat, beta);
        const float c = alphahat / rho, s = beta / rho;
        const float thetanew = s * alpha;
        alphabar = c * alpha;
        const float rhobarold = rhobar, zetaold = zeta;
        const float thetabar = sbar * rho, rhotemp = cbar * rho;
        rhobar = d2norm(cbar * rho, thetanew);
        cbar = cbar * rho / rhobar;
        sbar = thetanew / rhobar;
        zeta = cbar * zetabar;
        zetabar = -sbar * zetabar;
        {                                                                             /* :545-547 */
            const float c1 = thetabar * rho / (rhoold * rhobarold);
            for (int i = 0; i < n; ++i) hbar[i] = h[i] - c1 * hbar[i];
            const float c2 = zeta / (rho * rhobar);
            for (int i = 0; i < n; ++i) x[i] = x[i] + c2 * hbar[i];
            const float c3 = thetanew / rho;
            for (int i = 0; i < n; ++i) h[i] = v[i] - c3 * h[i];
        }
        const float betaacute = chat * betadd, betacheck = -shat * betadd;
        const float betahat = c * betaacute;
        betadd = -s * betaacute;
        const float thetatildeold = thetatilde;
        const float rhotildeold = d2norm(rhodold, thetabar);
        const float ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
        thetatilde = stildeold * rhobar;
        rhodold = ctildeold * rhobar;
        betad = -stildeold * betad + ctildeold * betahat;
        tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
        const float taud = (zeta - thetatilde * tautildeold) / rhodold;
        d = d + betacheck * betacheck;
        {
            const float e1 = betad - taud;
            *normr = sqrtf(d + e1 * e1 + betadd * betadd);
        }
        normA2 = normA2 + beta * beta;
        *normA = sqrtf(normA2);
        normA2 = normA2 + alpha * alpha;
        maxrbar = maxrbar > rhobarold ? maxrbar : rhobarold;
        if (*itn > 1) minrbar = minrbar < rhobarold ? minrbar : rhobarold;
        *condA = (maxrbar > rhotemp ? maxrbar : rhotemp) / (minrbar < rhotemp ? minrbar : rhotemp);
        *normAr = fabsf(zetabar);
        *normx = dso_dnrm2(n, x);
        const float test1 = *normr / normb;
        const float test2 = *normAr / (*normA * *normr);
        const float test3 = 1.0f / *condA;
        const float t1 = test1 / (1.0f + *normA * *normx / normb);
        const float rtol = btol + atol * *normA * *normx / normb;
        if (*itn >= itnlim) *istop = 7;                                               /* :607-613 */
        if (1.0f + test3 <= 1.0f) *istop = 6;
        if (1.0f + test2 <= 1.0f) *istop = 5;
        if (1.0f + t1 <= 1.0f) *istop = 4;
        if (test3 <= ctol) *istop = 3;
        if (test2 <= atol) *istop = 2;
        if (test1 <= rtol) *istop = 1;
        if (*istop != 0) break;
    }
done:
    if (damped && *istop == 2) *istop = 3;                                            /* :651 */
    free(h); free(hbar); free(u); free(v); free(localV);
}

/* getpercentile.f90:1-51: heap sort of a copy, then the elements int(0.25 N) and int(0.75 N) (1-based).
 * The returned values do not depend on the sorting method (a sorted array is a sorted array). */
static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}
void dso_getpercentile(int n, const float *array, float *q25, float *q75)
{
    float *ra = malloc((size_t)n * 4);
    memcpy(ra, array, (size_t)n * 4);
    qsort(ra, (size_t)n, 4, cmp_float);
    /* idx = int(0.25*N): single-precision product truncated (getpercentile.f90:27-30) */
    const int i25 = (int)(0.25f * (float)n), i75 = (int)(0.75f * (float)n);
    *q25 = ra[i25 - 1];
    *q75 = ra[i75 - 1];
    free(ra);
}

/* main.f90:361-466: from the forward call's output (dsyn, COO matrix with nar entries) and the observations to the
 * damped, regularised system LSMR solves.  In/out: rw, iw (iw[0] is set to the final nar; iw[1..nar] rows,
 * iw[nar+1..2 nar] columns), col (extended by the regularisation rows).  Out: cbst(dall + maxvp) right-hand
 * side, datweight(dall), norm(maxvp) (DWS), *m_out, *nar_out, dws[2] = {max, mean}.  nar_in = CalSurfG's nar. */
void dso_iteration_system(int nx, int ny, int nz, int dall, int nar_in, float *rw, int *iw, int *col,
                          const float *obst, const float *dsyn, float threshold0, float weight0,
                          float *cbst, float *datweight, float *norm, int *m_out, int *nar_out, float *dws)
{
    const int maxvp = (nx - 2) * (ny - 2) * (nz - 1);
    int nar = nar_in;
    for (int i = 0; i < dall; ++i) cbst[i] = obst[i] - dsyn[i];                       /* :361-363 */
    float q25, q75;
    dso_getpercentile(dall, cbst, &q25, &q75);                                        /* :365 */
    for (int i = 0; i < dall; ++i) {                                                  /* :366-372 */
        datweight[i] = 1.0f;
        if (cbst[i] < q25 * threshold0 || cbst[i] > q75 * threshold0) { datweight[i] = 0.0f; cbst[i] = 0.0f; }
    }
    for (int i = 0; i < nar; ++i) rw[i] = rw[i] * datweight[iw[1 + i] - 1];          /* :378-380 */
    for (int i = 0; i < maxvp; ++i) norm[i] = 0.0f;                                   /* :382-385 */
    for (int i = 0; i < nar; ++i) norm[col[i] - 1] = norm[col[i] - 1] + fabsf(rw[i]);
    float averdws = 0.0f, maxnorm = 0.0f;                                             /* :386-392 */
    for (int i = 0; i < maxvp; ++i) { averdws = averdws + norm[i]; if (norm[i] > maxnorm) maxnorm = norm[i]; }
    averdws = averdws / (float)maxvp;
    dws[0] = maxnorm; dws[1] = averdws;
    const float weight = weight0;                                                     /* :414 */
    int count3 = 0;
    const int nvz = ny - 2, nvx = nx - 2;
    for (int k = 1; k <= nz - 1; ++k)                                                 /* :420-457 */
        for (int j = 1; j <= nvz; ++j)
            for (int i = 1; i <= nvx; ++i) {
                const int c0 = (k - 1) * nvz * nvx + (j - 1) * nvx + i;
                count3 += 1;
                if (i == 1 || i == nvx || j == 1 || j == nvz || k == 1 || k == nz - 1) {
                    col[nar] = c0; rw[nar] = 2.0f * weight; iw[1 + nar] = dall + count3;
                    cbst[dall + count3 - 1] = 0.0f;
                    nar += 1;
                } else {
                    const int cc[7] = { c0, c0 - 1, c0 + 1, (k - 1) * nvz * nvx + (j - 2) * nvx + i, (k - 1) * nvz * nvx + j * nvx + i,
                                        (k - 2) * nvz * nvx + (j - 1) * nvx + i, k * nvz * nvx + (j - 1) * nvx + i };
                    for (int q = 0; q < 7; ++q) {
                        col[nar + q] = cc[q];
                        rw[nar + q] = q == 0 ? 6.0f * weight : -1.0f * weight;
                        iw[1 + nar + q] = dall + count3;
                    }
                    cbst[dall + count3 - 1] = 0.0f;
                    nar += 7;
                }
            }
    *m_out = dall + count3;                                                           /* :458-459 */
    iw[0] = nar;                                                                      /* :461-464 */
    for (int i = 0; i < nar; ++i) iw[1 + nar + i] = col[i];
    *nar_out = nar;
}

/* main.f90:520-535: clamp the update to +-0.5 and the model to [Minvel, Maxvel]; vsf(nx, ny, nz) column-major */
void dso_model_update(int nx, int ny, int nz, float *dv, float *vsf, float minvel, float maxvel)
{
    for (int k = 1; k <= nz - 1; ++k)
        for (int j = 1; j <= ny - 2; ++j)
            for (int i = 1; i <= nx - 2; ++i) {
                float *d = &dv[(k - 1) * (nx - 2) * (ny - 2) + (j - 1) * (nx - 2) + i - 1];
                if (*d >= 0.500f) *d = 0.500f;
                if (*d <= -0.500f) *d = -0.500f;
                float *vv = &vsf[(size_t)(k - 1) * nx * ny + (size_t)j * nx + i];      /* vsf(i+1, j+1, k) */
                *vv = *vv + *d;
                if (*vv < minvel) *vv = minvel;
                if (*vv > maxvel) *vv = maxvel;
            }
}
