/* TEST INFRASTRUCTURE ONLY -- see dsurf_oracle.h.
 *
 * Eikonal / ray / Frechet side of the CalSurfG hot path, restated in C with the reference's
 * fp32 operation order (compile with -ffp-contract=off, SSE math).  Integer powers follow what the
 * reference's compiler emits (positive powers are left-to-right chains, x**3 = (x*x)*x); this is
 * part of what tests/test_oracle_vs_ref.py pins bitwise.
 */
#include "dsurf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const float PI_F = 3.1415926535898f; /* CalSurfG.f90:196 */

static inline float p2(float x) { return x * x; }
static inline float p3(float x) { return x * (x * x); }

/* 1-based (iz, ix) access into a column-major (ld, *) field */
#define AT(a, ld, iz, ix) ((a)[(size_t)((ix) - 1) * (size_t)(ld) + (size_t)((iz) - 1)])

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                   */

void dso_grid_init(dso_grid *g, int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd)
{
    g->nx = nx; g->ny = ny;
    g->nvx = nx - 2; g->nvz = ny - 2;
    g->gdx = gd; g->gdz = gd;
    g->sgdl = 8; g->sgs = 8;
    g->earth = 6371.0f;
    g->goxd = goxd; g->gozd = gozd; g->dvxd = dvxd; g->dvzd = dvzd;
    g->dvx = dvxd * PI_F / 180.0f;
    g->dvz = dvzd * PI_F / 180.0f;
    g->gox = (90.0f - goxd) * PI_F / 180.0f;
    g->goz = gozd * PI_F / 180.0f;
    g->nnx = (g->nvx - 1) * gd + 1;
    g->nnz = (g->nvz - 1) * gd + 1;
    g->dnx = g->dvx / (float)gd;
    g->dnz = g->dvz / (float)gd;
}

/* cubic B-spline basis at parameter u; CalSurfG.f90:1509-1512 */
static void bspl4(float u, float w[4])
{
    w[0] = p3(1.0f - u) / 6.0f;
    w[1] = (4.0f - 6.0f * p2(u) + 3.0f * p3(u)) / 6.0f;
    w[2] = (1.0f + 3.0f * u + 3.0f * p2(u) - 3.0f * p3(u)) / 6.0f;
    w[3] = p3(u) / 6.0f;
}

/* velv(i,j) = real(pv(i*(nvx+2)+j+1)), i = 0..nvz+1 (z), j = 0..nvx+1 (x); :1487-1494 */
static float *load_velv(const dso_grid *g, const double *pv)
{
    int n = g->nx * g->ny;
    float *v = (float *)malloc(sizeof(float) * (size_t)n);
    for (int k = 0; k < n; ++k) v[k] = (float)pv[k];
    return v; /* velv(i,j) == v[i*nx + j] */
}

void dso_gridder(const dso_grid *g, const double *pv, float *veln)
{
    const int nx = g->nx, gdx = g->gdx, gdz = g->gdz, nvx = g->nvx, nvz = g->nvz, ld = g->nnz;
    float *velv = load_velv(g, pv);
    float (*ui)[4] = malloc(sizeof(float[4]) * (size_t)(gdx + 1));
    float (*vi)[4] = malloc(sizeof(float[4]) * (size_t)(gdz + 1));
    for (int i = 1; i <= gdx + 1; ++i) { float u = (float)gdx; u = (float)(i - 1) / u; bspl4(u, ui[i - 1]); }
    for (int i = 1; i <= gdz + 1; ++i) { float u = (float)gdz; u = (float)(i - 1) / u; bspl4(u, vi[i - 1]); }
    for (int i = 1; i <= nvz - 1; ++i) {
        int conz = (i == nvz - 1) ? gdz + 1 : gdz;
        for (int j = 1; j <= nvx - 1; ++j) {
            int conx = (j == nvx - 1) ? gdx + 1 : gdx;
            for (int l = 1; l <= conz; ++l) {
                int stz = gdz * (i - 1) + l;
                for (int m = 1; m <= conx; ++m) {
                    int stx = gdx * (j - 1) + m;
                    float sumi = 0.0f;
                    for (int i1 = 1; i1 <= 4; ++i1) {
                        float sumj = 0.0f;
                        for (int j1 = 1; j1 <= 4; ++j1)
                            sumj = sumj + ui[m - 1][j1 - 1] * velv[(i - 2 + i1) * nx + (j - 2 + j1)];
                        sumi = sumi + vi[l - 1][i1 - 1] * sumj;
                    }
                    AT(veln, ld, stz, stx) = sumi;
                }
            }
        }
    }
    free(ui); free(vi); free(velv);
}

int dso_source_box(const dso_grid *g, float x, float z, dso_box *b)
{
    int isx = (int)((x - g->gox) / g->dnx) + 1;
    int isz = (int)((z - g->goz) / g->dnz) + 1;
    if (isx < 1 || isx > g->nnx || isz < 1 || isz > g->nnz) return -1;
    if (isx == g->nnx) isx -= 1;
    if (isz == g->nnz) isz -= 1;
    b->vnl = isx - g->sgs; if (b->vnl < 1) b->vnl = 1;
    b->vnr = isx + g->sgs; if (b->vnr > g->nnx) b->vnr = g->nnx;
    b->vnt = isz - g->sgs; if (b->vnt < 1) b->vnt = 1;
    b->vnb = isz + g->sgs; if (b->vnb > g->nnz) b->vnb = g->nnz;
    b->nnx = (b->vnr - b->vnl) * g->sgdl + 1;
    b->nnz = (b->vnb - b->vnt) * g->sgdl + 1;
    b->dnx = g->dvx / (float)(g->gdx * g->sgdl);
    b->dnz = g->dvz / (float)(g->gdz * g->sgdl);
    b->gox = g->gox + g->dnx * (float)(b->vnl - 1);
    b->goz = g->goz + g->dnz * (float)(b->vnt - 1);
    return 0;
}

void dso_bsplrefine(const dso_grid *g, const double *pv, const dso_box *b, float *velr)
{
    const int nx = g->nx, gdx = g->gdx, gdz = g->gdz, sgdl = g->sgdl, nvx = g->nvx, nvz = g->nvz;
    const int nrxr = gdx * sgdl, nrzr = gdz * sgdl, ld = b->nnz;
    float *velv = load_velv(g, pv);
    float (*ub)[4] = malloc(sizeof(float[4]) * (size_t)(nrxr + 1));
    float (*vb)[4] = malloc(sizeof(float[4]) * (size_t)(nrzr + 1));
    for (int j = 1; j <= nrxr + 1; ++j) { float u = (float)nrxr; u = (float)(j - 1) / u; bspl4(u, ub[j - 1]); }
    for (int i = 1; i <= nrzr + 1; ++i) { float v = (float)nrzr; v = (float)(i - 1) / v; bspl4(v, vb[i - 1]); }
    const int origx = (b->vnl - 1) * sgdl + 1, origz = (b->vnt - 1) * sgdl + 1;
    for (int i = 1; i <= nvz - 1; ++i) {
        int conz = (i == nvz - 1) ? nrzr + 1 : nrzr;
        /* cheap rejection of vertex cells that cannot touch the box (no effect on results) */
        if (gdz * (i - 1) + (conz - 1) / sgdl + 1 < b->vnt || gdz * (i - 1) + 1 > b->vnb) continue;
        for (int j = 1; j <= nvx - 1; ++j) {
            int conx = (j == nvx - 1) ? nrxr + 1 : nrxr;
            if (gdx * (j - 1) + (conx - 1) / sgdl + 1 < b->vnl || gdx * (j - 1) + 1 > b->vnr) continue;
            for (int k = 1; k <= conz; ++k) {
                int st1 = gdz * (i - 1) + (k - 1) / sgdl + 1;
                if (st1 < b->vnt || st1 > b->vnb) continue;
                st1 = nrzr * (i - 1) + k;
                for (int l = 1; l <= conx; ++l) {
                    int st2 = gdx * (j - 1) + (l - 1) / sgdl + 1;
                    if (st2 < b->vnl || st2 > b->vnr) continue;
                    st2 = nrxr * (j - 1) + l;
                    float sum[4];
                    for (int i1 = 1; i1 <= 4; ++i1) {
                        float s = 0.0f;
                        for (int j1 = 1; j1 <= 4; ++j1)
                            s = s + ub[l - 1][j1 - 1] * velv[(i - 2 + i1) * nx + (j - 2 + j1)];
                        sum[i1 - 1] = vb[k - 1][i1 - 1] * s;
                    }
                    int idm1 = st1 - origz + 1, idm2 = st2 - origx + 1;
                    if (idm1 < 1 || idm1 > b->nnz) continue;
                    if (idm2 < 1 || idm2 > b->nnx) continue;
                    AT(velr, ld, idm1, idm2) = sum[0] + sum[1] + sum[2] + sum[3];
                }
            }
        }
    }
    free(ub); free(vb); free(velv);
}

/* ------------------------------------------------------------------------------------------ */
/* FMM state: module globalp + traveltime, CalSurfG.f90:181-271                               */

typedef struct {
    int nnx, nnz, ld;
    float gox, goz, dnx, dnz, earth;
    const float *veln;
    float *ttn;
    int *nsts;           /* -1 far, 0 alive, >0 heap slot */
    int ntr;             /* heap size */
    int *hx, *hz;        /* heap back-pointers, 1-based */
    /* refined-stage exit test operands (:396-407) */
    int vnl, vnr, vnt, vnb;
} fmm;

#define TT(f, iz, ix) AT((f)->ttn, (f)->ld, iz, ix)
#define ST(f, iz, ix) AT((f)->nsts, (f)->ld, iz, ix)
#define VL(f, iz, ix) AT((f)->veln, (f)->ld, iz, ix)

/* fouds2 core on explicit alive predicate; shared by the FMM and by the masked evaluator */
typedef int (*alive_fn)(const void *ctx, int iz, int ix);

static float fouds2_eval(int nnx, int nnz, int ld, float gox, float dnx, float dnz, float earth,
                         const float *veln, const float *ttn, alive_fn alive, const void *ctx,
                         int iz, int ix)
{
    int tsw1 = 0;
    float travm = 0.0f;
    const float slown = 1.0f / AT(veln, ld, iz, ix);
    const float ri = earth;
    const float risti = ri * sinf(gox + (float)(ix - 1) * dnx);
    for (int j = ix - 1; j <= ix + 1; j += 2) {
        if (j < 1 || j > nnx) continue;
        int swj = -1, j2;
        if (j == ix - 1) { j2 = j - 1; if (j2 >= 1) { if (alive(ctx, iz, j2)) swj = 0; } }
        else { j2 = j + 1; if (j2 <= nnx) { if (alive(ctx, iz, j2)) swj = 0; } }
        const int aj = alive(ctx, iz, j);
        if (aj && swj == 0) {
            swj = -1;
            if (AT(ttn, ld, iz, j) > AT(ttn, ld, iz, j2)) swj = 0;
        } else swj = -1;
        for (int k = iz - 1; k <= iz + 1; k += 2) {
            if (k < 1 || k > nnz) continue;
            int swk = -1, k2;
            if (k == iz - 1) { k2 = k - 1; if (k2 >= 1) { if (alive(ctx, k2, ix)) swk = 0; } }
            else { k2 = k + 1; if (k2 <= nnz) { if (alive(ctx, k2, ix)) swk = 0; } }
            const int ak = alive(ctx, k, ix);
            if (ak && swk == 0) {
                swk = -1;
                if (AT(ttn, ld, k, ix) > AT(ttn, ld, k2, ix)) swk = 0;
            } else swk = -1;
            int swsol = 0;
            float a = 0, b = 0, c = 0, u, v, em, tref = 0, tdiv = 1;
            if (swj == 0) {
                swsol = 1;
                if (swk == 0) {
                    u = 2.0f * ri * dnx;
                    v = 2.0f * risti * dnz;
                    em = 4.0f * AT(ttn, ld, iz, j) - AT(ttn, ld, iz, j2) - 4.0f * AT(ttn, ld, k, ix);
                    em = em + AT(ttn, ld, k2, ix);
                    a = p2(v) + p2(u);
                    b = 2.0f * em * p2(u);
                    c = p2(u) * (p2(em) - p2(slown) * p2(v));
                    tref = 4.0f * AT(ttn, ld, iz, j) - AT(ttn, ld, iz, j2);
                    tdiv = 3.0f;
                } else if (ak) {
                    u = risti * dnz;
                    v = 2.0f * ri * dnx;
                    em = 3.0f * AT(ttn, ld, k, ix) - 4.0f * AT(ttn, ld, iz, j) + AT(ttn, ld, iz, j2);
                    a = p2(v) + 9.0f * p2(u);
                    b = 6.0f * em * p2(u);
                    c = p2(u) * (p2(em) - p2(slown) * p2(v));
                    tref = AT(ttn, ld, k, ix);
                    tdiv = 1.0f;
                } else {
                    u = 2.0f * ri * dnx;
                    a = 1.0f;
                    b = 0.0f;
                    c = -(p2(u) * p2(slown));
                    tref = 4.0f * AT(ttn, ld, iz, j) - AT(ttn, ld, iz, j2);
                    tdiv = 3.0f;
                }
            } else if (aj) {
                swsol = 1;
                if (swk == 0) {
                    u = ri * dnx;
                    v = 2.0f * risti * dnz;
                    em = 3.0f * AT(ttn, ld, iz, j) - 4.0f * AT(ttn, ld, k, ix) + AT(ttn, ld, k2, ix);
                    a = p2(v) + 9.0f * p2(u);
                    b = 6.0f * em * p2(u);
                    c = p2(u) * (p2(em) - p2(v) * p2(slown));
                    tref = AT(ttn, ld, iz, j);
                    tdiv = 1.0f;
                } else if (ak) {
                    u = ri * dnx;
                    v = risti * dnz;
                    em = AT(ttn, ld, k, ix) - AT(ttn, ld, iz, j);
                    a = p2(u) + p2(v);
                    b = -(2.0f * p2(u) * em);
                    c = p2(u) * (p2(em) - p2(v) * p2(slown));
                    tref = AT(ttn, ld, iz, j);
                    tdiv = 1.0f;
                } else {
                    a = 1.0f;
                    b = 0.0f;
                    c = -(p2(slown) * p2(ri) * p2(dnx));
                    tref = AT(ttn, ld, iz, j);
                    tdiv = 1.0f;
                }
            } else {
                if (swk == 0) {
                    swsol = 1;
                    u = 2.0f * risti * dnz;
                    a = 1.0f;
                    b = 0.0f;
                    c = -(p2(u) * p2(slown));
                    tref = 4.0f * AT(ttn, ld, k, ix) - AT(ttn, ld, k2, ix);
                    tdiv = 3.0f;
                } else if (ak) {
                    swsol = 1;
                    a = 1.0f;
                    b = 0.0f;
                    c = -(p2(slown) * p2(risti) * p2(dnz));
                    tref = AT(ttn, ld, k, ix);
                    tdiv = 1.0f;
                }
            }
            if (swsol) {
                float rd1 = p2(b) - 4.0f * a * c;
                if (rd1 < 0.0f) rd1 = 0.0f;
                float tdsh = (-b + sqrtf(rd1)) / (2.0f * a);
                float trav = (tref + tdsh) / tdiv;
                if (tsw1) travm = (trav < travm) ? trav : travm;
                else { travm = trav; tsw1 = 1; }
            }
        }
    }
    return travm;
}

static int alive_nsts(const void *ctx, int iz, int ix)
{
    const fmm *f = (const fmm *)ctx;
    return ST(f, iz, ix) == 0;
}

static void fouds2(fmm *f, int iz, int ix)
{
    TT(f, iz, ix) = fouds2_eval(f->nnx, f->nnz, f->ld, f->gox, f->dnx, f->dnz, f->earth, f->veln,
                                f->ttn, alive_nsts, f, iz, ix);
}

typedef struct { const unsigned char *m; int ld; } mask_ctx;
static int alive_mask(const void *ctx, int iz, int ix)
{
    const mask_ctx *c = (const mask_ctx *)ctx;
    return AT(c->m, c->ld, iz, ix) != 0;
}

float dso_fouds2_masked(int nnx, int nnz, int ld, float gox, float dnx, float dnz, float earth,
                        const float *veln, const float *ttn, const unsigned char *alive, int iz, int ix)
{
    mask_ctx c = { alive, ld };
    return fouds2_eval(nnx, nnz, ld, gox, dnx, dnz, earth, veln, ttn, alive_mask, &c, iz, ix);
}

/* binary min-heap keyed on ttn, positions mirrored in nsts; CalSurfG.f90:768-921 */
static inline float hkey(const fmm *f, int p) { return TT(f, f->hz[p], f->hx[p]); }

static inline void hswap(fmm *f, int p, int q)
{
    int tx = f->hx[p], tz = f->hz[p];
    f->hx[p] = f->hx[q]; f->hz[p] = f->hz[q];
    f->hx[q] = tx; f->hz[q] = tz;
}

/* Diagnostic only (DSO_TIE_STATS=1): what decides which of two neighbouring tree entries with bit-equal keys the
 * reference pops first?  Counted per exact tie between the popped node and a neighbour still in the tree. */
static long g_tie[8];          /* [0] ties, [1] popped one was inserted earlier, [2..5] popped one lies at x-, x+, z-, z+ of the other */
static int *g_ins = NULL;      /* insertion number per node */
static long g_ins_n = 0, g_ins_cap = 0;
void dso_tie_stats(long *out, int reset) { for (int i = 0; i < 8; ++i) { out[i] = g_tie[i]; if (reset) g_tie[i] = 0; } }

static void sift_up(fmm *f, int iz, int ix, int tpc)
{
    int tpp = tpc / 2;
    while (tpp > 0) {
        if (TT(f, iz, ix) < hkey(f, tpp)) {
            ST(f, iz, ix) = tpp;
            ST(f, f->hz[tpp], f->hx[tpp]) = tpc;
            hswap(f, tpc, tpp);
            tpc = tpp;
            tpp = tpc / 2;
        } else tpp = 0;
    }
}

static void addtree(fmm *f, int iz, int ix)
{
    if (g_ins) g_ins[(size_t)(ix - 1) * f->ld + (iz - 1)] = (int)(++g_ins_n);
    f->ntr += 1;
    ST(f, iz, ix) = f->ntr;
    f->hx[f->ntr] = ix;
    f->hz[f->ntr] = iz;
    sift_up(f, iz, ix, f->ntr);
}

/* Diagnostic only (never used for parity): DSO_VALID_HEAP=1 in the environment makes updtree restore the heap
 * property when a key was RAISED as well.  The reference does not (below), so its march can pop nodes out of
 * order; comparing the two tells which differences of the product come from that. */
static int valid_heap_mode(void)
{
    static int mode = -1;
    if (mode < 0) { const char *e = getenv("DSO_VALID_HEAP"); mode = (e && e[0] == '1') ? 1 : 0; }
    return mode;
}

static void sift_down_from(fmm *f, int tpp)
{
    for (;;) {
        int tpc = 2 * tpp;
        if (tpc > f->ntr) break;
        if (tpc < f->ntr && hkey(f, tpc) > hkey(f, tpc + 1)) tpc += 1;
        if (!(hkey(f, tpc) < hkey(f, tpp))) break;
        ST(f, f->hz[tpp], f->hx[tpp]) = tpc;
        ST(f, f->hz[tpc], f->hx[tpc]) = tpp;
        hswap(f, tpc, tpp);
        tpp = tpc;
    }
}

/* only ever moves an entry towards the root, also when its key was raised (:894-921) */
static long g_heap_violations = 0;     /* diagnostic counter: updates that left a parent above a smaller child */
long dso_heap_violations(int reset) { const long v = g_heap_violations; if (reset) g_heap_violations = 0; return v; }

static void updtree(fmm *f, int iz, int ix)
{
    sift_up(f, iz, ix, ST(f, iz, ix));
    if (valid_heap_mode()) { sift_down_from(f, ST(f, iz, ix)); return; }
    const int p = ST(f, iz, ix), c = 2 * p;
    if ((c <= f->ntr && hkey(f, c) < hkey(f, p)) || (c + 1 <= f->ntr && hkey(f, c + 1) < hkey(f, p))) {
        g_heap_violations += 1;
        if (getenv("DSO_TRACE_HEAP") && g_heap_violations <= 12)
            fprintf(stderr, "heap violation %ld: node ix %d iz %d key %.7f raised above a child (heap size %d, position %d)\n", g_heap_violations, ix - 1, iz - 1, TT(f, iz, ix), f->ntr, p);
    }
}

/* Diagnostic only (never used for parity): DSO_TIE_POLICY=1 in the environment makes the sift-down of downtree prefer the RIGHT child
 * when the two children carry bit-equal keys (the reference prefers the left one, `>` below).  Both are valid binary heaps and both
 * are valid Fast Marching orders; they differ only in which of two exactly tied narrow-band nodes is accepted first.  Comparing the
 * two fields measures how far the reference's own answer depends on that accident (tests/tools/tie_sensitivity.py). */
static int tie_policy(void)
{
    static int mode = -1;
    if (mode < 0) { const char *e = getenv("DSO_TIE_POLICY"); mode = (e && e[0] == '1') ? 1 : 0; }
    return mode;
}

static void downtree(fmm *f)
{
    if (f->ntr == 1) { f->ntr = 0; return; }
    ST(f, f->hz[f->ntr], f->hx[f->ntr]) = 1;
    f->hx[1] = f->hx[f->ntr]; f->hz[1] = f->hz[f->ntr];
    f->ntr -= 1;
    int tpp = 1, tpc = 2;
    while (tpc < f->ntr) {
        if (hkey(f, tpc) > hkey(f, tpc + 1) || (tie_policy() && hkey(f, tpc) == hkey(f, tpc + 1))) tpc += 1;
        if (hkey(f, tpc) < hkey(f, tpp)) {
            ST(f, f->hz[tpp], f->hx[tpp]) = tpc;
            ST(f, f->hz[tpc], f->hx[tpc]) = tpp;
            hswap(f, tpc, tpp);
            tpp = tpc;
            tpc = 2 * tpp;
        } else tpc = f->ntr + 1;
    }
    if (tpc == f->ntr) {
        if (hkey(f, tpc) < hkey(f, tpp)) {
            ST(f, f->hz[tpp], f->hx[tpp]) = tpc;
            ST(f, f->hz[tpc], f->hx[tpc]) = tpp;
            hswap(f, tpc, tpp);
        }
    }
}

/* bilinear, CalSurfG.f90:2328-2349; nv[i][j]: i = x offset, j = z offset */
static float bilinear(const float nv[2][2], float dnx, float dnz, float dsx, float dsz)
{
    float biv = 0.0f;
    for (int i = 1; i <= 2; ++i)
        for (int j = 1; j <= 2; ++j) {
            float produ = (1.0f - fabsf(((float)(i - 1) * dnx - dsx) / dnx)) *
                          (1.0f - fabsf(((float)(j - 1) * dnz - dsz) / dnz));
            biv = biv + nv[i - 1][j - 1] * produ;
        }
    return biv;
}

/* travel, CalSurfG.f90:288-487. urg: 0 plain, 1 refined stage (early exit), 2 continue from nsts>0 */
static int travel(fmm *f, float scx, float scz, int urg)
{
    if (getenv("DSO_TIE_STATS")) {
        const long need = (long)f->ld * f->nnx;
        if (need > g_ins_cap) { free(g_ins); g_ins = (int *)calloc((size_t)need, sizeof(int)); g_ins_cap = need; }
    }
    int isx = (int)((scx - f->gox) / f->dnx) + 1;
    int isz = (int)((scz - f->goz) / f->dnz) + 1;
    if (isx < 1 || isx > f->nnx || isz < 1 || isz > f->nnz) return -1;
    if (isx == f->nnx) isx -= 1;
    if (isz == f->nnz) isz -= 1;
    f->ntr = 0;
    if (urg == 2) {
        for (int i = 1; i <= f->nnx; ++i)
            for (int j = 1; j <= f->nnz; ++j)
                if (ST(f, j, i) > 0) addtree(f, j, i);
    } else {
        for (int i = 1; i <= f->nnx; ++i)
            for (int j = 1; j <= f->nnz; ++j) ST(f, j, i) = -1;
        float vss[2][2];
        for (int i = 1; i <= 2; ++i)
            for (int j = 1; j <= 2; ++j) vss[i - 1][j - 1] = VL(f, isz - 1 + j, isx - 1 + i);
        float dsx = (scx - f->gox) - (float)(isx - 1) * f->dnx;
        float dsz = (scz - f->goz) - (float)(isz - 1) * f->dnz;
        float vsrc = bilinear(vss, f->dnx, f->dnz, dsx, dsz);
        for (int i = 1; i <= 2; ++i)
            for (int j = 1; j <= 2; ++j) {
                /* note: distances in radians, not km (:371) */
                float ds = sqrtf(p2(dsx - (float)(i - 1) * f->dnx) + p2(dsz - (float)(j - 1) * f->dnz));
                TT(f, isz - 1 + j, isx - 1 + i) = 2.0f * ds / (vss[i - 1][j - 1] + vsrc);
                addtree(f, isz - 1 + j, isx - 1 + i);
            }
    }
    while (f->ntr > 0) {
        int ix, iz;
        if (urg == 1) {
            ix = f->hx[1]; iz = f->hz[1];
            int swrg = 0;
            /* literal test of :396-407: vnr/vnb (coarse indices) against the *refined* nnx/nnz */
            if (ix == 1) { if (f->vnl != 1) swrg = 1; }
            if (ix == f->nnx) { if (f->vnr != f->nnx) swrg = 1; }
            if (iz == 1) { if (f->vnt != 1) swrg = 1; }
            if (iz == f->nnz) { if (f->vnb != f->nnz) swrg = 1; }
            if (swrg) { ST(f, iz, ix) = 0; break; }
        }
        ix = f->hx[1]; iz = f->hz[1];
        if (urg == 2 && getenv("DSO_TRACE_POPS")) {      /* diagnostic: the first accepts of the coarse stage */
            static int shown = 0;
            if (shown < atoi(getenv("DSO_TRACE_POPS"))) { ++shown; fprintf(stderr, "pop %d: ix %d iz %d T %.7f (heap %d)\n", shown, ix - 1, iz - 1, TT(f, iz, ix), f->ntr); }
        }
        if (g_ins && urg != 1) {
            const float key = TT(f, iz, ix);
            const int nx4[4] = { ix - 1, ix + 1, ix, ix }, nz4[4] = { iz, iz, iz - 1, iz + 1 };
            for (int q = 0; q < 4; ++q) {
                if (nx4[q] < 1 || nx4[q] > f->nnx || nz4[q] < 1 || nz4[q] > f->nnz) continue;
                if (ST(f, nz4[q], nx4[q]) <= 0 || TT(f, nz4[q], nx4[q]) != key) continue;
                g_tie[0] += 1;
                if (g_ins[(size_t)(ix - 1) * f->ld + (iz - 1)] < g_ins[(size_t)(nx4[q] - 1) * f->ld + (nz4[q] - 1)]) g_tie[1] += 1;
                g_tie[2 + (q ^ 1)] += 1;       /* the popped node seen from the neighbour */
            }
        }
        ST(f, iz, ix) = 0;
        downtree(f);
        for (int i = ix - 1; i <= ix + 1; i += 2) {
            if (i >= 1 && i <= f->nnx) {
                if (ST(f, iz, i) == -1) { fouds2(f, iz, i); addtree(f, iz, i); }
                else if (ST(f, iz, i) > 0) { fouds2(f, iz, i); updtree(f, iz, i); }
            }
        }
        for (int i = iz - 1; i <= iz + 1; i += 2) {
            if (i >= 1 && i <= f->nnz) {
                if (ST(f, i, ix) == -1) { fouds2(f, i, ix); addtree(f, i, ix); }
                else if (ST(f, i, ix) > 0) { fouds2(f, i, ix); updtree(f, i, ix); }
            }
        }
    }
    return 0;
}

static void fmm_alloc_heap(fmm *f, size_t nodes)
{
    /* the reference sizes the tree at snb*nnx*nnz and never checks; allocate the safe bound */
    f->hx = (int *)malloc(sizeof(int) * (nodes + 2));
    f->hz = (int *)malloc(sizeof(int) * (nodes + 2));
}

int dso_travel_plain(int nnx, int nnz, float gox, float goz, float dnx, float dnz, float earth,
                     const float *veln, float x, float z, float *ttn)
{
    fmm f;
    memset(&f, 0, sizeof f);
    f.nnx = nnx; f.nnz = nnz; f.ld = nnz; f.gox = gox; f.goz = goz; f.dnx = dnx; f.dnz = dnz;
    f.earth = earth; f.veln = veln; f.ttn = ttn;
    f.nsts = (int *)malloc(sizeof(int) * (size_t)nnx * (size_t)nnz);
    fmm_alloc_heap(&f, (size_t)nnx * (size_t)nnz);
    int rc = travel(&f, x, z, 0);
    free(f.nsts); free(f.hx); free(f.hz);
    return rc;
}

int dso_solve_source(const dso_grid *g, const double *pv, const float *veln, float x, float z,
                     dso_box *b, float *ttn, float *ttnr_out, int *nstsr_out, float *inj_t, int *inj_s)
{
    if (dso_source_box(g, x, z, b) != 0) return -1;
    const size_t nr = (size_t)b->nnx * (size_t)b->nnz;
    const size_t nc = (size_t)g->nnx * (size_t)g->nnz;
    float *velr = (float *)malloc(sizeof(float) * nr);
    float *ttr = (float *)malloc(sizeof(float) * nr);
    int *str = (int *)malloc(sizeof(int) * nr);
    dso_bsplrefine(g, pv, b, velr);

    fmm f;
    memset(&f, 0, sizeof f);
    fmm_alloc_heap(&f, nr > nc ? nr : nc);
    /* refined stage, travel(urg=1) */
    f.nnx = b->nnx; f.nnz = b->nnz; f.ld = b->nnz;
    f.gox = b->gox; f.goz = b->goz; f.dnx = b->dnx; f.dnz = b->dnz; f.earth = g->earth;
    f.veln = velr; f.ttn = ttr; f.nsts = str;
    f.vnl = b->vnl; f.vnr = b->vnr; f.vnt = b->vnt; f.vnb = b->vnb;
    for (size_t k = 0; k < nr; ++k) ttr[k] = 0.0f;
    int rc = travel(&f, x, z, 1);
    if (rc != 0) { free(velr); free(ttr); free(str); free(f.hx); free(f.hz); return rc; }
    if (ttnr_out) memcpy(ttnr_out, ttr, sizeof(float) * nr);
    if (nstsr_out) memcpy(nstsr_out, str, sizeof(int) * nr);

    /* map every sgdl-th refined node onto the coarse grid (:1293-1303) */
    int *stc = (int *)malloc(sizeof(int) * nc);
    for (size_t k = 0; k < nc; ++k) { stc[k] = -1; ttn[k] = 0.0f; }
    const int ldc = g->nnz;
    for (int k = 1; k <= b->nnz; k += g->sgdl) {
        int idm1 = b->vnt + (k - 1) / g->sgdl;
        for (int l = 1; l <= b->nnx; l += g->sgdl) {
            int idm2 = b->vnl + (l - 1) / g->sgdl;
            AT(stc, ldc, idm1, idm2) = AT(str, b->nnz, k, l);
            if (AT(stc, ldc, idm1, idm2) >= 0) AT(ttn, ldc, idm1, idm2) = AT(ttr, b->nnz, k, l);
        }
    }
    /* alive nodes that touch a far node re-enter the narrow band (:1332-1349) */
    for (int k = 1; k <= g->nnx; ++k)
        for (int l = 1; l <= g->nnz; ++l)
            if (AT(stc, ldc, l, k) == 0) {
                if (l - 1 >= 1) { if (AT(stc, ldc, l - 1, k) == -1) AT(stc, ldc, l, k) = 1; }
                if (l + 1 <= g->nnz) { if (AT(stc, ldc, l + 1, k) == -1) AT(stc, ldc, l, k) = 1; }
                if (k - 1 >= 1) { if (AT(stc, ldc, l, k - 1) == -1) AT(stc, ldc, l, k) = 1; }
                if (k + 1 <= g->nnx) { if (AT(stc, ldc, l, k + 1) == -1) AT(stc, ldc, l, k) = 1; }
            }
    if (inj_t) memcpy(inj_t, ttn, sizeof(float) * nc);
    if (inj_s) memcpy(inj_s, stc, sizeof(int) * nc);

    /* coarse stage, travel(urg=2) */
    f.nnx = g->nnx; f.nnz = g->nnz; f.ld = g->nnz;
    f.gox = g->gox; f.goz = g->goz; f.dnx = g->dnx; f.dnz = g->dnz;
    f.veln = veln; f.ttn = ttn; f.nsts = stc;
    rc = travel(&f, x, z, 2);
    free(velr); free(ttr); free(str); free(stc); free(f.hx); free(f.hz);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* receivers                                                                                  */

static float min_cell_km(const dso_grid *g)
{
    float dpl = g->dnx * g->earth;
    float rd1 = g->dnz * g->earth * sinf(g->gox);
    if (rd1 < dpl) dpl = rd1;
    rd1 = g->dnz * g->earth * sinf(g->gox + (float)(g->nnx - 1) * g->dnx);
    if (rd1 < dpl) dpl = rd1;
    return dpl;
}

int dso_srtimes(const dso_grid *g, const float *veln, const float *ttn,
                float scx, float scz, float rcx1, float rcz1, float *t)
{
    const int ld = g->nnz;
    const float gox = g->gox, goz = g->goz, dnx = g->dnx, dnz = g->dnz, earth = g->earth;
    int irx = (int)((rcx1 - gox) / dnx) + 1;
    int irz = (int)((rcz1 - goz) / dnz) + 1;
    int sw = 0;
    if (irx < 1 || irx > g->nnx) sw = 1;
    if (irz < 1 || irz > g->nnz) sw = 1;
    if (sw) return -1;
    if (irx == g->nnx) irx -= 1;
    if (irz == g->nnz) irz -= 1;
    int isx = (int)((scx - gox) / dnx) + 1;
    int isz = (int)((scz - goz) / dnz) + 1;
    float dpl = min_cell_km(g);
    float sred = p2((scx - rcx1) * earth);
    sred = sred + p2((scz - rcz1) * earth * sinf(rcx1));
    sred = sqrtf(sred);
    if (sred < dpl) sw = 1;
    if (isx == irx) { if (isz == irz) sw = 1; }
    float trr;
    if (sw) {
        /* NOTE: the reference does not clamp isx/isz here (:1703-1704); for a source on the last node
         * row/column it reads veln one node past the grid (undefined).  Clamped here and in the product. */
        float vss[2][2];
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) {
                int cz = isz - 1 + l, cx = isx - 1 + k;
                if (cz > g->nnz) cz = g->nnz;
                if (cx > g->nnx) cx = g->nnx;
                vss[k - 1][l - 1] = AT(veln, ld, cz, cx);
            }
        float drx = (scx - gox) - (float)(isx - 1) * dnx;
        float drz = (scz - goz) - (float)(isz - 1) * dnz;
        float vels = bilinear(vss, dnx, dnz, drx, drz);
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) vss[k - 1][l - 1] = AT(veln, ld, irz - 1 + l, irx - 1 + k);
        drx = (rcx1 - gox) - (float)(irx - 1) * dnx;
        drz = (rcz1 - goz) - (float)(irz - 1) * dnz;
        float velr = bilinear(vss, dnx, dnz, drx, drz);
        trr = 2.0f * sred / (vels + velr);
    } else {
        float drx = (rcx1 - gox) - (float)(irx - 1) * dnx;
        float drz = (rcz1 - goz) - (float)(irz - 1) * dnz;
        trr = 0.0f;
        for (int k = 1; k <= 2; ++k)
            for (int l = 1; l <= 2; ++l) {
                float produ = (1.0f - fabsf(((float)(l - 1) * dnz - drz) / dnz)) *
                              (1.0f - fabsf(((float)(k - 1) * dnx - drx) / dnx));
                trr = trr + AT(ttn, ld, irz - 1 + l, irx - 1 + k) * produ;
            }
    }
    *t = trr;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* rays + Frechet kernel on the B-spline vertices                                             */

/* velocity at a point from the four corners of node cell (ipx, ipz); :2163-2172 */
static float cell_velocity(const dso_grid *g, const float *veln, int ipx, int ipz, float drx, float drz)
{
    float vel = 0.0f;
    for (int l = 1; l <= 2; ++l)
        for (int m = 1; m <= 2; ++m) {
            float produ = (1.0f - fabsf(((float)(m - 1) * g->dnz - drz) / g->dnz));
            produ = produ * (1.0f - fabsf(((float)(l - 1) * g->dnx - drx) / g->dnx));
            if (ipz - 1 + m <= g->nnz && ipx - 1 + l <= g->nnx)
                vel = vel + AT(veln, g->nnz, ipz - 1 + m, ipx - 1 + l) * produ;
        }
    return vel;
}

/* optional record of the ray's points rgx(j), rgz(j) (CalSurfG.f90:1910-1911, :2043-2076: receiver, every step's end
 * point after clamping, finally the source) for dso_rpaths_path below: what the reference's disabled dump
 * (:2276-2283, raypath.out) would write */
static _Thread_local float *t_path = NULL;
static _Thread_local int t_path_cap = 0, t_path_n = 0;
static void path_push(float x, float z)
{
    if (t_path && t_path_n < t_path_cap) { t_path[2 * t_path_n] = x; t_path[2 * t_path_n + 1] = z; }
    t_path_n += 1;
}

int dso_rpaths(const dso_grid *g, const dso_box *b, const float *veln, const float *ttn,
               const float *ttnr, const int *nstsr, float scx, float scz, float surfrcx, float surfrcz,
               float *fdm, int *rbint, int *nsteps)
{
    const int nnx = g->nnx, nnz = g->nnz, ld = g->nnz, ldr = b->nnz;
    const int nnxr = b->nnx, nnzr = b->nnz;
    const float gox = g->gox, goz = g->goz, dnx = g->dnx, dnz = g->dnz, earth = g->earth;
    const float goxr = b->gox, gozr = b->goz, dnxr = b->dnx, dnzr = b->dnz;
    const float dvx = g->dvx, dvz = g->dvz;
    const int gdx = g->gdx, gdz = g->gdz;
    const int ldf = g->nvz + 2;
    const long maxrp = (long)nnx * (long)nnz;

    int isx = (int)((scx - goxr) / dnxr) + 1;
    int isz = (int)((scz - gozr) / dnzr) + 1;
    float dpl = 0.5f * min_cell_km(g);
    memset(fdm, 0, sizeof(float) * (size_t)(g->nvz + 2) * (size_t)(g->nvx + 2));
    if (nsteps) *nsteps = 0;

    int ipx = (int)((surfrcx - gox) / dnx) + 1;
    int ipz = (int)((surfrcz - goz) / dnz) + 1;
    int sw = 0;
    if (ipx < 1 || ipx >= nnx) sw = 1;
    if (ipz < 1 || ipz >= nnz) sw = 1;
    if (sw) return -1;

    float rgx = surfrcx, rgz = surfrcz;
    path_push(rgx, rgz);
    float sred = p2((scx - rgx) * earth);
    sred = sred + p2((scz - rgz) * earth * sinf(rgx));
    sred = sqrtf(sred);
    if (sred < 2.0f * dpl) { sw = 1; path_push(scx, scz); }

    int ipxr = (int)((surfrcx - goxr) / dnxr) + 1;
    int ipzr = (int)((surfrcz - gozr) / dnzr) + 1;
    int igref = 1;
    if (ipxr < 1 || ipxr >= nnxr) igref = 0;
    if (ipzr < 1 || ipzr >= nnzr) igref = 0;
    if (igref) {
        if (AT(nstsr, ldr, ipzr, ipxr) != 0 || AT(nstsr, ldr, ipzr + 1, ipxr) != 0) igref = 0;
        if (AT(nstsr, ldr, ipzr, ipxr + 1) != 0 || AT(nstsr, ldr, ipzr + 1, ipxr + 1) != 0) igref = 0;
    }
    if (!sw && igref && ipxr == isx && ipzr == isz) { sw = 1; path_push(scx, scz); }

    for (long j = 1; j <= maxrp; ++j) {
        if (sw) break;
        float dtx, dtz;
        if (igref) {
            dtx = AT(ttnr, ldr, ipzr, ipxr + 1) - AT(ttnr, ldr, ipzr, ipxr);
            dtx = dtx + AT(ttnr, ldr, ipzr + 1, ipxr + 1) - AT(ttnr, ldr, ipzr + 1, ipxr);
            dtx = dtx / (2.0f * earth * dnxr);
            dtz = AT(ttnr, ldr, ipzr + 1, ipxr) - AT(ttnr, ldr, ipzr, ipxr);
            dtz = dtz + AT(ttnr, ldr, ipzr + 1, ipxr + 1) - AT(ttnr, ldr, ipzr, ipxr + 1);
            dtz = dtz / (2.0f * earth * sinf(rgx) * dnzr);
        } else {
            dtx = AT(ttn, ld, ipz, ipx + 1) - AT(ttn, ld, ipz, ipx);
            dtx = dtx + AT(ttn, ld, ipz + 1, ipx + 1) - AT(ttn, ld, ipz + 1, ipx);
            dtx = dtx / (2.0f * earth * dnx);
            dtz = AT(ttn, ld, ipz + 1, ipx) - AT(ttn, ld, ipz, ipx);
            dtz = dtz + AT(ttn, ld, ipz + 1, ipx + 1) - AT(ttn, ld, ipz, ipx + 1);
            dtz = dtz / (2.0f * earth * sinf(rgx) * dnz);
        }
        float rd1 = sqrtf(p2(dtx) + p2(dtz));
        float rgx1 = rgx - dpl * dtx / (earth * rd1);
        float rgz1 = rgz - dpl * dtz / (earth * sinf(rgx) * rd1);
        if (nsteps) *nsteps += 1;

        const int ipxo = ipx, ipzo = ipz;
        ipxr = (int)((rgx1 - goxr) / dnxr) + 1;
        ipzr = (int)((rgz1 - gozr) / dnzr) + 1;
        igref = 1;
        if (ipxr < 1 || ipxr >= nnxr) igref = 0;
        if (ipzr < 1 || ipzr >= nnzr) igref = 0;
        if (igref) {
            if (AT(nstsr, ldr, ipzr, ipxr) != 0 || AT(nstsr, ldr, ipzr + 1, ipxr) != 0) igref = 0;
            if (AT(nstsr, ldr, ipzr, ipxr + 1) != 0 || AT(nstsr, ldr, ipzr + 1, ipxr + 1) != 0) igref = 0;
        }
        ipx = (int)((rgx1 - gox) / dnx) + 1;
        ipz = (int)((rgz1 - goz) / dnz) + 1;

        sred = p2((scx - rgx1) * earth);
        sred = sred + p2((scz - rgz1) * earth * sinf(rgx1));
        sred = sqrtf(sred);
        sw = 0;
        if (sred < 2.0f * dpl) sw = 1;
        if (!sw && igref && ipxr == isx && ipzr == isz) sw = 1;

        if (ipx < 1) { rgx1 = gox; ipx = 1; *rbint = 1; }
        if (ipx >= nnx) { rgx1 = gox + (float)(nnx - 1) * dnx; ipx = nnx - 1; *rbint = 1; }
        if (ipz < 1) { rgz1 = goz; ipz = 1; *rbint = 1; }
        if (ipz >= nnz) { rgz1 = goz + (float)(nnz - 1) * dnz; ipz = nnz - 1; *rbint = 1; }
        path_push(rgx1, rgz1);
        if (sw) path_push(scx, scz);

        /* split the segment at B-spline cell faces (:2112-2156) */
        const int ivx = (ipx - 1) / gdx + 1, ivz = (ipz - 1) / gdz + 1;
        const int ivxo = (ipxo - 1) / gdx + 1, ivzo = (ipzo - 1) / gdz + 1;
        int nhp = 0, chp[4];
        float vrat[4];
        if (ivx != ivxo) {
            nhp += 1;
            float xi = (ivx > ivxo) ? gox + (float)(ivx - 1) * dvx : gox + (float)ivx * dvx;
            vrat[nhp - 1] = (xi - rgx) / (rgx1 - rgx);
            chp[nhp - 1] = 1;
        }
        if (ivz != ivzo) {
            nhp += 1;
            float zi = (ivz > ivzo) ? goz + (float)(ivz - 1) * dvz : goz + (float)ivz * dvz;
            float r = (zi - rgz) / (rgz1 - rgz);
            if (nhp == 1) { vrat[0] = r; chp[0] = 2; }
            else if (r >= vrat[nhp - 2]) { vrat[nhp - 1] = r; chp[nhp - 1] = 2; }
            else {
                vrat[nhp - 1] = vrat[nhp - 2]; chp[nhp - 1] = chp[nhp - 2];
                vrat[nhp - 2] = r; chp[nhp - 2] = 2;
            }
        }
        nhp += 1;
        vrat[nhp - 1] = 1.0f;
        chp[nhp - 1] = 0;

        float drx = (rgx - gox) - (float)(ipxo - 1) * dnx;
        float drz = (rgz - goz) - (float)(ipzo - 1) * dnz;
        float vel = cell_velocity(g, veln, ipxo, ipzo, drx, drz);
        drx = (rgx - gox) - (float)(ivxo - 1) * dvx;
        drz = (rgz - goz) - (float)(ivzo - 1) * dvz;
        float vi[4], wi[4], vio[4], wio[4];
        bspl4(drx / dvx, vi);
        bspl4(drz / dvz, wi);
        int ivxt = ivxo, ivzt = ivzo;
        for (int k = 1; k <= nhp; ++k) {
            float velo = vel;
            memcpy(vio, vi, sizeof vi);
            memcpy(wio, wi, sizeof wi);
            if (k > 1) {
                if (chp[k - 2] == 1) ivxt = ivx;
                else if (chp[k - 2] == 2) ivzt = ivz;
            }
            float rigz = rgz + vrat[k - 1] * (rgz1 - rgz);
            float rigx = rgx + vrat[k - 1] * (rgx1 - rgx);
            int ipxt = (int)((rigx - gox) / dnx) + 1;
            int ipzt = (int)((rigz - goz) / dnz) + 1;
            drx = (rigx - gox) - (float)(ipxt - 1) * dnx;
            drz = (rigz - goz) - (float)(ipzt - 1) * dnz;
            /* same weights as cell_velocity but with the x offset in the outer loop (:2216-2224) */
            vel = 0.0f;
            for (int m = 1; m <= 2; ++m)
                for (int n = 1; n <= 2; ++n) {
                    float produ = (1.0f - fabsf(((float)(n - 1) * dnz - drz) / dnz));
                    produ = produ * (1.0f - fabsf(((float)(m - 1) * dnx - drx) / dnx));
                    if (ipzt - 1 + n <= nnz && ipxt - 1 + m <= nnx)
                        vel = vel + AT(veln, ld, ipzt - 1 + n, ipxt - 1 + m) * produ;
                }
            drx = (rigx - gox) - (float)(ivxt - 1) * dvx;
            drz = (rigz - goz) - (float)(ivzt - 1) * dvz;
            bspl4(drx / dvx, vi);
            bspl4(drz / dvz, wi);
            float dinc = (k == 1) ? vrat[0] * dpl : (vrat[k - 1] - vrat[k - 2]) * dpl;
            for (int l = 1; l <= 4; ++l)
                for (int m = 1; m <= 4; ++m) {
                    float r1 = vi[m - 1] * wi[l - 1] / p2(vel);
                    float r2 = vio[m - 1] * wio[l - 1] / p2(velo);
                    r1 = -(r1 + r2) * dinc / 2.0f;
                    float *cell = &fdm[(size_t)(ivxt - 2 + m) * (size_t)ldf + (size_t)(ivzt - 2 + l)];
                    *cell = r1 + *cell;
                }
        }
        rgx = rgx1; rgz = rgz1;
    }
    return 0;
}

/* dso_rpaths plus the ray's points: path gets up to cap (colatitude, longitude) pairs in radians, *npath the number
 * of points the ray has (it may exceed cap) */
int dso_rpaths_path(const dso_grid *g, const dso_box *b, const float *veln, const float *ttn,
                    const float *ttnr, const int *nstsr, float scx, float scz, float surfrcx, float surfrcz,
                    float *fdm, int *rbint, int *nsteps, float *path, int cap, int *npath)
{
    t_path = path; t_path_cap = cap; t_path_n = 0;
    const int rc = dso_rpaths(g, b, veln, ttn, ttnr, nstsr, scx, scz, surfrcx, surfrcz, fdm, rbint, nsteps);
    *npath = t_path_n;
    t_path = NULL; t_path_cap = 0;
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* engine-level batch: travel times for (period slot, source) units with given velocity maps    */

#include <omp.h>

int dso_traveltimes(int nx, int ny, float goxd, float gozd, float dvxd, float dvzd, int gd,
                    int kmax, int nsrcsurf, int nrcf, const double *pv, const int *nsrcsurf1,
                    const int *nrc1, const float *scxf, const float *sczf, const float *rcxf,
                    const float *rczf, float *dsurf, int nthreads)
{
    dso_grid g;
    dso_grid_init(&g, nx, ny, goxd, gozd, dvxd, dvzd, gd);
    const size_t nc = (size_t)g.nnx * (size_t)g.nnz, nv = (size_t)nx * (size_t)ny;
    /* flatten the (knumi, srcnum) loop nest of CalSurfG.f90:1144-1145 and prefix-sum the outputs */
    int nunits = 0;
    for (int k = 0; k < kmax; ++k) nunits += nsrcsurf1[k];
    int *uk = (int *)malloc(sizeof(int) * (size_t)(nunits + 1)), *us = (int *)malloc(sizeof(int) * (size_t)(nunits + 1));
    size_t *first = (size_t *)malloc(sizeof(size_t) * (size_t)(nunits + 1));
    size_t tot = 0;
    int u = 0;
    for (int k = 0; k < kmax; ++k)
        for (int s = 0; s < nsrcsurf1[k]; ++s) {
            uk[u] = k; us[u] = s; first[u] = tot;
            tot += (size_t)nrc1[(size_t)k * nsrcsurf + s];
            ++u;
        }
    /* one diced grid per period slot (the reference re-dices per source; same values) */
    float *veln = (float *)malloc(sizeof(float) * nc * (size_t)kmax);
    for (int k = 0; k < kmax; ++k) dso_gridder(&g, pv + nv * (size_t)k, veln + nc * (size_t)k);
    int rc = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int q = 0; q < nunits; ++q) {
        const int k = uk[q], s = us[q];
        float *ttn = (float *)malloc(sizeof(float) * nc);
        dso_box b;
        const float x = scxf[(size_t)k * nsrcsurf + s], z = sczf[(size_t)k * nsrcsurf + s];
        if (dso_solve_source(&g, pv + nv * (size_t)k, veln + nc * (size_t)k, x, z, &b, ttn, NULL, NULL, NULL, NULL) != 0) {
#pragma omp atomic write
            rc = -1;
        } else {
            const int nr = nrc1[(size_t)k * nsrcsurf + s];
            for (int r = 0; r < nr; ++r) {
                const size_t ri = ((size_t)k * nsrcsurf + s) * (size_t)nrcf + (size_t)r;
                float t = 0.0f;
                if (dso_srtimes(&g, veln + nc * (size_t)k, ttn, x, z, rcxf[ri], rczf[ri], &t) != 0) {
#pragma omp atomic write
                    rc = -2;
                }
                dsurf[first[q] + (size_t)r] = t;
            }
        }
        free(ttn);
    }
    free(veln); free(uk); free(us); free(first);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* whole boundary: CalSurfG (CalSurfG.f90:939-1459) and synthetic (:2412-2865)                  */

static inline float r2(float x) { return x * x; }
static inline float r3(float x) { return (x * x) * x; }
static inline float r4(float x) { return ((x * x) * x) * x; }

/* Frechet row of one ray from its vertex kernel fdm; CalSurfG.f90:1383-1432.
 * sen_*: (nx*ny, kmax, nz) doubles; slot: 0-based period slot. Appends to rw/iw/col. */
static void assemble_row(int nx, int ny, int nz, const float *vels, const float *depz, const float *fdm,
                         const double *sen_vs, const double *sen_vp, const double *sen_rho, int kmax, int slot,
                         float *row, int rownum, float *rw, int *iw, int *col, int *nar)
{
    const int nvx = nx - 2, nvz = ny - 2, nparpi = nvx * nvz * (nz - 1);
    const size_t ncol = (size_t)nx * ny;
    const float ftol = 1e-4f;
    const int shallow = depz[nz - 2] < 35.0f;
    for (int n = 0; n < nparpi; ++n) row[n] = 0.0f;
    for (int jj = 1; jj <= nvz; ++jj)
        for (int kk = 1; kk <= nvx; ++kk) {
            const float f = fdm[(size_t)kk * (nvz + 2) + jj];
            if (!(fabsf(f) >= ftol)) continue;
            const size_t c = (size_t)jj * nx + kk;            /* 0-based column of vertex (kk+1, jj+1) */
            for (int k = 1; k <= nz - 1; ++k) {
                const float v = vels[(size_t)(k - 1) * ncol + c];
                float coe_a, vpft;
                if (shallow) {
                    coe_a = 2.0947f - (0.8206f * 2.0f) * v + (0.2683f * 3.0f) * r2(v) - (0.0251f * 4.0f) * r3(v);
                    vpft = 0.9409f + 2.0947f * v - 0.8206f * r2(v) + 0.2683f * r3(v) - 0.0251f * r4(v);
                } else {
                    coe_a = 2.2110f - (0.8984f * 2.0f) * v + (0.2786f * 3.0f) * r2(v) - (0.02412f * 4.0f) * r3(v);
                    vpft = 0.9098f + 2.2110f * v - 0.8984f * r2(v) + 0.2786f * r3(v) - 0.02412f * r4(v);
                }
                const float coe_rho = coe_a * (1.6612f - (0.4721f * 2.0f) * vpft + (0.0671f * 3.0f) * r2(vpft) -
                                               (0.0043f * 4.0f) * r3(vpft) + (0.000106f * 5.0f) * r4(vpft));
                const size_t si = ((size_t)(k - 1) * kmax + slot) * ncol + c;
                const double val = (sen_vp[si] * (double)coe_a + sen_rho[si] * (double)coe_rho + sen_vs[si]) * (double)f;
                row[(size_t)(k - 1) * nvx * nvz + (size_t)(jj - 1) * nvx + (kk - 1)] = (float)val;
            }
        }
    for (int n = 1; n <= nparpi; ++n)
        if (fabsf(row[n - 1]) > ftol) {
            *nar += 1;
            rw[*nar - 1] = row[n - 1];
            iw[*nar] = rownum;          /* iw(nar+1) */
            col[*nar - 1] = n;
        }
}

int dso_calsurfg(const int *pnx, const int *pny, const int *pnz, const int *pnparpi, const float *vels,
                 int *iw, float *rw, int *col, float *dsurf,
                 const float *goxdf, const float *gozdf, const float *dvxdf, const float *dvzdf,
                 const int *pkmaxRc, const int *pkmaxRg, const int *pkmaxLc, const int *pkmaxLg,
                 const double *tRc, const double *tRg, const double *tLc, const double *tLg,
                 const int *wavetype, const int *igrt, const int *periods, const float *depz,
                 const float *pminthk, const float *scxf, const float *sczf, const float *rcxf,
                 const float *rczf, const int *nrc1, const int *nsrcsurf1, const int *pkmax,
                 const int *pnsrcsurf, const int *pnrcf, int *nar)
{
    const int nx = *pnx, ny = *pny, nz = *pnz, kmax = *pkmax, nsrcsurf = *pnsrcsurf, nrcf = *pnrcf;
    const int kmaxRc = *pkmaxRc, kmaxRg = *pkmaxRg, kmaxLc = *pkmaxLc, kmaxLg = *pkmaxLg;
    const float minthk = *pminthk;
    const size_t ncol = (size_t)nx * ny;
    (void)pnparpi;
    dso_grid g;
    dso_grid_init(&g, nx, ny, *goxdf, *gozdf, *dvxdf, *dvzdf, 8);
    const size_t nc = (size_t)g.nnx * g.nnz;
    const int kmx = kmax > 0 ? kmax : 1;
    double *pvRc = calloc(ncol * kmx, 8), *pvRg = calloc(ncol * (kmaxRg > 0 ? kmaxRg : 1), 8);
    double *pvLc = calloc(ncol * kmx, 8), *pvLg = calloc(ncol * (kmaxLg > 0 ? kmaxLg : 1), 8);
    /* combined sensitivity arrays sen_*(nx*ny, kmax, nz), filled per type at their slot offsets */
    double *sen_vs = calloc(ncol * kmx * nz, 8), *sen_vp = calloc(ncol * kmx * nz, 8), *sen_rho = calloc(ncol * kmx * nz, 8);
    const int kmax1 = kmaxRc, kmax2 = kmaxRc + kmaxRg, kmax3 = kmaxRc + kmaxRg + kmaxLc;
    struct { int n, off, iwave, igr; const double *t; double *pv; } ty[4] = {
        { kmaxRc, 0, 2, 0, tRc, pvRc }, { kmaxRg, kmax1, 2, 1, tRg, pvRg }, { kmaxLc, kmax2, 1, 0, tLc, pvLc }, { kmaxLg, kmax3, 1, 1, tLg, pvLg } };
    for (int q = 0; q < 4; ++q) {
        if (ty[q].n <= 0) continue;
        if (ty[q].igr == 1)   /* phase velocities at the group periods overwrite the head of pvRc / pvLc (:1110, :1128) */
            dso_caldespersion(nx, ny, nz, vels, q == 1 ? pvRc : pvLc, ty[q].iwave, 0, ty[q].n, ty[q].t, depz, minthk);
        double *svs = malloc(ncol * ty[q].n * nz * 8), *svp = malloc(ncol * ty[q].n * nz * 8), *srh = malloc(ncol * ty[q].n * nz * 8);
        dso_depthkernel(nx, ny, nz, vels, ty[q].pv, svs, svp, srh, ty[q].iwave, ty[q].igr, ty[q].n, ty[q].t, depz, minthk);
        for (int k = 0; k < nz; ++k)
            for (int p = 0; p < ty[q].n; ++p) {
                const size_t src = ((size_t)k * ty[q].n + p) * ncol, dst = ((size_t)k * kmax + ty[q].off + p) * ncol;
                memcpy(sen_vs + dst, svs + src, ncol * 8); memcpy(sen_vp + dst, svp + src, ncol * 8); memcpy(sen_rho + dst, srh + src, ncol * 8);
            }
        free(svs); free(svp); free(srh);
    }
    *nar = 0;
    int count1 = 0, rc = 0, rbint = 0;
    float *veln = malloc(4 * nc), *ttn = malloc(4 * nc), *ttnr = malloc(4 * 129 * 129), *fdm = malloc(4 * ncol);
    int *nstsr = malloc(4 * 129 * 129);
    float *row = malloc(4 * (size_t)(nx - 2) * (ny - 2) * (nz > 1 ? nz - 1 : 1));
    for (int knumi = 1; knumi <= kmax && rc == 0; ++knumi)
        for (int srcnum = 1; srcnum <= nsrcsurf1[knumi - 1] && rc == 0; ++srcnum) {
            const size_t sk = (size_t)(knumi - 1) * nsrcsurf + (srcnum - 1);
            const int wt = wavetype[sk], gr = igrt[sk], per = periods[sk];
            const double *velf = NULL;
            if (wt == 2 && gr == 0) velf = pvRc + ncol * (size_t)(per - 1);
            if (wt == 2 && gr == 1) velf = pvRg + ncol * (size_t)(per - 1);
            if (wt == 1 && gr == 0) velf = pvLc + ncol * (size_t)(per - 1);
            if (wt == 1 && gr == 1) velf = pvLg + ncol * (size_t)(per - 1);
            if (!velf) { rc = -4; break; }
            const int igroup = gr == 1 ? 2 : 1;
            int count11 = count1;
            const float x = scxf[sk], z = sczf[sk];
            for (int ig = 1; ig <= igroup && rc == 0; ++ig) {
                if (ig == 2 && wt == 2) velf = pvRc + ncol * (size_t)(per - 1);
                if (ig == 2 && wt == 1) velf = pvLc + ncol * (size_t)(per - 1);
                dso_gridder(&g, velf, veln);
                dso_box b;
                if (dso_solve_source(&g, velf, veln, x, z, &b, ttn, ttnr, nstsr, NULL, NULL) != 0) { rc = -3; break; }
                for (int istep = 1; istep <= nrc1[sk]; ++istep) {
                    const size_t ri = sk * (size_t)nrcf + (size_t)(istep - 1);
                    if (ig == 1) {
                        float t;
                        if (dso_srtimes(&g, veln, ttn, x, z, rcxf[ri], rczf[ri], &t) != 0) { rc = -3; break; }
                        count1 += 1;
                        dsurf[count1 - 1] = t;
                    }
                    if (gr == 0 || (ig == 2 && gr == 1)) {
                        count11 += 1;
                        if (dso_rpaths(&g, &b, veln, ttn, ttnr, nstsr, x, z, rcxf[ri], rczf[ri], fdm, &rbint, NULL) != 0) { rc = -3; break; }
                        assemble_row(nx, ny, nz, vels, depz, fdm, sen_vs, sen_vp, sen_rho, kmax, knumi - 1, row, count11, rw, iw, col, nar);
                    }
                }
            }
        }
    free(pvRc); free(pvRg); free(pvLc); free(pvLg); free(sen_vs); free(sen_vp); free(sen_rho);
    free(veln); free(ttn); free(ttnr); free(fdm); free(nstsr); free(row);
    return rc;
}

int dso_synthetic(const int *pnx, const int *pny, const int *pnz, const int *pnparpi, const float *vels,
                  float *obst,
                  const float *goxdf, const float *gozdf, const float *dvxdf, const float *dvzdf,
                  const int *pkmaxRc, const int *pkmaxRg, const int *pkmaxLc, const int *pkmaxLg,
                  const double *tRc, const double *tRg, const double *tLc, const double *tLg,
                  const int *wavetype, const int *igrt, const int *periods, const float *depz,
                  const float *pminthk, const float *scxf, const float *sczf, const float *rcxf,
                  const float *rczf, const int *nrc1, const int *nsrcsurf1, const int *pkmax,
                  const int *pnsrcsurf, const int *pnrcf, const float *noiselevel)
{
    /* noise: obst = t + t*gaussian()*noiselevel with an unseeded generator in the reference (:2840);
     * the oracle supports noiselevel == 0 only (deterministic part) */
    const int nx = *pnx, ny = *pny, nz = *pnz, kmax = *pkmax, nsrcsurf = *pnsrcsurf, nrcf = *pnrcf;
    const size_t ncol = (size_t)nx * ny;
    (void)pnparpi; (void)noiselevel;
    dso_grid g;
    dso_grid_init(&g, nx, ny, *goxdf, *gozdf, *dvxdf, *dvzdf, 5);
    const size_t nc = (size_t)g.nnx * g.nnz;
    const int n4[4] = { *pkmaxRc, *pkmaxRg, *pkmaxLc, *pkmaxLg };
    const int iw4[4] = { 2, 2, 1, 1 }, ig4[4] = { 0, 1, 0, 1 };
    const double *t4[4] = { tRc, tRg, tLc, tLg };
    double *pv4[4];
    for (int q = 0; q < 4; ++q) {
        pv4[q] = calloc(ncol * (n4[q] > 0 ? n4[q] : 1), 8);
        if (n4[q] > 0) dso_caldespersion(nx, ny, nz, vels, pv4[q], iw4[q], ig4[q], n4[q], t4[q], depz, *pminthk);
    }
    float *veln = malloc(4 * nc), *ttn = malloc(4 * nc);
    int count1 = 0, rc = 0;
    for (int knumi = 1; knumi <= kmax && rc == 0; ++knumi)
        for (int srcnum = 1; srcnum <= nsrcsurf1[knumi - 1] && rc == 0; ++srcnum) {
            const size_t sk = (size_t)(knumi - 1) * nsrcsurf + (srcnum - 1);
            const int wt = wavetype[sk], gr = igrt[sk], per = periods[sk];
            const int q = (wt == 2 ? 0 : 2) + (gr == 1 ? 1 : 0);
            const double *velf = pv4[q] + ncol * (size_t)(per - 1);
            dso_gridder(&g, velf, veln);
            dso_box b;
            if (dso_solve_source(&g, velf, veln, scxf[sk], sczf[sk], &b, ttn, NULL, NULL, NULL, NULL) != 0) { rc = -3; break; }
            for (int istep = 1; istep <= nrc1[sk]; ++istep) {
                const size_t ri = sk * (size_t)nrcf + (size_t)(istep - 1);
                float t;
                if (dso_srtimes(&g, veln, ttn, scxf[sk], sczf[sk], rcxf[ri], rczf[ri], &t) != 0) { rc = -3; break; }
                obst[count1++] = t;
            }
        }
    for (int q = 0; q < 4; ++q) free(pv4[q]);
    free(veln); free(ttn);
    return rc;
}

/* aprod.f90:7-60 */
void dso_aprod(const int *mode, const int *m, const int *n, float *x, float *y, const int *leniw, const int *lenrw,
               const int *iw, const float *rw)
{
    (void)m; (void)n; (void)leniw; (void)lenrw;
    const int kk = iw[0];
    const int *row = iw + 1, *col = iw + 1 + kk;
    for (int k = 0; k < kk; ++k) {
        if (*mode == 1) y[row[k] - 1] = y[row[k] - 1] + rw[k] * x[col[k] - 1];
        else x[col[k] - 1] = x[col[k] - 1] + rw[k] * y[row[k] - 1];
    }
}
