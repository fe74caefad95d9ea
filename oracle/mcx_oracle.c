/*
 * oracle/mcx_oracle.c -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * CPU restatement of the mcmcf90 sampling hot path for one chain.  Every
 * function cites the reference lines it follows.  Arithmetic conventions:
 *
 *  - code that is Fortran source inside the reference (MCMC_*.F90, matutils.F90
 *    covmat, mcmcrand.F90, dchud.f, dchdd.f) is restated one IEEE operation per
 *    source operator, NO fused multiply-add (the reference is built for generic
 *    x86-64, which has none); build this file with -ffp-contract=off;
 *  - BLAS/LAPACK are NOT vendored by the reference and are unpinned ("any
 *    implementation", testcases/Makefile:11, INSTALL.txt:14-15).  We pin them as
 *    the netlib reference algorithms (BLAS 3.8.0 level-1/2, LAPACK unblocked
 *    dpotf2/dtrti2/dlauu2, classic drotg/dnrm2) with each dot-product / axpy accumulation done
 *    as an fma() chain, in netlib loop order except dtrmv('U','T'), whose dot products run
 *    ascending (see mcxo_trmv_ut);
 *  - libm log/exp are pinned in mcx_math.h.
 *
 * Parity status: pinned against the real reference compiled from /root/reference
 * (oracle/_ref, flang + MKL + Philox interposer): accept sequences identical,
 * floating-point state within 1e-9 relative (different BLAS/libm rounding);
 * see tests/test_oracle_vs_reference.py and tests/golden/.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <float.h>
#include "mcx_oracle.h"
#include "mcx_math.h"
#include "mcx_targets.h"
#include "mcx_svd.h"

double mcxo_log(double x) { return mcxm_log(x); }
double mcxo_exp(double x) { return mcxm_exp(x); }

/* ------------------------------------------------------------------ config */

/* mcmcinit.F90:184-230 */
void mcxo_cfg_defaults(mcxo_cfg *c)
{
    memset(c, 0, sizeof *c);
    c->nsimu = 0; c->doadapt = 1; c->doburnin = 0; c->burnintime = 0; c->badaptint = -1;
    c->greedy = 0; c->scalelimit = 0.05; c->scalefactor = 2.5; c->drscale = 0.0;
    c->adaptint = 100; c->adapthist = 0; c->adaptend = 0; c->initcmatn = 0;
    c->N0 = 1.0; c->S02 = 0.0; c->updatesigma = 1; c->condmax = 0.0;
    c->method = MCXO_METHOD_DRAM; c->alphatarget = 0.234; c->nuparam = 0.7;
}

/* mcmcinit.F90:235-368 (sstype handling and dump-file names are host plumbing) */
int mcxo_cfg_check(mcxo_cfg *c)
{
    if (c->adapthist < 0) c->adapthist = 0;
    if (c->adaptint < 0) { c->adaptint = 0; c->doadapt = 0; }
    if (c->burnintime < 0) c->burnintime = 0;
    if (c->badaptint <= 0) c->badaptint = c->adaptint;
    if (c->badaptint == 0) c->doburnin = 0;
    if (c->initcmatn < 0) c->initcmatn = 0;
    if (c->scalelimit < 0.0 || c->scalelimit > 0.5) return -1;   /* :260-263 stop */
    if (c->scalefactor < 0.0) c->scalefactor = 1.0;
    if (c->method == MCXO_METHOD_SCAM) {
        c->doscam = 1;
        if (c->condmax <= 0.0) c->condmax = 1.0e15;
        c->doburnin = 0; c->drscale = 0.0;
    } else c->doscam = 0;
    if (c->method == MCXO_METHOD_RAM) c->drscale = 0.0;
    if (c->method == MCXO_METHOD_ER) c->drscale = 0.0;          /* 'no dr with er', MCMC_run_er.F90:26-29 */
    c->dodr = (c->drscale > 0.0);
    c->usesvd = (c->condmax > 0.0);
    return 0;
}

/* ------------------------------------------------------------------ targets */

double mcxo_ssfun(const mcxo_target *t, const double *th)
{
    switch (t->kind) {
    case MCXO_TARGET_GAUSS:   return mcxt_ss_gauss(t->npar, th, t->mu, t->lam);
    case MCXO_TARGET_BANANA:  return mcxt_ss_banana(t->npar, th, t->banana_b);
    case MCXO_TARGET_EXPDATA: return mcxt_ss_expdata(th, t->ndata, t->xdata, t->ydata);
    }
    return NAN;
}
void mcxo_ssfun_cols(const mcxo_target *t, const double *th, double *ss)
{
    if (t->ny > 1) mcxt_ss_expdata_cols(th, t->ndata, t->xdata, t->ydata, t->ny, ss);    /* the one multi-column target */
    else ss[0] = mcxo_ssfun(t, th);
}
double mcxo_priorfun(const mcxo_target *t, const double *th) { return mcxt_prior(t->npar, th, t->pri_mu, t->pri_sig); }
int mcxo_checkbounds(const mcxo_target *t, const double *th) { return mcxt_inbounds(t->npar, th, t->lo, t->hi); }

/* ------------------------------------------------------------------ RNG: mcmcrand.F90 */

/* normal_bm, mcmcrand.F90:166-190: Marsaglia polar, second deviate cached across calls */
double mcxo_normal(mcxo_rng *g)
{
    if (!g->saved) {
        double x1, x2, xx;
        do {
            x1 = mcxo_uniform(g); x2 = mcxo_uniform(g);       /* call random_number(x), x(2) */
            x1 = 2.0 * x1 - 1.0; x2 = 2.0 * x2 - 1.0;
            xx = x1 * x1 + x2 * x2;
        } while (!(xx < 1.0 && xx != 0.0));
        double z = sqrt(-2.0 * mcxm_log(xx) / xx);
        g->saved_y = z * x1; g->saved = 1;
        return z * x2;
    }
    g->saved = 0;
    return g->saved_y;
}

/* random_gamma + gammar_mt, mcmcrand.F90:86-162 (Marsaglia-Tsang) */
static double gammar_mt(mcxo_rng *g, double a, double b)
{
    double aa = a, bb = b;
    if (aa < 1.0) {                                   /* :136-146 (prints a warning in the reference) */
        double u = mcxo_uniform(g);
        bb = bb * mcxm_powu(u, 1.0 / aa);
        aa = aa + 1.0;
    }
    double d = aa - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    double x, v, u;
    for (;;) {
        do { x = mcxo_normal(g); v = 1.0 + c * x; } while (!(v > 0.0));
        v = (v * v) * v;
        u = mcxo_uniform(g);
        double x2 = x * x;
        if (u < 1.0 - 0.0331 * (x2 * x2)) break;
        if (mcxm_log(u) < 0.5 * x2 + d * (1.0 - v + mcxm_log(v))) break;
    }
    return bb * d * v;
}
double mcxo_gamma(mcxo_rng *g, double a, double b)
{
    if (a < 1.0) {                                    /* random_gamma, mcmcrand.F90:102-105 (the route MCMC_DRAM.F90:201 takes):
                                                         u first, then gammar_mt(1+a, b) * u**(1/a); u**e pinned in
                                                         mcx_math.h */
        double u = mcxo_uniform(g);
        return gammar_mt(g, 1.0 + a, b) * mcxm_powu(u, 1.0 / a);
    }
    return gammar_mt(g, a, b);
}

/* ------------------------------------------------------------------ BLAS / LAPACK (pinned netlib order) */

#define A_(M, i, j, n) (M)[(size_t)(i) + (size_t)(j) * (size_t)(n)]

/* dtrmv('U','T','N'), matutils.F90:108-109: x <- R'x, i.e. p_j = sum_{i<=j} R(i,j) x_i.  The reference
 * leaves the BLAS unpinned, so the accumulation order inside each dot product is ours to fix: ascending
 * i as one fma chain from 0 (netlib's reference loop runs i = j..1; MKL, which the reference build here
 * links, is blocked).  Ascending order is what lets the device accumulate the NEXT proposal while
 * DCHUD streams the freshly updated rows 1..p of the factor. */
void mcxo_trmv_ut(int n, const double *R, double *x)
{
    for (int j = n - 1; j >= 0; --j) {
        double temp = 0.0;
        for (int i = 0; i <= j; ++i) temp = fma(A_(R, i, j, n), x[i], temp);
        x[j] = temp;
    }
}

/* The same product in netlib dtrmv's own order (temp = x(j) a(j,j); temp += a(i,j) x(i), i = j-1..1), as an fma chain.
 * Used for the one proposal that follows a successful Cholesky downdate of MCMC_adapt_ram: DCHDD finishes every
 * column from its diagonal upwards, so this is the order in which the device can accumulate the next proposal while
 * the downdated factor streams out. */
void mcxo_trmv_ut_desc(int n, const double *R, double *x)
{
    for (int j = n - 1; j >= 0; --j) {
        double temp = A_(R, j, j, n) * x[j];
        for (int i = j - 1; i >= 0; --i) temp = fma(A_(R, i, j, n), x[i], temp);
        x[j] = temp;
    }
}

/* dpotf2('U'): A = U'U, upper triangle overwritten, lower untouched (matutils.F90:363 via dpotrf) */
int mcxo_potrf_u(int n, double *A)
{
    for (int j = 0; j < n; ++j) {
        double dot = 0.0;                              /* ddot(j-1, a(1,j), a(1,j)) */
        for (int i = 0; i < j; ++i) dot = fma(A_(A, i, j, n), A_(A, i, j, n), dot);
        double ajj = A_(A, j, j, n) - dot;
        if (!(ajj > 0.0)) { A_(A, j, j, n) = ajj; return j + 1; }
        ajj = sqrt(ajj);
        A_(A, j, j, n) = ajj;
        if (j < n - 1) {
            double rinv = 1.0 / ajj;
            for (int k = j + 1; k < n; ++k) {          /* dgemv('T') then dscal(1/ajj) */
                double t = 0.0;
                for (int i = 0; i < j; ++i) t = fma(A_(A, i, k, n), A_(A, i, j, n), t);
                A_(A, j, k, n) = (A_(A, j, k, n) - t) * rinv;
            }
        }
    }
    return 0;
}

/* dpotri('U') = dtrti2('U','N') + dlauu2('U'): A (holding U) -> upper triangle of inv(U'U) (MCMC_adapt.F90:219) */
int mcxo_potri_u(int n, double *A)
{
    for (int j = 0; j < n; ++j) if (A_(A, j, j, n) == 0.0) return j + 1;
    /* dtrti2: for j: a(j,j) = 1/a(j,j); x = a(1:j-1,j); x <- T x (dtrmv 'U','N','N' with the inverted block); x *= -a(j,j) */
    for (int j = 0; j < n; ++j) {
        A_(A, j, j, n) = 1.0 / A_(A, j, j, n);
        double ajj = -A_(A, j, j, n);
        /* dtrmv('U','N','N', j): for jj = 1..j: if x(jj)!=0: temp=x(jj); x(i) += temp*a(i,jj), i<jj; x(jj) *= a(jj,jj) */
        for (int jj = 0; jj < j; ++jj) {
            double temp = A_(A, jj, j, n);
            if (temp != 0.0) {
                for (int i = 0; i < jj; ++i) A_(A, i, j, n) = fma(temp, A_(A, i, jj, n), A_(A, i, j, n));
                A_(A, jj, j, n) = temp * A_(A, jj, jj, n);
            }
        }
        for (int i = 0; i < j; ++i) A_(A, i, j, n) = ajj * A_(A, i, j, n);
    }
    /* dlauu2('U'): A <- U U' */
    for (int i = 0; i < n; ++i) {
        double aii = A_(A, i, i, n);
        if (i < n - 1) {
            double dot = 0.0;                          /* ddot(n-i+1, a(i,i:n), a(i,i:n)) */
            for (int k = i; k < n; ++k) dot = fma(A_(A, i, k, n), A_(A, i, k, n), dot);
            A_(A, i, i, n) = dot;
            /* dgemv('N', i-1, n-i, 1, a(1,i+1), a(i,i+1) (row), aii, a(1,i)):
               y = aii*y first, then for each column k: y(r) += a(i,k)*a(r,k) */
            for (int r = 0; r < i; ++r) A_(A, r, i, n) = aii * A_(A, r, i, n);
            for (int k = i + 1; k < n; ++k) {
                double temp = A_(A, i, k, n);
                if (temp != 0.0)
                    for (int r = 0; r < i; ++r) A_(A, r, i, n) = fma(temp, A_(A, r, k, n), A_(A, r, i, n));
            }
        } else {
            for (int r = 0; r <= i; ++r) A_(A, r, i, n) = aii * A_(A, r, i, n);
        }
    }
    return 0;
}

int mcxo_symsvd(int n, double *G, double *V, double *s) { return mcxs_symsvd(n, G, V, s); }

/* dgemv (matutils.F90:161, matmulx), alpha = 1, beta = 0, netlib 3.8.0 loop order with fma accumulation:
 * 'N': y = 0; for j: temp = x(j); y(i) += temp*a(i,j).   'T': y(j) = sum_i a(i,j) x(i), i ascending. */
void mcxo_gemv(int trans, int n, const double *A, const double *x, double *y)
{
    if (!trans) {
        for (int i = 0; i < n; ++i) y[i] = 0.0;
        for (int j = 0; j < n; ++j) {
            double temp = x[j];
            for (int i = 0; i < n; ++i) y[i] = fma(temp, A_(A, i, j, n), y[i]);
        }
    } else {
        for (int j = 0; j < n; ++j) {
            double temp = 0.0;
            for (int i = 0; i < n; ++i) temp = fma(A_(A, i, j, n), x[i], temp);
            y[j] = temp;
        }
    }
}

/* classic netlib drotg (BLAS 3.8.0); dchud.f:138 only uses r (into da), c, s */
void mcxo_rotg(double *da, double *db, double *c, double *s)
{
    double a = *da, b = *db, roe = b, r, z;
    if (fabs(a) > fabs(b)) roe = a;
    double scale = fabs(a) + fabs(b);
    if (scale == 0.0) { *c = 1.0; *s = 0.0; r = 0.0; z = 0.0; }
    else {
        double t1 = a / scale, t2 = b / scale;
        r = scale * sqrt(t1 * t1 + t2 * t2);
        r = copysign(1.0, roe) * r;
        *c = a / r; *s = b / r;
        z = 1.0;
        if (fabs(a) > fabs(b)) z = *s;
        if (fabs(b) >= fabs(a) && *c != 0.0) z = 1.0 / *c;
    }
    *da = r; *db = z;
}

/* classic netlib dnrm2 (scale/ssq form), dchdd.f:149 */
double mcxo_nrm2(int n, const double *x)
{
    if (n < 1) return 0.0;
    if (n == 1) return fabs(x[0]);
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i) {
        if (x[i] != 0.0) {
            double absxi = fabs(x[i]);
            if (scale < absxi) { double q = scale / absxi; ssq = 1.0 + ssq * (q * q); scale = absxi; }
            else { double q = absxi / scale; ssq = ssq + q * q; }
        }
    }
    return scale * sqrt(ssq);
}

/* DCHUD with nz = 0, dchud.f:122-139 (vendored Fortran: no fma) */
void mcxo_chud(int p, double *R, const double *x, double *c, double *s)
{
    for (int j = 0; j < p; ++j) {
        double xj = x[j];
        for (int i = 0; i < j; ++i) {
            double rij = A_(R, i, j, p);
            double t = c[i] * rij + s[i] * xj;
            xj = c[i] * xj - s[i] * rij;
            A_(R, i, j, p) = t;
        }
        mcxo_rotg(&A_(R, j, j, p), &xj, &c[j], &s[j]);
    }
}

/* DCHDD with nz = 0, dchdd.f:141-179; returns info (-1: not positive definite, R untouched) */
int mcxo_chdd(int p, double *R, const double *x, double *c, double *s)
{
    s[0] = x[0] / A_(R, 0, 0, p);
    for (int j = 1; j < p; ++j) {
        double dot = 0.0;                              /* ddot(j-1, r(1,j), s) */
        for (int i = 0; i < j; ++i) dot = fma(A_(R, i, j, p), s[i], dot);
        s[j] = x[j] - dot;
        s[j] = s[j] / A_(R, j, j, p);
    }
    double norm = mcxo_nrm2(p, s);
    if (!(norm < 1.0)) return -1;
    double alpha = sqrt(1.0 - norm * norm);
    for (int ii = 0; ii < p; ++ii) {
        int i = p - ii - 1;
        double scale = alpha + fabs(s[i]);
        double a = alpha / scale, b = s[i] / scale;
        norm = sqrt(a * a + b * b);
        c[i] = a / norm; s[i] = b / norm;
        alpha = scale * norm;
    }
    for (int j = 0; j < p; ++j) {
        double xx = 0.0;
        for (int i = j; i >= 0; --i) {
            double rij = A_(R, i, j, p);
            double t = c[i] * xx + s[i] * rij;
            A_(R, i, j, p) = c[i] * rij - s[i] * xx;
            xx = t;
        }
    }
    return 0;
}

/* covmat, matutils.F90:232-341.  x: n rows, row r at x + r*ldx, p columns.
 * w: nw == n per-row weights, nw == 1 common weight, nw == 0 none. cmat col-major full symmetric. */
void mcxo_covmat(int n, int p, const double *x, int ldx, const double *w, int nw,
                 double *cmat, double *xmean, double *wsum, int update)
{
    double w2, wsum2;
    if (nw == n && w) { w2 = -1.0; wsum2 = 0.0; for (int i = 0; i < n; ++i) wsum2 = wsum2 + w[i]; }
    else if (nw == 1 && w) { w2 = w[0]; wsum2 = (double)n * w2; }
    else { w2 = 1.0; wsum2 = (double)n; }
    int doupdate = (update && *wsum > 0.0);
    double *d = (double *)malloc(sizeof(double) * (size_t)p);
    if (doupdate) {
        for (int i = 0; i < n; ++i) {                  /* :283-310 */
            const double *row = x + (size_t)i * ldx;
            for (int a = 0; a < p; ++a) d[a] = row[a] - xmean[a];
            double w3 = (w2 == -1.0) ? w[i] : w2;
            double f1 = w3 / (*wsum + w3 - 1.0);
            double f2 = *wsum / (*wsum + w3);
            for (int b = 0; b < p; ++b)
                for (int a = 0; a < p; ++a) {
                    double o = d[a] * d[b];
                    A_(cmat, a, b, p) = A_(cmat, a, b, p) + f1 * (f2 * o - A_(cmat, a, b, p));
                }
            double f3 = w3 / (*wsum + w3);
            for (int a = 0; a < p; ++a) xmean[a] = xmean[a] + f3 * d[a];
            *wsum = w3 + *wsum;
        }
    } else {                                           /* :311-338 batch two-pass */
        for (int a = 0; a < p; ++a) {
            double sacc = 0.0;
            for (int r = 0; r < n; ++r) sacc = sacc + x[(size_t)r * ldx + a] * ((w2 == -1.0) ? w[r] : w2);
            d[a] = sacc / wsum2;
        }
        for (int a = 0; a < p; ++a)
            for (int b = 0; b <= a; ++b) {
                double acc = 0.0;
                for (int r = 0; r < n; ++r) {
                    double xa = x[(size_t)r * ldx + a] - d[a];
                    double xb = (x[(size_t)r * ldx + b] - d[b]) * ((w2 == -1.0) ? w[r] : w2);
                    acc = acc + xa * xb;
                }
                double v = acc / (wsum2 - 1.0);
                A_(cmat, a, b, p) = v;
                if (a != b) A_(cmat, b, a, p) = v;
            }
        if (xmean) for (int a = 0; a < p; ++a) xmean[a] = d[a];
        if (wsum) *wsum = wsum2;
    }
    free(d);
}

/* ------------------------------------------------------------------ step primitives: MCMC_DRAM.F90 */

/* MCMC_alpha, MCMC_DRAM.F90:100-118 (nycol = 1); log_realmin = log(tiny(0d0)), mcmcprec.F90:36 */
#define MCX_LOG_REALMIN (-708.39641853226408)
double mcxo_alpha(double ss1, double pri1, double ss2, double pri2, double sigma2)
{
    double tst = -0.5 * ((ss2 - ss1) / sigma2 + (pri2 - pri1));
    if (tst >= 0.0) return 1.0;
    if (tst < MCX_LOG_REALMIN) return 0.0;
    return mcxm_exp(tst);
}

/* MCMC_reject, MCMC_DRAM.F90:140-155: u drawn only if 0 < alpha < 1; accept iff u <= alpha */
static int mcmc_reject(mcxo_chain *c, double alpha)
{
    if (alpha >= 1.0) return 0;
    if (alpha > 0.0) { double u = mcxo_uniform(&c->rng); if (u <= alpha) return 0; }
    return 1;
}

static double min1(double x) { return (1.0 < x) ? 1.0 : x; }   /* min(1.0, x), NaN propagates */

/* y = S x with S symmetric stored in the upper triangle (dsymv 'U', matutils.F90:180), then sum(y*x) */
static double quadform_sym_upper(int n, const double *S, const double *x)
{
    double q = 0.0;
    for (int i = 0; i < n; ++i) {
        double y = 0.0;
        for (int j = 0; j < n; ++j) {
            double sij = (j >= i) ? A_(S, i, j, n) : A_(S, j, i, n);
            y = (j == 0) ? sij * x[0] : fma(sij, x[j], y);
        }
        q = q + y * x[i];
    }
    return q;
}

/* sum((a - b)/sigma2) and sum(a/sigma2) over the response columns, the way the reference's array expressions reduce
 * (from 0, ascending); with one column 0 + x = x */
static double colsum_diff(const mcxo_chain *c, const double *a, const double *b)
{
    double s = 0.0;
    for (int j = 0; j < c->ny; ++j) s = s + (a[j] - b[j]) / c->sigma2v[j];
    return s;
}
/* MCMC_alpha, MCMC_DRAM.F90:100-118, vector form */
static double alpha_cols(const mcxo_chain *c, const double *ss1, double pri1, const double *ss2, double pri2)
{
    double tst = -0.5 * (colsum_diff(c, ss2, ss1) + (pri2 - pri1));
    if (tst >= 0.0) return 1.0;
    if (tst < MCX_LOG_REALMIN) return 0.0;
    return mcxm_exp(tst);
}

/* MCMC_DR_alpha13, MCMC_DRAM.F90:162-186 */
static double dr_alpha13(mcxo_chain *c, const double *oldpar, const double *ss1, double pri1,
                         const double *newpar, const double *ss2, double pri2, double alpha12,
                         const double *newpar2, const double *ss3, double pri3)
{
    int n = c->npar; double alpha32;
    if (alpha12 == 0.0) alpha32 = 0.0;
    else { double tst32 = -0.5 * (colsum_diff(c, ss2, ss3) + (pri2 - pri3)); alpha32 = min1(mcxm_exp(tst32)); }
    double l2 = -0.5 * (colsum_diff(c, ss3, ss1) + (pri3 - pri1));
    double *d1 = (double *)malloc(sizeof(double) * 2 * (size_t)n), *d2 = d1 + n;
    for (int i = 0; i < n; ++i) { d1[i] = newpar2[i] - newpar[i]; d2[i] = oldpar[i] - newpar[i]; }
    double q1 = -0.5 * (quadform_sym_upper(n, c->iC, d1) - quadform_sym_upper(n, c->iC, d2));
    free(d1);
    return min1(mcxm_exp(l2 + q1) * (1.0 - alpha32) / (1.0 - alpha12));
}

/* MCMC_updatesigma2, MCMC_DRAM.F90:192-206 */
static void updatesigma2(mcxo_chain *c, const double *ss)
{
    if (c->cfg.updatesigma != 0) {
        for (int j = 0; j < c->ny; ++j) {
            double g = mcxo_gamma(&c->rng, c->cfg.N0 / 2.0 + (double)c->nobsv[j] / 2.0, 2.0 / (c->cfg.N0 * c->S02 + ss[j]));
            c->sigma2v[j] = 1.0 / g;
        }
        c->sigma2 = c->sigma2v[0];
    }
}

/* MCMC_propose, MCMC_DRAM.F90:20-31 (usesvd == 0): newpar = oldpar + R'z; z kept for RAM */
static void propose(mcxo_chain *c, const double *oldpar, const double *R, double *newpar, double *zout)
{
    int n = c->npar;
    double *z = newpar;
    for (int i = 0; i < n; ++i) z[i] = mcxo_normal(&c->rng);
    if (zout) memcpy(zout, z, sizeof(double) * (size_t)n);
    if (R == c->R && c->last_u) memcpy(c->last_u, z, sizeof(double) * (size_t)n);
    if (c->cfg.usesvd) {                               /* matmulx(R, z): full dgemv 'N' (MCMC_DRAM.F90:27) */
        double *y = (double *)malloc(sizeof(double) * (size_t)n);
        mcxo_gemv(0, n, R, z, y);
        memcpy(z, y, sizeof(double) * (size_t)n);
        free(y);
    } else if (c->trmv_desc && R == c->R) mcxo_trmv_ut_desc(n, R, z);
    else mcxo_trmv_ut(n, R, z);
    for (int i = 0; i < n; ++i) newpar[i] = oldpar[i] + z[i];
}

/* MCMC_savechain 'memory' mode, MCMC_aux.F90:167-185 */
static void savechain(mcxo_chain *c, const double *par, const double *ss, int reject)
{
    int nc = c->npar + 1, ny = c->ny;
    if (reject) c->chain[(size_t)(c->chainind - 1) * nc + c->npar] += 1.0;
    else {
        c->chainind += 1;
        double *row = c->chain + (size_t)(c->chainind - 1) * nc;
        memcpy(row, par, sizeof(double) * (size_t)c->npar);
        row[c->npar] = 1.0;
        for (int j = 0; j < ny; ++j) c->sschain[(size_t)(c->chainind - 1) * (ny + 1) + j] = ss[j];
    }
    c->sschain[(size_t)(c->chainind - 1) * (ny + 1) + ny] = c->chain[(size_t)(c->chainind - 1) * nc + c->npar];
    if (c->cfg.updatesigma != 0) for (int j = 0; j < ny; ++j) c->s2chain[(size_t)(c->simuind - 1) * ny + j] = c->sigma2v[j];
}

/* ------------------------------------------------------------------ adaptation: MCMC_adapt.F90 */

/* covtor_svd (matutils.F90:378-453) / scam_svd (:583-653): dgesvd('A','N') of the covariance, singular values
 * floored at s(1)/condmax.  scam: R = U, std = sqrt(s).  otherwise R = U diag(sqrt(s)).
 * returns info: 0, -1 (values were floored), n (s(1) == 0). */
static int svd_factor(int n, const double *cmat, double condmax, int scam, double *R0, double *std)
{
    double *G = (double *)malloc(sizeof(double) * (size_t)n * n * 2 + sizeof(double) * n);
    double *V = G + (size_t)n * n, *sv = V + (size_t)n * n;
    memcpy(G, cmat, sizeof(double) * (size_t)n * n);
    mcxs_symsvd(n, G, V, sv);
    int info = 0;
    if (sv[0] == 0.0) { free(G); return n; }
    double tol = sv[0] / condmax;
    if (sv[n - 1] <= tol) {
        for (int i = 0; i < n; ++i) if (sv[i] < tol) sv[i] = tol;
        info = -1;
    }
    if (scam) {
        memcpy(R0, V, sizeof(double) * (size_t)n * n);
        for (int i = 0; i < n; ++i) std[i] = sqrt(sv[i]);
    } else {
        for (int i = 0; i < n; ++i) {
            double sq = sqrt(sv[i]);
            for (int k = 0; k < n; ++k) A_(R0, k, i, n) = sq * A_(V, k, i, n);        /* dscal */
        }
    }
    free(G);
    return info;
}

/* MCMC_calculate_R, MCMC_adapt.F90:181-230 */
int mcxo_calculate_R(mcxo_chain *c, double *cmat)
{
    int n = c->npar;
    double *R0 = (double *)malloc(sizeof(double) * (size_t)n * n);
    int info = 0;
    c->n_calcR += 1; c->svd_floored = 0;
    if (c->cfg.doscam) {                                           /* :189-200 */
        info = svd_factor(n, cmat, c->cfg.condmax, 1, R0, c->qcovstd);
        if (info > 0) { free(R0); c->info_last = info; return info; }
        info = 0;
        memcpy(c->R, R0, sizeof(double) * (size_t)n * n);
        free(R0);
        c->info_last = 0;
        return 0;
    }
    if (c->cfg.usesvd) {                                           /* :204-209 */
        info = svd_factor(n, cmat, c->cfg.condmax, 0, R0, NULL);
        if (info == -1) {                                          /* cmat = matmul(R0, transpose(R0)) */
            c->svd_floored = 1;
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    double acc = 0.0;
                    for (int k = 0; k < n; ++k) acc = fma(A_(R0, i, k, n), A_(R0, j, k, n), acc);
                    A_(cmat, i, j, n) = acc;
                }
            info = 0;
        }
    } else {
        memcpy(R0, cmat, sizeof(double) * (size_t)n * n);
        info = mcxo_potrf_u(n, R0);
    }
    if (info == 0) {
        double sq = sqrt((double)n);
        for (size_t k = 0; k < (size_t)n * n; ++k) c->R[k] = R0[k] * 2.4 / sq;
        if (c->cfg.dodr) {
            memcpy(c->iC, c->R, sizeof(double) * (size_t)n * n);
            int info2 = mcxo_potri_u(n, c->iC);                    /* on the upper triangle of R, also when R is U sqrt(s) */
            if (info2 != 0) { free(R0); return -2000 - info2; }       /* reference stops, :220-223 */
            for (size_t k = 0; k < (size_t)n * n; ++k) c->R2[k] = c->R[k] / c->cfg.drscale;
        }
    }
    free(R0);
    c->info_last = info;
    return info;
}

/* MCMC_adapt, MCMC_adapt.F90:12-174 (printing omitted) */
static int adapt(mcxo_chain *c, int simuind)
{
    const mcxo_cfg *g = &c->cfg;
    int n = c->npar, nc = n + 1;
    if (g->doadapt == 0 && g->doburnin == 0) return 0;
    if (g->adaptend > 0 && simuind > g->adaptend) return 0;
    int m1 = (g->adaptint != 0) ? (simuind % g->adaptint) : 1;
    int m2 = (g->badaptint != 0) ? (simuind % g->badaptint) : 1;
    if (m1 != 0 && m2 != 0) return 0;

    if (simuind < g->burnintime && g->doburnin != 0 && m2 == 0) {          /* :60-102 */
        double staypc = (double)c->stayed / (double)simuind;
        c->ad_istartind = c->chainind;
        double sf = g->scalefactor;
        if (staypc > 1.0 - g->scalelimit) {
            for (size_t k = 0; k < (size_t)n * n; ++k) c->R[k] = c->R[k] / sf;
            if (g->dodr) for (size_t k = 0; k < (size_t)n * n; ++k) { c->R2[k] = c->R2[k] / sf; c->iC[k] = c->iC[k] * sf * sf; }
            return 0;
        } else if (staypc < g->scalelimit) {
            for (size_t k = 0; k < (size_t)n * n; ++k) c->R[k] = c->R[k] * sf;
            if (g->dodr) for (size_t k = 0; k < (size_t)n * n; ++k) { c->R2[k] = c->R2[k] * sf; c->iC[k] = c->iC[k] / sf / sf; }
            return 0;
        } else if (g->greedy != 0) {
            c->chainwsum = (double)g->initcmatn;
            memcpy(c->chaincmat, c->cmat0, sizeof(double) * (size_t)n * n);
            memcpy(c->chainmean, c->par0, sizeof(double) * (size_t)n);
            double one = 1.0;
            mcxo_covmat(c->chainind, n, c->chain, nc, &one, 1, c->chaincmat, c->chainmean, &c->chainwsum, 1);
            c->ad_lastfreq = (int)c->chain[(size_t)(c->chainind - 1) * nc + n];
        }
        c->ad_lastind = c->chainind;
    } else if (simuind >= g->burnintime + g->adaptint + g->adapthist && g->doadapt != 0) {   /* :105-159 */
        if (simuind == g->burnintime + g->adaptint + g->adapthist) {
            c->chainwsum = (double)g->initcmatn;
            memcpy(c->chaincmat, c->cmat0, sizeof(double) * (size_t)n * n);
            memcpy(c->chainmean, c->par0, sizeof(double) * (size_t)n);
        }
        int nrows_max = c->chainind;
        double *w = (double *)malloc(sizeof(double) * (size_t)nrows_max);
        if (g->adapthist > 1) {                                             /* AP window :116-136 */
            int istart = c->chainind;
            int histsum = (int)c->chain[(size_t)(istart - 1) * nc + n];
            while (histsum < g->adapthist && istart > 1) {
                istart -= 1;
                histsum += (int)c->chain[(size_t)(istart - 1) * nc + n];
            }
            c->ad_istart = istart;
            int newfreq = (int)c->chain[(size_t)(istart - 1) * nc + n];
            int nr = c->chainind - istart + 1;
            for (int r = 0; r < nr; ++r) w[r] = c->chain[(size_t)(istart - 1 + r) * nc + n];
            w[0] = (double)(newfreq - histsum + g->adapthist);
            mcxo_covmat(nr, n, c->chain + (size_t)(istart - 1) * nc, nc, w, nr, c->chaincmat, c->chainmean, &c->chainwsum, 0);
        } else {                                                            /* AM :138-157 */
            int lastind = c->ad_lastind;
            int newfreq = (int)c->chain[(size_t)(lastind - 1) * nc + n];
            int nr = c->chainind - lastind + 1;
            for (int r = 0; r < nr; ++r) w[r] = c->chain[(size_t)(lastind - 1 + r) * nc + n];
            w[0] = (double)(newfreq - c->ad_lastfreq);
            c->ad_istart = lastind;
            mcxo_covmat(nr, n, c->chain + (size_t)(lastind - 1) * nc, nc, w, nr, c->chaincmat, c->chainmean, &c->chainwsum, 1);
            c->ad_lastfreq = (int)c->chain[(size_t)(c->chainind - 1) * nc + n];
            c->ad_lastind = c->chainind;
        }
        free(w);
    } else {
        return 0;
    }
    int info = mcxo_calculate_R(c, c->chaincmat);        /* info != 0: warning, old R kept :168-171 */
    if (info <= -1000) return info;
    return 0;
}

/* MCMC_adapt_ram, MCMC_run_ram.F90:104-179 (doupdate = 1 branch) */
static int adapt_ram(mcxo_chain *c, int simuind, const double *u, double alpha, double *work)
{
    const mcxo_cfg *g = &c->cfg;
    int n = c->npar;
    if (g->doadapt == 0) return 0;
    if (simuind < g->burnintime && g->doburnin != 0) return 0;
    double a = 1.0 / pow((double)(float)simuind, g->nuparam) * (alpha - g->alphatarget);
    double su = 0.0;
    for (int i = 0; i < n; ++i) su = su + u[i] * u[i];
    double *x = work, *cc = work + n, *ss = work + 2 * n;
    if (a >= 0.0) {
        for (int i = 0; i < n; ++i) x[i] = u[i] / su * a;
        mcxo_chud(n, c->R, x, cc, ss);
        c->trmv_desc = 0;
    } else {
        for (int i = 0; i < n; ++i) x[i] = -(u[i] / su * a);
        int info = mcxo_chdd(n, c->R, x, cc, ss);
        c->trmv_desc = (info == 0 && !c->cfg.usesvd);
        if (info != 0) {                                                    /* matutils.F90:719-722 stop */
            if (!c->ram_downdate_fail) c->ram_downdate_fail = simuind;
            if (!c->continue_on_downdate_fail) return -3000;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ chain object */

mcxo_chain *mcxo_chain_create(const mcxo_cfg *cfg, const mcxo_target *tgt, const double *par0,
                              const double *cmat0, double sigma2, int nobs, uint32_t seed, uint32_t chain_id)
{
    return mcxo_chain_create_ny(cfg, tgt, par0, cmat0, &sigma2, &nobs, 1, seed, chain_id);
}

mcxo_chain *mcxo_chain_create_ny(const mcxo_cfg *cfg, const mcxo_target *tgt, const double *par0,
                                 const double *cmat0, const double *sigma2v, const int *nobsv, int ny,
                                 uint32_t seed, uint32_t chain_id)
{
    if (ny < 1 || ny > MCXO_NYMAX) return NULL;
    if ((tgt->ny > 1 ? tgt->ny : 1) != ny) return NULL;
    const double sigma2 = sigma2v[0]; const int nobs = nobsv[0];
    mcxo_chain *c = (mcxo_chain *)calloc(1, sizeof *c);
    c->ny = ny;
    for (int j = 0; j < ny; ++j) { c->sigma2v[j] = sigma2v[j]; c->nobsv[j] = nobsv[j]; }
    int n = tgt->npar; size_t nn = (size_t)n * n, ns = (size_t)(cfg->nsimu > 0 ? cfg->nsimu : 1);
    c->cfg = *cfg; c->tgt = *tgt; c->npar = n;
    mcxo_rng_init(&c->rng, seed, chain_id);
    c->par0 = (double *)malloc(sizeof(double) * n); memcpy(c->par0, par0, sizeof(double) * n);
    c->cmat0 = (double *)malloc(sizeof(double) * nn); memcpy(c->cmat0, cmat0, sizeof(double) * nn);
    c->sigma2 = sigma2; c->nobs = nobs;
    c->R = (double *)calloc(nn, sizeof(double)); c->R2 = (double *)calloc(nn, sizeof(double)); c->iC = (double *)calloc(nn, sizeof(double));
    c->chaincmat = (double *)malloc(sizeof(double) * nn); c->chainmean = (double *)malloc(sizeof(double) * n);
    c->chain = (double *)calloc(ns * (size_t)(n + 1), sizeof(double));
    c->sschain = (double *)calloc(ns * (size_t)(ny + 1), sizeof(double));
    c->s2chain = (double *)calloc(ns * (size_t)ny, sizeof(double));
    c->accepted = (uint8_t *)calloc(ns, 1);
    c->alpha_trace = (double *)calloc(ns, sizeof(double));
    c->oldpar = (double *)malloc(sizeof(double) * n);
    c->qcovstd = (double *)calloc((size_t)n, sizeof(double));
    c->last_u = (double *)calloc((size_t)n, sizeof(double));
    c->ad_istart = 1; c->ad_istartind = 1; c->ad_lastind = 1; c->ad_lastfreq = 0;
    /* MCMC_init.F90:99-116 */
    memcpy(c->chaincmat, cmat0, sizeof(double) * nn);
    memcpy(c->chainmean, par0, sizeof(double) * n);
    c->chainwsum = (double)cfg->initcmatn;
    double *tmp = (double *)malloc(sizeof(double) * nn); memcpy(tmp, cmat0, sizeof(double) * nn);
    int info = mcxo_calculate_R(c, tmp);
    free(tmp);
    if (info != 0) { mcxo_chain_free(c); return NULL; }   /* 'could not factor the initial covariance' */
    c->S02 = (cfg->S02 <= 0.0) ? sigma2 : cfg->S02;
    c->chainind = 0; c->simuind = 1;
    return c;
}

void mcxo_chain_free(mcxo_chain *c)
{
    if (!c) return;
    free(c->par0); free(c->cmat0); free(c->R); free(c->R2); free(c->iC); free(c->chaincmat); free(c->chainmean);
    free(c->chain); free(c->sschain); free(c->s2chain); free(c->accepted); free(c->alpha_trace); free(c->oldpar); free(c->qcovstd); free(c->last_u);
    free(c);
}

/* ------------------------------------------------------------------ MCMC_run1 / MCMC_run1_er: the arithmetic of one invocation
 * The one-evaluation-per-invocation protocol (MCMC_run1.F90:31-256, MCMC_run1_er.F90:28-234) keeps its state in files; its
 * state machine is restated in oracle/run1.py, the three pieces of arithmetic here, on a freshly created chain (MCMC_init:
 * R, R2, iC of cmat0; the stream of this invocation). */
/* MCMC_run1.F90:131-143: alpha = MCMC_DR_alpha13(oldpar2, .., oldpar1, .., alpha12, newpar, ..) in DR stage 2, else
 * MCMC_alpha(oldpar1 -> newpar); returns reject = MCMC_reject(alpha) */
int mcxo_run1_decide(mcxo_chain *c, int drstage, const double *oldpar2, const double *ssprev2, double sspri2,
                     const double *oldpar1, const double *ssprev1, double sspri1, double alpha12,
                     const double *newpar, const double *ss, double sspri, double *alpha_out)
{
    double alpha;
    if (drstage > 1 && c->cfg.dodr)
        alpha = dr_alpha13(c, oldpar2, ssprev2, sspri2, oldpar1, ssprev1, sspri1, alpha12, newpar, ss, sspri);
    else
        alpha = alpha_cols(c, ssprev1, sspri1, ss, sspri);
    *alpha_out = alpha;
    return mcmc_reject(c, alpha);
}
/* MCMC_run1.F90:185-189: newpar = MCMC_propose(from, R2) in DR stage 2, else (from, R) */
void mcxo_run1_propose(mcxo_chain *c, int stage, const double *from, double *newpar)
{
    propose(c, from, (stage > 1 && c->cfg.dodr) ? c->R2 : c->R, newpar, NULL);
}
/* MCMC_sscrit, MCMC_DRAM.F90:124-135 (called at MCMC_run1_er.F90:168) */
double mcxo_run1_sscrit(mcxo_chain *c, const double *ssprev1, double sspri1)
{
    double u = mcxo_uniform(&c->rng);
    double s1 = 0.0;
    for (int j = 0; j < c->ny; ++j) s1 = s1 + ssprev1[j] / c->sigma2v[j];
    return -2.0 * mcxm_log(u) + s1 + sspri1;
}

/* MCMC_run (MCMC_run.F90:12-114) and MCMC_run_ram (MCMC_run_ram.F90:13-83).
 * First call does the pre-loop part; continues from simuind+1 up to `upto` (<= nsimu). */
int mcxo_chain_run(mcxo_chain *c, int upto)
{
    const mcxo_cfg *g = &c->cfg;
    int n = c->npar;
    if (upto > g->nsimu) upto = g->nsimu;
    double *newpar = (double *)malloc(sizeof(double) * (size_t)n * 6);
    double *newpar2 = newpar + n, *z = newpar + 2 * n, *work = newpar + 3 * n;
    int rc = 0;
    const int ny = c->ny;
    double ss2v[MCXO_NYMAX], ss3v[MCXO_NYMAX];
#define SS1_TAKE(src) do { for (int j_ = 0; j_ < ny; ++j_) c->ss1v[j_] = (src)[j_]; c->ss1 = c->ss1v[0]; } while (0)
    if (c->chainind == 0) {
        memcpy(c->oldpar, c->par0, sizeof(double) * n);
        c->sspri1 = mcxo_priorfun(&c->tgt, c->oldpar);
        mcxo_ssfun_cols(&c->tgt, c->oldpar, c->ss1v); c->ss1 = c->ss1v[0];
        c->simuind = 1;
        savechain(c, c->oldpar, c->ss1v, 0);
        c->accepted[0] = 1;
        c->alpha12 = 0.0;
    }
    for (int i = c->simuind + 1; i <= upto; ++i) {
        c->simuind = i;
        int reject, inb;
        double pri2 = 0, pri3 = 0;
        for (int j_ = 0; j_ < ny; ++j_) { ss2v[j_] = 0.0; ss3v[j_] = 0.0; }
        if (g->method == MCXO_METHOD_SCAM) {                        /* MCMC_run_scam.F90:38-88 */
            int rejall = 1;
            double *rot = work;                                      /* work has 3n doubles */
            for (int j = 0; j < n; ++j) {
                /* MCMC_propose_sc (:94-117): rotate, perturb component j, rotate back */
                mcxo_gemv(1, n, c->R, c->oldpar, rot);
                double zj = mcxo_normal(&c->rng) * c->qcovstd[j];
                rot[j] = rot[j] + zj;
                mcxo_gemv(0, n, c->R, rot, newpar);
                c->nprop++;
                inb = mcxo_checkbounds(&c->tgt, newpar);
                if (!inb) { if (!g->dodr) c->bndstayed++; c->alpha12 = 0.0; reject = 1; }
                else {
                    pri2 = mcxo_priorfun(&c->tgt, newpar); mcxo_ssfun_cols(&c->tgt, newpar, ss2v);
                    c->alpha12 = alpha_cols(c, c->ss1v, c->sspri1, ss2v, pri2);
                    reject = mcmc_reject(c, c->alpha12);
                }
                if (!reject) { SS1_TAKE(ss2v); c->sspri1 = pri2; memcpy(c->oldpar, newpar, sizeof(double) * n); rejall = 0; }
            }
            reject = rejall;
            c->alpha_trace[i - 1] = c->alpha12;
            if (reject) c->stayed++;
            c->accepted[i - 1] = (uint8_t)!reject;
            updatesigma2(c, c->ss1v);
            savechain(c, c->oldpar, c->ss1v, reject);
            rc = adapt(c, i);
            if (rc != 0) break;
            continue;
        }
        if (g->method == MCXO_METHOD_ER) {                          /* MCMC_run_er.F90:46-104 */
            propose(c, c->oldpar, c->R, newpar, NULL);
            c->nprop++;
            inb = mcxo_checkbounds(&c->tgt, newpar);
            if (!inb) { c->bndstayed++; reject = 1; }
            else {
                double u = mcxo_uniform(&c->rng);                    /* MCMC_sscrit, MCMC_DRAM.F90:124-135 */
                double s1 = 0.0;                                     /* sum(ss1/sigma2) */
                for (int j_ = 0; j_ < ny; ++j_) s1 = s1 + c->ss1v[j_] / c->sigma2v[j_];
                double sscrit = -2.0 * mcxm_log(u) + s1 + c->sspri1;
                pri2 = mcxo_priorfun(&c->tgt, newpar);
                if (pri2 >= sscrit) { reject = 1; c->erstayed++; }
                else {
                    sscrit = c->sigma2v[0] * (sscrit - pri2);       /* MCMC_run_er.F90:72 "problem here if nycol > 1" */
                    mcxo_ssfun_cols(&c->tgt, newpar, ss2v);          /* default ssfunction_er: no early exit */
                    double s2 = 0.0;                                 /* sum(ss2) */
                    for (int j_ = 0; j_ < ny; ++j_) s2 = s2 + ss2v[j_];
                    reject = (s2 >= sscrit) ? 1 : 0;
                }
            }
            c->alpha_trace[i - 1] = c->alpha12;
            if (reject) c->stayed++;
            else { SS1_TAKE(ss2v); c->sspri1 = pri2; memcpy(c->oldpar, newpar, sizeof(double) * n); }
            c->accepted[i - 1] = (uint8_t)!reject;
            updatesigma2(c, c->ss1v);
            savechain(c, c->oldpar, c->ss1v, reject);
            rc = adapt(c, i);
            if (rc != 0) break;
            continue;
        }
        if (g->method == MCXO_METHOD_RAM) {
            propose(c, c->oldpar, c->R, newpar, z);
            c->nprop++;
            inb = mcxo_checkbounds(&c->tgt, newpar);
            if (!inb) { c->bndstayed++; reject = 1; }              /* alpha12 left stale: MCMC_run_ram.F90:52-54 */
            else {
                pri2 = mcxo_priorfun(&c->tgt, newpar); mcxo_ssfun_cols(&c->tgt, newpar, ss2v);
                c->alpha12 = alpha_cols(c, c->ss1v, c->sspri1, ss2v, pri2);
                reject = mcmc_reject(c, c->alpha12);
            }
        } else {
            propose(c, c->oldpar, c->R, newpar, NULL);
            c->nprop++;
            inb = mcxo_checkbounds(&c->tgt, newpar);
            if (!inb) {
                if (!g->dodr) c->bndstayed++;
                for (int j_ = 0; j_ < ny; ++j_) ss2v[j_] = DBL_MAX;
                pri2 = DBL_MAX; c->alpha12 = 0.0; reject = 1;
            } else {
                pri2 = mcxo_priorfun(&c->tgt, newpar); mcxo_ssfun_cols(&c->tgt, newpar, ss2v);
                c->alpha12 = alpha_cols(c, c->ss1v, c->sspri1, ss2v, pri2);
                reject = mcmc_reject(c, c->alpha12);
            }
            if (reject && g->dodr) {                                /* MCMC_run.F90:65-91 */
                c->drtries++;
                propose(c, c->oldpar, c->R2, newpar2, NULL);
                c->nprop++;
                inb = mcxo_checkbounds(&c->tgt, newpar2);
                if (!inb) { c->bndstayed++; reject = 1; }
                else {
                    pri3 = mcxo_priorfun(&c->tgt, newpar2); mcxo_ssfun_cols(&c->tgt, newpar2, ss3v);
                    double a13 = dr_alpha13(c, c->oldpar, c->ss1v, c->sspri1, newpar, ss2v, pri2, c->alpha12, newpar2, ss3v, pri3);
                    reject = mcmc_reject(c, a13);
                    if (!reject) { c->draccepted++; memcpy(newpar, newpar2, sizeof(double) * n); for (int j_ = 0; j_ < ny; ++j_) ss2v[j_] = ss3v[j_]; pri2 = pri3; }
                }
            }
        }
        c->alpha_trace[i - 1] = c->alpha12;
        if (reject) c->stayed++;
        else { SS1_TAKE(ss2v); c->sspri1 = pri2; memcpy(c->oldpar, newpar, sizeof(double) * n); }
        c->accepted[i - 1] = (uint8_t)!reject;
        updatesigma2(c, c->ss1v);
        savechain(c, c->oldpar, c->ss1v, reject);
        if (g->method == MCXO_METHOD_RAM) rc = adapt_ram(c, i, z, c->alpha12, work);
        else rc = adapt(c, i);
        if (rc != 0) break;
    }
    free(newpar);
    return rc;
}
