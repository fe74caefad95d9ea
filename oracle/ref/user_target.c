/*
 * oracle/ref/user_target.c -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * The user-callback side (ssfunction / checkbounds, external_inc.h:12-33) of the
 * driver program that is linked to the REAL Fortran reference in oracle/_ref/.
 * mcmcf90 leaves these to the user, so this is our code, not reference code; it
 * evaluates the same built-in targets as the oracle (oracle/mcx_targets.h) so
 * that any difference between a reference run and the oracle comes from the
 * sampler, not from the likelihood.
 *
 * The target is read once from ./mcxtarget.txt (written by oracle/gen_golden.py):
 *   kind npar
 *   gauss  : mu[npar], lam[npar*npar] row-major
 *   banana : b
 *   expdata: ndata, x[ndata], y[ndata]
 *   expdata with several response columns (kind 3): ny, ndata, x[ndata], y[ny*ndata] column after column
 *   nbounds (0 or npar) then lo[npar], hi[npar]
 * Numbers are C99 hex floats (exact).
 */
#include <stdio.h>
#include <stdlib.h>
#include "../mcx_targets.h"

static int g_loaded = 0, g_kind, g_npar, g_ndata, g_nb, g_ny = 1;
static double *g_mu, *g_lam, g_b, *g_x, *g_y, *g_lo, *g_hi;

double mcxref_ss(const double *theta, int npar);

static double rd(FILE *f)
{
    char buf[128];
    if (fscanf(f, "%127s", buf) != 1) { fprintf(stderr, "user_target: short mcxtarget.txt\n"); exit(2); }
    return strtod(buf, NULL);
}
static double *rdv(FILE *f, int n)
{
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) v[i] = rd(f);
    return v;
}
static void load(void)
{
    if (g_loaded) return;
    FILE *f = fopen("mcxtarget.txt", "r");
    if (!f) { fprintf(stderr, "user_target: cannot open mcxtarget.txt\n"); exit(2); }
    g_kind = (int)rd(f); g_npar = (int)rd(f);
    if (g_kind == 0) { g_mu = rdv(f, g_npar); g_lam = rdv(f, g_npar * g_npar); }
    else if (g_kind == 1) g_b = rd(f);
    else if (g_kind == 2) { g_ndata = (int)rd(f); g_x = rdv(f, g_ndata); g_y = rdv(f, g_ndata); }
    else if (g_kind == 3) { g_ny = (int)rd(f); g_ndata = (int)rd(f); g_x = rdv(f, g_ndata); g_y = rdv(f, g_ny * g_ndata); }
    g_nb = (int)rd(f);
    if (g_nb > 0) { g_lo = rdv(f, g_nb); g_hi = rdv(f, g_nb); }
    fclose(f);
    g_loaded = 1;
}

/* ssfunction(theta, npar, ny): ny values */
void mcxref_ss_cols(const double *theta, int npar, int ny, double *ss)
{
    load();
    if (npar != g_npar || ny != g_ny) { fprintf(stderr, "user_target: npar / ny mismatch\n"); exit(2); }
    if (g_kind == 3) mcxt_ss_expdata_cols(theta, g_ndata, g_x, g_y, g_ny, ss);
    else ss[0] = mcxref_ss(theta, npar);
}

double mcxref_ss(const double *theta, int npar)
{
    load();
    if (npar != g_npar) { fprintf(stderr, "user_target: npar mismatch\n"); exit(2); }
    if (g_kind == 0) return mcxt_ss_gauss(npar, theta, g_mu, g_lam);
    if (g_kind == 1) return mcxt_ss_banana(npar, theta, g_b);
    return mcxt_ss_expdata(theta, g_ndata, g_x, g_y);
}

int mcxref_inbounds(const double *theta, int npar)
{
    load();
    if (g_nb <= 0) return 1;
    return mcxt_inbounds(npar, theta, g_lo, g_hi);
}
