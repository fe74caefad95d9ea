/*
 * oracle/mcx_targets.h -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * The "user" side of the runs: ssfunction (-2 log likelihood), the default
 * Gaussian priorfun (priorfun.f90:96-100) and a box checkbounds.  These are
 * user code in mcmcf90 (link-time callbacks, external_inc.h:4-33), so their
 * arithmetic is ours to define; one definition is shared by the oracle and by
 * the driver linked to the real reference (oracle/ref/user_target.c), and the HIP
 * engine restates it on the device.
 *
 *   gauss   : testcases/mcmcrun4.F90:47 pattern, ss = (th-mu)' Lam (th-mu)
 *   banana  : ss = th1^2/100 + (th2 + b th1^2 - 100 b)^2 + sum_{k>=3} th_k^2
 *   expdata : testcases/mcmcrun.F90:89,104 pattern, ss = sum (y - th1 exp(-th2 x))^2
 */
#ifndef MCX_ORACLE_TARGETS_H
#define MCX_ORACLE_TARGETS_H
#include "mcx_math.h"

static inline double mcxt_ss_gauss(int d, const double *th, const double *mu, const double *lam)
{
    /* ss = v' Lam v, v = th - mu.  The target is build-defined (it is not part of mcmcf90), so is its operation order:
     *   y_i = sum_j lam(i,j) v_j          one fma chain ascending in j, from 0;
     *   rows in blocks of 16; in block t four interleaved partial chains q_k, k = 0..3, over the rows
     *   16t + k + 4r, r = 0..3: q_k = y_i v_i for r = 0, then fma(y_i, v_i, q_k);
     *   ss = q_0 of block 0, then ss = ss + q_k, k ascending inside a block, blocks ascending.
     * This is the order in which a 16x16 f64 matrix-core tile leaves y in registers (each q_k is lane-local), and it
     * costs a lane-per-chain kernel nothing. */
    double ss = 0.0;
    for (int t0 = 0; t0 < d; t0 += 16)
        for (int k = 0; k < 4 && t0 + k < d; ++k) {
            double q = 0.0;
            for (int r = 0; r < 4; ++r) {
                const int i = t0 + k + 4 * r;
                if (i >= d) break;
                const double *row = lam + (long)i * d;
                double y = 0.0;
                for (int j = 0; j < d; ++j) y = fma(row[j], th[j] - mu[j], y);
                const double vi = th[i] - mu[i];
                q = (r == 0) ? y * vi : fma(y, vi, q);
            }
            ss = (t0 == 0 && k == 0) ? q : ss + q;
        }
    return ss;
}

static inline double mcxt_ss_banana(int d, const double *th, double b)
{
    double t1 = th[0] * th[0];
    double q = fma(b, t1, th[1]) - 100.0 * b;
    double ss = fma(q, q, t1 / 100.0);
    for (int k = 2; k < d; ++k) ss = fma(th[k], th[k], ss);
    return ss;
}

static inline double mcxt_ss_expdata(const double *th, int n, const double *x, const double *y)
{
    double ss = 0.0;
    for (int i = 0; i < n; ++i) {
        double r = y[i] - th[0] * mcxm_exp(-(th[1] * x[i]));
        ss = fma(r, r, ss);
    }
    return ss;
}

/* ny response columns (nycol > 1, MCMC_DRAM.F90:100-118: one sigma2 per column): column j is th0 * exp(-th(1+j) x)
 * against y + j*n, so npar = 1 + ny; column 0 alone is mcxt_ss_expdata */
static inline void mcxt_ss_expdata_cols(const double *th, int n, const double *x, const double *y, int ny, double *ss)
{
    for (int j = 0; j < ny; ++j) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            double r = y[(long)j * n + i] - th[0] * mcxm_exp(-(th[1 + j] * x[i]));
            s = fma(r, r, s);
        }
        ss[j] = s;
    }
}

/* priorfun.f90:96-100: sum(((theta-mu)/sig)**2, mask=sig>0); library Fortran, no fma */
static inline double mcxt_prior(int d, const double *th, const double *pmu, const double *psig)
{
    double p = 0.0;
    if (!pmu || !psig) return 0.0;
    for (int i = 0; i < d; ++i)
        if (psig[i] > 0.0) { double t = (th[i] - pmu[i]) / psig[i]; p = p + t * t; }
    return p;
}

static inline int mcxt_inbounds(int d, const double *th, const double *lo, const double *hi)
{
    for (int i = 0; i < d; ++i) {
        if (lo && !(th[i] > lo[i])) return 0;
        if (hi && !(th[i] < hi[i])) return 0;
    }
    return 1;
}
#endif
