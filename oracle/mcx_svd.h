/*
 * oracle/mcx_svd.h -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * Singular value decomposition of a symmetric positive semi-definite matrix, the only use the reference
 * makes of LAPACK dgesvd('A','N') (matutils.F90:409 covtor_svd, :615 scam_svd, both on a covariance matrix).
 * The reference leaves LAPACK unpinned, and an SVD's singular-vector signs (and the basis inside a cluster of
 * equal singular values) are implementation-defined, so the accept/reject sequence of a SCAM run depends on
 * which LAPACK is linked.  We pin the routine as the one-sided Jacobi method (Hestenes 1958; de Rijk 1989
 * row-cyclic ordering), stated here operation by operation; the engine repeats it on the host/device, and
 * oracle/ref/dgesvd_shim.c puts the same routine under the real Fortran reference for the SCAM fixtures
 * (an MKL-linked run of the same case is kept as a statistical cross-check, tests/test_oracle_scam.py).
 *
 *   G <- A (n x n, column-major), V <- I
 *   sweeps: for p < q: alpha = g_p.g_p, beta = g_q.g_q, gamma = g_p.g_q -- every dot product of the routine is EIGHT partial
 *           fma chains over the rows k = j, j + 8, j + 16, ... (j = 0..7), added pairwise,
 *           ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)): a fixed order that eight lanes can run side by side (round 2;
 *           one chain over all rows before, which left a GPU nothing to spread over lanes: DESIGN.md section 10);
 *           skip if gamma == 0 or |gamma| <= 1e-15 sqrt(alpha beta);
 *           zeta = (beta-alpha)/(2 gamma); t = sign(zeta)/(|zeta| + sqrt(1+zeta^2)); c = 1/sqrt(1+t^2); s = c t;
 *           (g_p, g_q) <- (c g_p - s g_q, s g_p + c g_q), same for (v_p, v_q);
 *           until a sweep rotates nothing (at most 60 sweeps)
 *   s_j = sqrt(g_j.g_j); columns ordered by descending s_j (selection, first maximum wins); U = V.
 * For a PSD matrix A = V diag(s) V'.
 */
#ifndef MCX_ORACLE_SVD_H
#define MCX_ORACLE_SVD_H
#include <math.h>
#include <stddef.h>

#define MCXS_TOL 1e-15
#define MCXS_MAXSWEEP 60

/* the routine's dot products: eight partial chains by row index mod 8, then the pairwise tree */
static inline double mcxs_tree8(const double *p) { return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7])); }
static inline void mcxs_dot3(int n, const double *gp, const double *gq, double *alpha, double *beta, double *gamma)
{
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, b[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, g[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < n; ++k) {
        const int j = k & 7;
        a[j] = fma(gp[k], gp[k], a[j]); b[j] = fma(gq[k], gq[k], b[j]); g[j] = fma(gp[k], gq[k], g[j]);
    }
    *alpha = mcxs_tree8(a); *beta = mcxs_tree8(b); *gamma = mcxs_tree8(g);
}
static inline double mcxs_sumsq(int n, const double *g)
{
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < n; ++k) a[k & 7] = fma(g[k], g[k], a[k & 7]);
    return mcxs_tree8(a);
}

/* G: in A, destroyed; V: out singular vectors (columns), s: out singular values, all column-major n x n / n */
static inline int mcxs_symsvd(int n, double *G, double *V, double *s)
{
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) V[(size_t)i + (size_t)j * n] = (i == j) ? 1.0 : 0.0;
    int sweep;
    for (sweep = 0; sweep < MCXS_MAXSWEEP; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double *gp = G + (size_t)p * n, *gq = G + (size_t)q * n;
                double alpha, beta, gamma;
                mcxs_dot3(n, gp, gq, &alpha, &beta, &gamma);
                if (gamma == 0.0) continue;
                if (fabs(gamma) <= MCXS_TOL * sqrt(alpha * beta)) continue;
                rotated = 1;
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int k = 0; k < n; ++k) {
                    double a = gp[k], b = gq[k];
                    gp[k] = c * a - sn * b; gq[k] = sn * a + c * b;
                }
                double *vp = V + (size_t)p * n, *vq = V + (size_t)q * n;
                for (int k = 0; k < n; ++k) {
                    double a = vp[k], b = vq[k];
                    vp[k] = c * a - sn * b; vq[k] = sn * a + c * b;
                }
            }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < n; ++j) {
        s[j] = sqrt(mcxs_sumsq(n, G + (size_t)j * n));
    }
    for (int i = 0; i < n - 1; ++i) {                 /* descending order, first maximum wins */
        int m = i;
        for (int j = i + 1; j < n; ++j) if (s[j] > s[m]) m = j;
        if (m != i) {
            double ts = s[i]; s[i] = s[m]; s[m] = ts;
            for (int k = 0; k < n; ++k) {
                double tv = V[(size_t)k + (size_t)i * n]; V[(size_t)k + (size_t)i * n] = V[(size_t)k + (size_t)m * n]; V[(size_t)k + (size_t)m * n] = tv;
            }
        }
    }
    return sweep;
}
#endif
