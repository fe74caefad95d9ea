/*
 * oracle/ref/dgesvd_shim.c -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * dgesvd for the SCAM / SVD-proposal fixtures: the reference's two call sites (matutils.F90:409, :615) are
 * dgesvd('A','N', n, n, cov, n, s, u, n, u, n, work, lwork, info) on a symmetric PSD covariance matrix.  LAPACK
 * is not part of the reference and is not pinned by it; this file answers those calls with the pinned one-sided
 * Jacobi routine of oracle/mcx_svd.h so that the Fortran chain, the oracle and the GPU share one set of singular
 * vectors (signs included).  It is linked only into oracle/_ref/mcxref_svd; oracle/_ref/mcxref keeps MKL's dgesvd.
 */
#include <stdlib.h>
#include <string.h>
#include "../mcx_svd.h"

void dgesvd_(const char *jobu, const char *jobvt, const int *m, const int *n, double *a, const int *lda,
             double *s, double *u, const int *ldu, double *vt, const int *ldvt, double *work, const int *lwork,
             int *info)
{
    (void)jobvt; (void)vt; (void)ldvt; (void)work; (void)lwork;
    const int nn = *n;
    if (*m != nn || (*jobu != 'A' && *jobu != 'a')) { *info = -1; return; }
    double *G = (double *)malloc(sizeof(double) * (size_t)nn * nn), *V = (double *)malloc(sizeof(double) * (size_t)nn * nn);
    for (int j = 0; j < nn; ++j) for (int i = 0; i < nn; ++i) G[(size_t)i + (size_t)j * nn] = a[(size_t)i + (size_t)j * *lda];
    mcxs_symsvd(nn, G, V, s);
    for (int j = 0; j < nn; ++j) for (int i = 0; i < nn; ++i) u[(size_t)i + (size_t)j * *ldu] = V[(size_t)i + (size_t)j * nn];
    free(G); free(V);
    *info = 0;
}
