/*
 * oracle/ref/dgesvd_logger.c -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * dgesvd_ interposed in front of the LAPACK the reference links (MKL here): forwards every call to the real routine
 * and appends what it returned -- n, the singular values and U -- to the file named by MCX_SVD_LOG.  Linked only into
 * oracle/_ref/mcxref_mkllog.  tests/test_oracle_svd.py feeds those factors to the oracle in place of its own (pinned
 * Jacobi) SVD, so that the rest of MCMC_run_scam / covtor_svd (MCMC_run_scam.F90:38-138, matutils.F90:378-453,583-653)
 * is checked against the MKL-linked Fortran chain decision by decision, independently of the pinned routine.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

typedef void (*dgesvd_fn)(const char *, const char *, const int *, const int *, double *, const int *, double *, double *,
                          const int *, double *, const int *, double *, const int *, int *, long, long);

void dgesvd_(const char *jobu, const char *jobvt, const int *m, const int *n, double *a, const int *lda,
             double *s, double *u, const int *ldu, double *vt, const int *ldvt, double *work, const int *lwork,
             int *info, long l1, long l2)
{
    static dgesvd_fn real = NULL;
    if (!real) real = (dgesvd_fn)dlsym(RTLD_NEXT, "dgesvd_");
    if (!real) { fprintf(stderr, "dgesvd_logger: no dgesvd_ behind this one\n"); abort(); }
    real(jobu, jobvt, m, n, a, lda, s, u, ldu, vt, ldvt, work, lwork, info, l1, l2);
    const char *path = getenv("MCX_SVD_LOG");
    if (!path) return;
    FILE *f = fopen(path, "ab");
    if (!f) return;
    const int nn = *n;
    fwrite(&nn, sizeof(int), 1, f);
    fwrite(info, sizeof(int), 1, f);
    fwrite(s, sizeof(double), (size_t)nn, f);
    for (int j = 0; j < nn; ++j) fwrite(u + (size_t)j * *ldu, sizeof(double), (size_t)nn, f);     /* column j of U */
    fclose(f);
}
