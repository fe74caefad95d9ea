/*
 * oracle/mcx_oracle.h -- TEST INFRASTRUCTURE. Not part of the product path.
 *
 * Plain-C restatement of the mcmcf90 sampling hot path (MCMC_run / MCMC_run_ram,
 * MCMC_adapt, MCMC_DRAM step primitives, mcmcrand, covmat, LINPACK dchud/dchdd)
 * for ONE chain, exactly as the Fortran reference runs it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Pinned against: the real reference compiled from /root/reference
 * (oracle/Makefile -> oracle/_ref/) driven by the same Philox stream
 * (oracle/ref/rng_interpose.c); fixtures in tests/golden/ (oracle/gen_golden.py).
 */
#ifndef MCX_ORACLE_H
#define MCX_ORACLE_H
#include <stdint.h>
#include "mcx_rng.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { MCXO_METHOD_DRAM = 0, MCXO_METHOD_RAM = 1, MCXO_METHOD_SCAM = 2, MCXO_METHOD_ER = 3 };
enum { MCXO_TARGET_GAUSS = 0, MCXO_TARGET_BANANA = 1, MCXO_TARGET_EXPDATA = 2 };

/* namelist /mcmc/ numeric control variables (mcmcinit.F90:74-82), defaults :184-230 */
typedef struct mcxo_cfg {
    int nsimu, doadapt, doburnin, adaptint, adapthist, badaptint, adaptend, initcmatn;
    int burnintime, greedy, updatesigma, method;
    double scalelimit, scalefactor, drscale, N0, S02, condmax, alphatarget, nuparam;
    /* derived by mcxo_cfg_check (mcmcinit.F90:235-368) */
    int dodr, doscam, usesvd;
} mcxo_cfg;

void mcxo_cfg_defaults(mcxo_cfg *c);
int  mcxo_cfg_check(mcxo_cfg *c);   /* 0 ok, <0 = the reference would stop */

/* Built-in targets: the "user" ssfunction / priorfun / checkbounds of the runs.
 * The same definitions (same operation order) are compiled into the reference
 * driver (oracle/ref/user_target.c) and into the HIP engine. */
typedef struct mcxo_target {
    int kind, npar;
    const double *mu;      /* gauss: mean[npar] */
    const double *lam;     /* gauss: precision, row-major lam[i*npar+j] */
    double banana_b;       /* banana: twist */
    int ndata;             /* expdata: y = th1*exp(-th2*x) */
    const double *xdata, *ydata;
    const double *lo, *hi;            /* box bounds, in-bounds iff lo<th<hi; NULL = none */
    const double *pri_mu, *pri_sig;   /* default Gaussian priors, sig<=0 = flat; NULL = none */
    int ny;                           /* response columns of ssfunction (nycol); 0 or 1 = one; > 1: expdata only, ydata ny x ndata */
} mcxo_target;

#define MCXO_NYMAX 32
double mcxo_ssfun(const mcxo_target *t, const double *theta);               /* column 0 */
void   mcxo_ssfun_cols(const mcxo_target *t, const double *theta, double *ss); /* all ny columns */
double mcxo_priorfun(const mcxo_target *t, const double *theta);
int    mcxo_checkbounds(const mcxo_target *t, const double *theta);

/* One chain, all module-level state of mcmcmod (mcmc.F90:28-60) */
typedef struct mcxo_chain {
    mcxo_cfg cfg;
    mcxo_target tgt;
    mcxo_rng rng;
    int npar;
    double *par0, *cmat0;                 /* cmat0 col-major npar x npar */
    double sigma2, S02; int nobs;         /* nycol = 1 */
    double *R, *R2, *iC;                  /* col-major npar x npar, upper used */
    double *chaincmat, *chainmean; double chainwsum;
    double *chain;                        /* row-major [nsimu][npar+1], last col = repeat count */
    double *sschain;                      /* [nsimu][2] */
    double *s2chain;                      /* [nsimu] (row simuind-1) */
    uint8_t *accepted;                    /* [nsimu], accepted[i-1] = 1 if iteration i moved */
    double *alpha_trace;                  /* [nsimu] alpha12 of each iteration (diagnostic) */
    int simuind, chainind;
    int stayed, bndstayed, draccepted, drtries;
    uint64_t nprop;                       /* proposals evaluated (stage 1 + stage 2) */
    /* function-static state of MCMC_adapt (MCMC_adapt.F90:15,19) */
    int ad_istart, ad_istartind, ad_lastind, ad_lastfreq;
    int info_last;                        /* last info of calculate_R */
    int ram_downdate_fail;                /* reference would STOP (matutils.F90:719-722) */
    /* run state */
    double *oldpar; double ss1, sspri1, alpha12;
    /* 0: stop at a failed downdate like the reference (matutils.F90:719-722); 1: record it and go on with R
     * untouched, which is what the multi-chain engine does (one bad chain must not end a million-chain run) */
    int continue_on_downdate_fail;
    double *qcovstd;                      /* SCAM: sqrt of the singular values (mcmc.F90:37) */
    int erstayed;                         /* early rejection on the prior (mcmc.F90:48) */
    /* nycol columns (mcmc.F90:30-33: sigma2(:), nobs(:); ss vectors): the scalars ss1 / sigma2 / nobs above mirror column 0;
     * sschain is [nsimu][ny+1] (ss columns, then the repeat count), s2chain [nsimu][ny] */
    int ny;
    double ss1v[MCXO_NYMAX], sigma2v[MCXO_NYMAX];
    int nobsv[MCXO_NYMAX];
    /* 1 after a successful Cholesky downdate of MCMC_adapt_ram: the next proposal's R'z accumulates from the diagonal
     * up (mcxo_trmv_ut_desc), which is the order DCHDD leaves the columns in; otherwise ascending (mcxo_trmv_ut) */
    int trmv_desc;
    /* the normal deviates of the latest first-stage proposal (MCMC_propose's u): what a pooled RAM tick of the engine
     * reads back; test restatements only (tests/test_gpu_pooled.py) */
    double *last_u;
    /* bookkeeping for tests that replace the factors of an adaptation from outside (tests/test_oracle_fuzz_reference.py):
     * calls of MCMC_calculate_R so far, and whether the latest covtor_svd floored its singular values (info = -1) */
    int n_calcR, svd_floored;
} mcxo_chain;

mcxo_chain *mcxo_chain_create(const mcxo_cfg *cfg, const mcxo_target *tgt, const double *par0,
                              const double *cmat0, double sigma2, int nobs,
                              uint32_t seed, uint32_t chain_id);
mcxo_chain *mcxo_chain_create_ny(const mcxo_cfg *cfg, const mcxo_target *tgt, const double *par0,
                                 const double *cmat0, const double *sigma2, const int *nobs, int ny,
                                 uint32_t seed, uint32_t chain_id);
void mcxo_chain_free(mcxo_chain *c);
/* MCMC_init tail + first row of MCMC_run*, then iterations 2..nsimu (or up to upto) */
int mcxo_chain_run(mcxo_chain *c, int upto);

/* MCMC_run1 / MCMC_run1_er, the arithmetic of one invocation (state machine: oracle/run1.py) */
int mcxo_run1_decide(mcxo_chain *c, int drstage, const double *oldpar2, const double *ssprev2, double sspri2,
                     const double *oldpar1, const double *ssprev1, double sspri1, double alpha12,
                     const double *newpar, const double *ss, double sspri, double *alpha_out);
void mcxo_run1_propose(mcxo_chain *c, int stage, const double *from, double *newpar);
double mcxo_run1_sscrit(mcxo_chain *c, const double *ssprev1, double sspri1);

/* numerics exposed for known-answer tests */
double mcxo_normal(mcxo_rng *g);
double mcxo_gamma(mcxo_rng *g, double a, double b);
void mcxo_trmv_ut(int n, const double *R, double *x);          /* x <- R'x, R upper col-major; dot products ascending */
void mcxo_trmv_ut_desc(int n, const double *R, double *x);     /* the same with netlib's own order: diagonal first, rows descending */
int  mcxo_potrf_u(int n, double *A);                            /* LAPACK dpotf2 'U' */
int  mcxo_potri_u(int n, double *A);                            /* dtrti2 + dlauu2 'U' */
void mcxo_rotg(double *da, double *db, double *c, double *s);
double mcxo_nrm2(int n, const double *x);
void mcxo_chud(int p, double *R, const double *x, double *c, double *s);
int  mcxo_chdd(int p, double *R, const double *x, double *c, double *s);
void mcxo_covmat(int n, int p, const double *x, int ldx, const double *w, int nw,
                 double *cmat, double *xmean, double *wsum, int update);
int  mcxo_calculate_R(mcxo_chain *c, double *cmat);
int  mcxo_symsvd(int n, double *G, double *V, double *s);      /* pinned dgesvd('A','N') of a PSD matrix, mcx_svd.h */
void mcxo_gemv(int trans, int n, const double *A, const double *x, double *y);   /* y = A x or A'x, netlib dgemv order */
double mcxo_log(double x);
double mcxo_exp(double x);
double mcxo_alpha(double ss1, double pri1, double ss2, double pri2, double sigma2);

#ifdef __cplusplus
}
#endif
#endif
