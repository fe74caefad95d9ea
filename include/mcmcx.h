/*
 * include/mcmcx.h -- C ABI of the MI355X-native adaptive-Metropolis engine (libmcmcx.so).
 *
 * This is the drop-in boundary for mcmcf90's sampling hot path.  A Fortran host
 * (mcmcf90_amd/fortran/mcmcx_mod.F90: module mcmcmod + subroutine mcmc_main, same
 * names as mcmc.F90:12 / mcmc_main.F90:12) binds these entry points through
 * ISO_C_BINDING; tests and bench.py bind them through ctypes.  Plain pointers and
 * sizes only, all host memory unless a name says "dev".  Every function returns
 * 0 on success, <0 on a fatal error (the reference would `stop`: doerror,
 * matutils.F90:764-789), >0 for a warning; mcmcx_last_error() returns the text.
 *
 * What each entry point replaces in the reference:
 *
 *   mcmcx_create            MCMC_init_namelist + check_mcmcinit_parameters   mcmcinit.F90:184-230, 235-368
 *   mcmcx_set_par0          MCMC_setpar0_vec                                 MCMC_init.F90:168-183
 *   mcmcx_set_cmat0         MCMC_setcmat0_mat                                MCMC_init.F90:201-218
 *   mcmcx_set_sigma2nobs    MCMC_setsigma2nobs_vec                           MCMC_init.F90:295-323
 *   mcmcx_set_target_*      the user's ssfunction (device-resident form)     external_inc.h:12-19
 *   mcmcx_set_bounds        the user's checkbounds (box form)                external_inc.h:29-32
 *   mcmcx_set_priors        default priorfun + priorsfile                    priorfun.f90:31-103
 *   mcmcx_init              MCMC_init tail: first MCMC_calculate_R, counters MCMC_init.F90:81-160
 *   mcmcx_run               MCMC_run / MCMC_run_ram loop, MCMC_adapt,        MCMC_run.F90:41-107,
 *                           MCMC_adapt_ram, MCMC_savechain                   MCMC_run_ram.F90:45-81, MCMC_adapt.F90:12-174
 *   mcmcx_get_chain         chain / sschain / s2chain module arrays          mcmc.F90:31-33, MCMC_aux.F90:167-185
 *   mcmcx_get_chaincov      chaincmat / chainmean / chainwsum                mcmc.F90:38-40
 *   mcmcx_get_counters      stayed, bndstayed, draccepted, drtries           mcmc.F90:46-55
 *   mcmcx_run1_*            the arithmetic of one MCMC_run1 / MCMC_run1_er call  MCMC_run1.F90:131-199, MCMC_run1_er.F90:122-182
 *
 * N independent chains run at once; chain c draws from the Philox4x32-10 stream
 * keyed (seed, chain_id0 + c) and, in the default "replicas" mode, is the
 * reference's single chain to the last bit of its accept/reject sequence.
 */
#ifndef MCMCX_H
#define MCMCX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCMCX_METHOD_DRAM 0   /* method = 'dram' (AM / DRAM), mcmc_main.F90:35-36 */
#define MCMCX_METHOD_RAM  1   /* method = 'ram',              mcmc_main.F90:33-34 */
#define MCMCX_METHOD_SCAM 2   /* method = 'scam',             mcmc_main.F90:29-30 */
#define MCMCX_METHOD_ER   3   /* method = 'er',               mcmc_main.F90:31-32 */

#define MCMCX_DEFAULT_SEED 0x6D636D63u

/* Numeric members of namelist /mcmc/ (mcmcinit.F90:74-82) plus the multi-chain extensions. */
typedef struct mcmcx_config {
    int32_t npar;          /* d */
    int32_t nchains;       /* N independent chains on this device */
    int32_t method;
    int32_t nsimu;
    int32_t doadapt, doburnin, adaptint, adapthist, badaptint, adaptend, initcmatn;
    int32_t burnintime, greedy, updatesigma;
    double  scalelimit, scalefactor, drscale, N0, S02, condmax, alphatarget, nuparam;
    uint32_t seed;         /* Philox key word 0 */
    uint32_t chain_id0;    /* Philox key word 1 of chain 0 (rank offset when sharded) */
    int32_t record_accept; /* keep the wavefront accept ballots of every iteration */
    int32_t record_chain;  /* keep every accepted row (the reference's chain/sschain/s2chain) */
    int32_t device;        /* HIP device ordinal */
    int32_t pooled;        /* 1: one proposal factor shared by all chains, adapted from the pooled empirical
                            * covariance of the current states (multi-chain extension; 0 = the reference's per-chain AM) */
    int32_t scam_fast;     /* method = 'scam' only, opt-in (default 0 = the reference's operations): the componentwise proposal
                            * newpar = U (U'oldpar + delta e_j) (MCMC_run_scam.F90:106-115: two dgemv) is formed as
                            * newpar = oldpar + delta U(:,j) -- the same point up to the rounding of U U' = I, one column
                            * of the rotation instead of two passes over it.  Accept sequences and states agree with the
                            * reference-order form to rounding level, not bit for bit (tests/test_gpu_scam_fast.py). */
} mcmcx_config;

typedef struct mcmcx_engine *mcmcx_handle;

void mcmcx_config_defaults(mcmcx_config *cfg);                 /* mcmcinit.F90:184-230 */
int  mcmcx_create(const mcmcx_config *cfg, mcmcx_handle *out);
int  mcmcx_destroy(mcmcx_handle h);
const char *mcmcx_last_error(void);
const char *mcmcx_version(void);
int32_t mcmcx_device_count(void);                              /* HIP devices visible to this process (0: none -- nothing will run) */
int mcmcx_device_info(int32_t device, char *buf, int32_t len); /* "name arch, pci bus id, CUs, memory" of one of them (diagnostics) */
/* the device's UUID as 32 hex digits (len >= 33) and out5 = {engine clock limit kHz, memory clock limit kHz, memory bus width
 * in bits, compute units, L2 bytes}: what tells two boxes of a pool apart in a benchmark line (diagnostics) */
int mcmcx_device_ident(int32_t device, char *uuid_hex, int32_t len, int32_t *out5);
const char *mcmcx_last_kernel(mcmcx_handle h);                /* the sampling kernel the last mcmcx_run launched, as spelled in the
                                                                  source ("step_kernel<true, false, false>", "scam_pooled_kernel", ...);
                                                                  "" before the first run or with host callbacks (diagnostics) */

int mcmcx_set_par0(mcmcx_handle h, const double *par0, int32_t npar);
int mcmcx_set_cmat0(mcmcx_handle h, const double *cmat0_colmajor, int32_t npar);
int mcmcx_set_sigma2nobs(mcmcx_handle h, const double *sigma2, const int32_t *nobs, int32_t nycol);
int mcmcx_set_target_gauss(mcmcx_handle h, const double *mu, const double *lam_rowmajor);
int mcmcx_set_target_banana(mcmcx_handle h, double b);
int mcmcx_set_target_expdata(mcmcx_handle h, int32_t ndata, const double *x, const double *y);
/* The same model with nycol response columns (a vector-valued ssfunction, external_inc.h:4-9; one sigma2 per column,
 * MCMC_DRAM.F90:100-118, 192-206): ss_j = sum_i (y_j(i) - theta_1 exp(-theta_{1+j} x_i))**2, npar = 1 + nycol,
 * y = [nycol][ndata].  Device-resident; the iteration is cut at the evaluations like the host-callback path (several
 * short launches per iteration, no host round trip).  Give the nycol sigma2 / nobs with mcmcx_set_sigma2nobs. */
int mcmcx_set_target_expdata_cols(mcmcx_handle h, int32_t ndata, int32_t nycol, const double *x, const double *y);
/* Host-callback target: the user's own ssfunction / priorfun / checkbounds (external_inc.h:4-33) behind plain C
 * signatures (the Fortran shim adapts the array-result / assumed-shape ABIs).  The engine calls them from the
 * thread that calls mcmcx_run, one chain after the other, at the points the reference does (MCMC_run.F90:47,55-56,
 * 69,74-75); the candidates make a D2H/H2D round trip per stage, so this is the plumbing path, not the fast one.
 * priorfun / checkbounds may be NULL (flat prior, no bounds = the library defaults). */
typedef void    (*mcmcx_ssfun_t)(const double *theta, int32_t npar, int32_t ny, double *ss_out, void *user);
typedef double  (*mcmcx_priorfun_t)(const double *theta, int32_t npar, void *user);
typedef int32_t (*mcmcx_checkbounds_t)(const double *theta, int32_t npar, void *user);
int mcmcx_set_target_host(mcmcx_handle h, mcmcx_ssfun_t ss, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb, void *user);
/* Batched form of the same (opt-in; external_inc.h:12-19 has no counterpart): ss_batch evaluates n candidates in one
 * call -- theta[n][npar] row-major (= Fortran theta(npar,n)), ss[n][ny] -- and, with nthreads > 1, is called from that
 * many threads at once on disjoint slices, so it must be re-entrant (the very first evaluation, at mcmcx_init, is made
 * from the calling thread alone, so load-on-first-call code is safe).  Bounds and prior stay per chain on the calling
 * thread, before the batch.  (method = 'er' with mcmcx_set_target_host_er keeps the per-chain path.) */
typedef void (*mcmcx_ssfun_batch_t)(const double *theta, int32_t npar, int32_t n, int32_t ny, double *ss_out, void *user);
int mcmcx_set_target_host_batch(mcmcx_handle h, mcmcx_ssfun_batch_t ss_batch, mcmcx_priorfun_t pri, mcmcx_checkbounds_t cb,
                                void *user, int32_t nthreads);
/* The user's ssfunction / priorfun / checkbounds as DEVICE code: a code object (hipcc --genco) whose kernel
 * `kernel_name` was defined with MCMCX_DEFINE_TARGET (include/mcmcx_target.h).  The engine launches it where the
 * reference calls the functions; candidates and results never leave HBM.  userdata (nbytes, may be NULL) is copied
 * to the device and handed to the functions. */
int mcmcx_set_target_module(mcmcx_handle h, const char *code_object_path, const char *kernel_name, const void *userdata, int64_t nbytes);
/* method='er' with host callbacks: the user's ssfunction_er(theta,npar,ny,sscrit) (external_inc.h:16-20), which may
 * stop summing once it passes sscrit; NULL (default) = ssfunction, like ssfunction_er0.f90.  Same `user` pointer. */
typedef void (*mcmcx_ssfun_er_t)(const double *theta, int32_t npar, int32_t ny, double sscrit, double *ss_out, void *user);
int mcmcx_set_target_host_er(mcmcx_handle h, mcmcx_ssfun_er_t ss_er);
int mcmcx_set_bounds(mcmcx_handle h, const double *lo, const double *hi);       /* NULL = unbounded side */
int mcmcx_set_priors(mcmcx_handle h, const double *mu, const double *sig);      /* sig <= 0: flat */
int mcmcx_set_stream(mcmcx_handle h, void *hip_stream);

int mcmcx_init(mcmcx_handle h);
/* iterations simuind+1 .. upto.  Returns MCMCX_INTERRUPTED (> 0) when a signal caught by
 * mcmcx_install_signal_handlers arrived: the run stopped at a launch boundary, mcmcx_simuind() says where, and
 * every getter is valid for the iterations done (the reference saves the chain "upto simuind" and stops,
 * MCMC_signal_handler.F90:95-107).  In pooled mode with a communicator of several ranks the run is left only at an
 * adaptation tick, by agreement of all ranks (every rank's stop flag travels with the pooled vector), so that no rank
 * is left waiting in a collective: that tick's adaptation IS applied before the ranks return, so mcmcx_clear_interrupt +
 * mcmcx_run resumes into the trajectory of an uninterrupted run.  Where no tick lies ahead of the call's `upto`
 * (doadapt = 0, past adaptend, the tail of a run) nothing collective remains and the rank that caught the signal returns
 * at its next launch boundary by itself.  A rank that fails marks the communicator and its peers' waits give up (< 0).
 * A run that failed inside a host-callback iteration (a HIP error around the user's ssfunction / priorfun / checkbounds) leaves
 * the handle unusable: with the phases fused, the next iteration's proposal -- its random draws included -- has already run, and a
 * second mcmcx_run would draw it again and leave the reference's stream order.  Later calls return -42; destroy the handle.
 * (mcmcx_create refuses npar > 4096 with -5: the packed triangle's int-typed indices, 64 npar (npar + 1) / 2 < 2**31.) */
int mcmcx_run(mcmcx_handle h, int32_t upto);
int mcmcx_sync(mcmcx_handle h);

/* ---- MCMC_run1 / MCMC_run1_er (mcmc_main_one, mcmc_main.F90:49-70): one evaluation of ssfunction per program
 * invocation, the chain's state carried in files between invocations (MCMC_run1.F90:14-29).  The file protocol is the
 * host's (the Fortran shim's mcmc_main_one); what is arithmetic in one invocation runs here, on the factors of
 * mcmcx_init (R, R2 = R/drscale, iC) and the chain's stream.  mcmcx_set_target_external declares that EVERY
 * evaluation of ssfunction / priorfun / checkbounds is the caller's -- mcmcx_init evaluates nothing, mcmcx_run refuses.
 * All chains at once; vectors are row-major per chain ([nchains][npar], [nchains][nycol], [nchains]).
 *   mcmcx_run1_decide   alpha = MCMC_alpha(oldpar1 -> newpar) in stage 1 (MCMC_run1.F90:141), or
 *                       MCMC_DR_alpha13(oldpar2, oldpar1, newpar) in stage 2 with drscale > 0 (:137-139; oldpar2 =
 *                       the current point, oldpar1 = the rejected first try); then reject = MCMC_reject(alpha) (:143).
 *                       The stage-2 arguments may be NULL in stage 1.
 *   mcmcx_run1_propose  newpar = MCMC_propose(from, R) (stage 1) or (from, R2) (stage 2 with drscale > 0) (:185-189)
 *   mcmcx_run1_sscrit   sscrit = MCMC_sscrit(ssprev1, sspri1) = -2 log u + sum(ss/sigma2) + pri  (MCMC_run1_er.F90:168) */
int mcmcx_set_target_external(mcmcx_handle h);
int mcmcx_run1_decide(mcmcx_handle h, int32_t drstage, const double *oldpar2, const double *ssprev2, const double *sspri2,
                      const double *oldpar1, const double *ssprev1, const double *sspri1, const double *alpha12,
                      const double *newpar, const double *ss, const double *sspri, double *alpha_out, int32_t *reject_out);
int mcmcx_run1_propose(mcmcx_handle h, int32_t stage, const double *from, double *newpar_out);
int mcmcx_run1_sscrit(mcmcx_handle h, const double *ssprev1, const double *sspri1, double *sscrit_out);
#define MCMCX_INTERRUPTED 2
/* SIGHUP, SIGINT, SIGTERM, SIGTSTP, SIGUSR1, SIGUSR2 (the set of signalqq.c:57-62) raise a flag that mcmcx_run
 * polls between kernel launches; mcmcx_interrupted() reads it, mcmcx_clear_interrupt() resets it */
int mcmcx_install_signal_handlers(void);
int mcmcx_interrupted(void);
void mcmcx_clear_interrupt(void);

int32_t mcmcx_simuind(mcmcx_handle h);
/* counters[0..7] = stayed, bndstayed, draccepted, drtries, chainind, status bits, erstayed, run length of the
 * current row, of one chain */
int mcmcx_get_counters(mcmcx_handle h, int32_t chain, int32_t *counters8);
/* over all chains: sums of stayed, bndstayed, draccepted, drtries, proposals (stage 1 + stage 2), RAM iterations that
 * were Cholesky downdates (a < 0, MCMC_run_ram.F90:168-172); [6] = the OR of every chain's status bits
 * (MCMCX_ST_*: where the reference would have stopped -- matutils.F90:719-722, MCMC_adapt.F90:221 -- or kept its old
 * factor, MCMC_adapt.F90:168-171) */
#define MCMCX_ST_RAM_DOWNDATE_FAIL 1
#define MCMCX_ST_CHOL_FAIL 2
#define MCMCX_ST_POTRI_FAIL 4
int mcmcx_get_totals(mcmcx_handle h, int64_t *totals7);      /* writes SEVEN values (five before library version 0.2) */
int mcmcx_get_totals_n(mcmcx_handle h, int64_t *totals, int32_t n);   /* the first n of them: the caller states its buffer */
int mcmcx_get_theta(mcmcx_handle h, double *theta_rowmajor /* [nchains][npar] */);
/* per chain: ss1, sspri1, sigma2, alpha12 */
int mcmcx_get_scalars(mcmcx_handle h, double *out /* [nchains][4] */);
/* per chain: uniforms drawn, polar cache flag, cached deviate */
int mcmcx_get_rng(mcmcx_handle h, int32_t chain, uint64_t *n, int32_t *saved, double *saved_y);
int mcmcx_get_R(mcmcx_handle h, int32_t chain, double *R_colmajor);    /* upper triangular, or the full SVD factor when condmax > 0 */
int mcmcx_get_qcovstd(mcmcx_handle h, int32_t chain, double *std);      /* SCAM: qcovstd (mcmc.F90:37) */
int mcmcx_get_chaincov(mcmcx_handle h, int32_t chain, double *cmat_colmajor, double *mean, double *wsum);
/* delayed-rejection state of one chain: R2 = R/drscale and the upper triangle of iC (mcmc.F90:36) */
int mcmcx_get_dr(mcmcx_handle h, int32_t chain, double *R2_colmajor, double *iC_colmajor);
/* accept flags of iterations 1..simuind for one chain (needs record_accept or record_chain) */
int mcmcx_get_accepted(mcmcx_handle h, int32_t chain, uint8_t *accepted);
/* raw wavefront ballots: masks[(it-1)*ntiles + tile], bit l = chain tile*64+l moved at iteration it */
int mcmcx_get_accept_masks(mcmcx_handle h, uint64_t *masks, int32_t *ntiles);
/* run-length compressed chain of one chain, like chain(1:chainind,:) / sschain / s2chain
 * (needs record_chain).  chain: [nrows][npar+1] row-major; ss: [nrows][nycol+1] (ss per column, repeat count);
 * s2: [simuind][nycol] */
int mcmcx_get_chain(mcmcx_handle h, int32_t chain, double *chain_out, double *ss_out, double *s2_out,
                    int32_t *nrows);
/* pooled moments of the current states of all chains of this device (shifted by par0):
 * out = [count, sum_j (th-par0)_j (npar), sum (th-par0)_j (th-par0)_k j<=k (npar(npar+1)/2)] */
int mcmcx_pooled_moments(mcmcx_handle h, double *out);
int32_t mcmcx_pooled_moments_len(mcmcx_handle h);
/* same, written to a device buffer of the caller (the operand of an RCCL all-reduce); asynchronous on
 * the engine's stream */
int mcmcx_pooled_moments_dev(mcmcx_handle h, void *dev_out);

/* Pooled mode across several GPUs: at every adaptation tick the engine writes its local moment vector
 * (mcmcx_pooled_moments_len doubles) to dev_buf, synchronises its stream and calls fn(user); fn must sum dev_buf
 * over all ranks in place (an RCCL all-reduce) and return after the result is visible.  Without a hook the local
 * moments are used (single GPU). */
typedef void (*mcmcx_exchange_t)(void *user);
int mcmcx_set_exchange(mcmcx_handle h, mcmcx_exchange_t fn, void *user, void *dev_buf);
/* pooled proposal state: chaincmat, chainmean, chainwsum and the shared factor R (column-major d x d) */
int mcmcx_get_pooled(mcmcx_handle h, double *cmat_colmajor, double *mean, double *wsum, double *R_colmajor);

/* ---- several GPUs of one node (no counterpart in the single-chain reference).  Chains are
 * sharded by rank (cfg.chain_id0 = rank * nchains keys the streams, so results do not depend on the GPU count); the
 * one exchange is the pooled moment vector, combined over RCCL (all-gather + a fixed pairwise tree over the ranks:
 * bit-identical on 1, 2, 4, 8 GPUs).  Attach a communicator before mcmcx_init; in pooled mode every adaptation tick
 * then uses the moments of ALL ranks.
 *   one process per GPU:   mcmcx_comm_create(key, rank, nranks, device, MCMCX_COMM_RCCL, &c) on every rank with the
 *                          same `key` (names a POSIX shm segment that carries the ncclUniqueId: one node, no MPI);
 *   one process, N GPUs:   mcmcx_comm_create_all(N, NULL, comms) (ncclCommInitAll), one engine per comms[i],
 *                          mcmcx_run_all / mcmcx_allreduce_moments_all drive them together.
 * MCMCX_COMM_HOST stages the gather through the shm segment instead of RCCL: for ranks that share one GPU. */
typedef struct mcmcx_comm *mcmcx_comm_t;
#define MCMCX_COMM_RCCL 0
#define MCMCX_COMM_HOST 1
int mcmcx_comm_create(const char *key, int32_t rank, int32_t nranks, int32_t device, int32_t backend, mcmcx_comm_t *out);
int mcmcx_comm_create_all(int32_t ndev, const int32_t *devices /* NULL: 0..ndev-1 */, mcmcx_comm_t *out /* [ndev] */);
int mcmcx_comm_destroy(mcmcx_comm_t c);
int32_t mcmcx_comm_rank(mcmcx_comm_t c);
int32_t mcmcx_comm_size(mcmcx_comm_t c);
int mcmcx_comm_barrier(mcmcx_comm_t c);
/* sum (op 0) / maximum (op 1) of n <= 512 host doubles over the ranks, in place (ncclAllReduce) */
int mcmcx_comm_allreduce_host(mcmcx_comm_t c, double *inout, int32_t n, int32_t op);
int mcmcx_set_comm(mcmcx_handle h, mcmcx_comm_t c);
/* pooled moments (layout of mcmcx_pooled_moments) of the chains of ALL ranks; collective: every rank calls it.
 * host_out may be NULL: the result stays on the device and the call is asynchronous on the engine's stream. */
int mcmcx_allreduce_moments(mcmcx_handle h, double *host_out);
int mcmcx_allreduce_moments_all(mcmcx_handle *hs, int32_t n, double *host_out);
/* mcmcx_run for the n engines of a one-process node, one host thread each */
int mcmcx_run_all(mcmcx_handle *hs, int32_t n, int32_t upto);

/* device time of the step kernel over all launches since the last reset, measured with HIP
 * events on the engine's stream; launches = number of step-kernel launches, steps = iterations */
int mcmcx_kernel_time(mcmcx_handle h, double *ms, int64_t *launches, int64_t *steps, int reset);

/* Test probes of the device primitives (no reference counterpart; used by tests/ only).
 * op: 0 log, 1 exp, 2 sqrt, 3 a/b, 4 fma(a,b,a), 5 drotg digest.  kind: 0 uniform, 1 normal, 2 gamma(a,b). */
int mcmcx_debug_math(int32_t op, int32_t n, const double *a, const double *b, double *out);
int mcmcx_debug_rng(uint32_t seed, uint32_t chain_id, int32_t kind, int32_t n, double a, double b, double *out,
                    uint64_t *nused);
/* The engine's kernel-selection tables (which sampling kernel a configuration runs: launch_step / launch_group / launch_scam of
 * mcx_api.hip), entry `index` as "family:name" into buf -- the name mcmcx_last_kernel reports after a run.  Returns the number
 * of entries (index = -1, buf = NULL: count only).  Needs no device.  tests/test_kernel_table.py requires a parity test per entry. */
int mcmcx_debug_kernel_table(int32_t index, char *buf, int32_t len);
/* every chain's SVD proposal factor (per-chain mode, condmax > 0) replaced by R (column-major d x d; scam: the rotation)
 * and qcovstd (scam; NULL = keep): a test feeds the factors an external dgesvd returned at an adaptation */
int mcmcx_debug_set_factor(mcmcx_handle h, const double *R_colmajor, const double *qcovstd);

#ifdef __cplusplus
}
#endif
#endif
