"""oracle/gen_golden.py -- TEST INFRASTRUCTURE. Not part of the product path.

Generates tests/golden/*.npz by running the REAL mcmcf90 reference (oracle/_ref/mcxref,
compiled from /root/reference by oracle/Makefile; flang -O2 + MKL) on the pinned Philox
stream.  Runs only in the dev container (the reference does not travel); the committed
fixtures are data only: the inputs of each run and what MCMC_writechains
(MCMC_aux.F90:17-85) wrote for it.

    python oracle/gen_golden.py          # rewrites tests/golden/

Each fixture holds: cfg_* (namelist values), problem arrays, and from the reference
    runlen    int32  the repeat-count column of chain.mat  (= the accept-index sequence)
    rows_head / rows_tail   first / last 16 accepted rows (theta)
    ss_head / ss_tail       matching sschain values
    s2_head / s2_tail       s2chain (if updatesigma)
    chaincmat, chainmean    mcmccovf.dat / mcmcmean.dat
    rng_n     number of uniforms the reference drew
    svd_s, svd_U, svd_ticks, rows_at_ticks   (c5 only) what MKL's dgesvd returned at MCMC_init and at each adaptation
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po, refrun as rr  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

XDATA = np.arange(11.0)
YDATA = np.array([9.33, 9.40, 8.99, 7.06, 7.13, 6.69, 4.69, 4.24, 4.77, 3.86, 4.02])   # testcases/data.dat:2-12


def corr_gauss(d, rho=0.5):
    S = rho ** np.abs(np.subtract.outer(np.arange(d), np.arange(d)))
    return np.linalg.inv(S)


def cases():
    """name -> (cfg kwargs, Problem kwargs, chain_id).  Mirrors BASELINE.json configs 1-4 at oracle-sized nsimu."""
    c = {}
    c["c1_expdata_dram"] = (dict(nsimu=10000, adaptint=100, burnintime=1000, doburnin=1, greedy=1, drscale=2.0,
                                 updatesigma=1, N0=1.0, S02=0.0, scalelimit=0.3),
                            dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5,
                                 nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0]), 0)
    c["c1_shipped_nml"] = (dict(nsimu=1000, adaptint=200, burnintime=1000, doburnin=1, drscale=0.0,
                                updatesigma=1, N0=1.0, S02=0.0),
                           dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5,
                                nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0]), 0)
    c["c1_priors_ap"] = (dict(nsimu=5000, adaptint=100, adapthist=300, updatesigma=1, drscale=3.0),
                         dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5,
                              nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0], pri_mu=[9.0, 0.1], pri_sig=[2.0, 0.0]), 2)
    d = 10
    c["c2_gauss10_am"] = (dict(nsimu=3000, adaptint=100, updatesigma=0),
                          dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d),
                               lam=np.eye(d)), 0)
    c["c2_gauss10_am_initcmatn"] = (dict(nsimu=3000, adaptint=100, updatesigma=0, initcmatn=50, adaptend=2000),
                                    dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d),
                                         mu=np.zeros(d), lam=np.eye(d)), 5)
    d = 20
    c["c3_banana20_dram"] = (dict(nsimu=3000, adaptint=100, updatesigma=0, drscale=2.0),
                             dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), b=0.1), 3)
    d = 50
    c["c4_gauss50_ram"] = (dict(nsimu=3000, method="ram", updatesigma=0),
                           dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d),
                                lam=corr_gauss(d)), 7)
    c["c4_gauss50_am"] = (dict(nsimu=2000, adaptint=100, updatesigma=0),
                          dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d),
                               lam=corr_gauss(d)), 11)
    # --- edge cases
    c["e1_expdata_ram_bounds_s2"] = (dict(nsimu=3000, method="ram", updatesigma=1, N0=1.0, S02=0.0),       # RAM + out-of-bounds
                                     dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[4.0, 0], [0, 0.05]],   # proposals: stale alpha12
                                          sigma2=0.5, nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0]), 9)         # (MCMC_run_ram.F90:52-54)
    c["e2_gauss1_am"] = (dict(nsimu=1500, adaptint=50, updatesigma=0),                                      # npar = 1
                         dict(kind="gauss", npar=1, par0=[3.0], cmat0=[[0.5]], mu=[1.0], lam=[[2.0]]), 13)
    c["e3_gauss3_ram_burnin"] = (dict(nsimu=1200, method="ram", updatesigma=0, doburnin=1, burnintime=300,    # RAM idle during burn-in
                                      alphatarget=0.4, nuparam=0.6),
                                 dict(kind="gauss", npar=3, par0=[1.0, -1.0, 0.5], cmat0=np.diag([2.0, 1.0, 3.0]),
                                      mu=[0.0, 0.0, 0.0], lam=[[2.0, 0.3, 0.0], [0.3, 1.0, -0.2], [0.0, -0.2, 0.5]]), 17)
    c["e4_banana9_dram_noadapt"] = (dict(nsimu=800, doadapt=0, updatesigma=0, drscale=3.0),                 # DR without adaptation, odd npar
                                    dict(kind="banana", npar=9, par0=np.zeros(9), cmat0=0.5 * np.eye(9), b=0.03), 19)
    c["e5_gauss17_am_short"] = (dict(nsimu=99, adaptint=100, updatesigma=0),                                # never reaches an adaptation; odd npar
                                dict(kind="gauss", npar=17, par0=np.linspace(-1, 1, 17), cmat0=0.05 * np.eye(17),
                                     mu=np.zeros(17), lam=corr_gauss(17, 0.3)), 23)
    c["e6_expdata_er_priors"] = (dict(nsimu=3000, method="er", adaptint=100, updatesigma=1),               # early rejection
                                 dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5,
                                      nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0], pri_mu=[9.0, 0.1], pri_sig=[0.5, 0.02]), 41)
    c["e7_gauss10_er"] = (dict(nsimu=2000, method="er", adaptint=100, updatesigma=0),
                          dict(kind="gauss", npar=10, par0=np.zeros(10), cmat0=0.01 * np.eye(10), mu=np.zeros(10),
                               lam=np.eye(10)), 42)
    # --- SVD paths: the reference is linked with the pinned Jacobi dgesvd (oracle/ref/dgesvd_shim.c), see mcx_svd.h
    d = 6
    Sg = 0.6 ** np.abs(np.subtract.outer(np.arange(d), np.arange(d))) * np.outer(np.logspace(0, 1.5, d), np.logspace(0, 1.5, d))
    g6 = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.05 * np.eye(d), mu=np.linspace(-1, 1, d), lam=np.linalg.inv(Sg))
    c["s1_gauss6_scam"] = (dict(nsimu=1500, method="scam", adaptint=100, updatesigma=0), g6, 31)
    c["s2_gauss6_dram_svd"] = (dict(nsimu=2000, adaptint=100, updatesigma=0, condmax=1e10), g6, 32)
    c["s3_gauss6_dram_svd_dr"] = (dict(nsimu=2000, adaptint=100, updatesigma=0, condmax=1e10, drscale=2.0), g6, 33)
    c["s4_expdata_scam_s2"] = (dict(nsimu=2000, method="scam", adaptint=100, updatesigma=1),
                               dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5,
                                    nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0]), 34)
    c["s5_banana20_scam"] = (dict(nsimu=400, method="scam", adaptint=100, updatesigma=0),
                             dict(kind="banana", npar=20, par0=np.zeros(20), cmat0=0.01 * np.eye(20), b=0.1), 35)
    # --- nycol = 2: ssfunction returns one ss per response column, one sigma2 / nobs per column (MCMC_DRAM.F90:100-118,
    # 124-135, 162-186, 192-206; sschain has nycol+1 columns, s2chain nycol)
    Y2 = np.vstack([YDATA, 9.0 * np.exp(-0.25 * XDATA) + 0.2 * np.cos(2.0 * XDATA)])
    m = dict(kind="expdata", npar=3, par0=[9.0, 0.1, 0.2], cmat0=np.diag([0.02, 0.0001, 0.0002]), sigma2=[0.5, 0.3], nobs=[11, 13],
             xdata=XDATA, ydata=Y2, lo=[0, 0, 0])
    c["m1_expdata2_dram_dr_s2"] = (dict(nsimu=4000, adaptint=100, updatesigma=1, drscale=2.0, N0=1.0, S02=0.0), m, 51)
    c["m2_expdata2_er_s2"] = (dict(nsimu=3000, method="er", adaptint=100, updatesigma=1), m, 52)
    c["m3_expdata2_scam_s2"] = (dict(nsimu=1500, method="scam", adaptint=100, updatesigma=1), m, 53)
    c["m4_expdata2_ram"] = (dict(nsimu=2500, method="ram", updatesigma=0, N0=1.0, S02=0.0),
                            dict(m, cmat0=np.diag([0.5, 0.005, 0.01]), sigma2=[0.5, 0.5]), 59)
    c["m5_expdata2_burnin_greedy_priors"] = (dict(nsimu=3000, adaptint=100, updatesigma=1, doburnin=1, burnintime=500, greedy=1, scalelimit=0.3),
                                              dict(m, pri_mu=[9.0, 0.1, 0.2], pri_sig=[1.0, 0.0, 0.1]), 55)
    # --- BASELINE config 5 at its own dimension (SURVEY.md section 8c asks for "d=200 SCAM 50 its"), through TWO
    # adaptations (iterations 100 and 200).  Past an adaptation the d=200 trajectory is not a function of the inputs alone:
    # the covariance of <= 200 rows has rank < 200, the rotation in its null space is whatever the LAPACK linked makes of
    # rounding noise, and the reference linked to another BLAS would not reproduce itself either.  So this fixture comes
    # from the reference with MKL's own dgesvd, every call logged (oracle/_ref/mcxref_mkllog, ref/dgesvd_logger.c), and
    # holds what those calls returned (svd_s, svd_U): a checker that takes the logged factors at the same adaptations must
    # reproduce the reference's 250 iterations x 200 componentwise proposals decision by decision
    # (tests/test_oracle_golden.py, tests/test_gpu_parity.py), which pins MCMC_run_scam.F90:38-138 and the covariance
    # MCMC_adapt hands to scam_svd (MCMC_adapt.F90:138-157, matutils.F90:583-653) independently of any SVD routine.
    # --- NaN propagation (the corner round 3 left unpinned): a response value that is not a number makes every ss
    # NaN.  The reference (flang -O2) then never moves and draws no uniform besides the normals: MCMC_alpha's comparisons are false and
    # exp(NaN) = NaN, MCMC_reject(NaN) takes neither branch (MCMC_DRAM.F90:147-152), and in MCMC_DR_alpha13 min(1, NaN) = NaN (:178,185) --
    # had flang's min returned 1, every second stage would be accepted.  Pins the compare-select form of min1() / MCMC_alpha in the oracle.
    ynan = np.array(YDATA, dtype=float); ynan[4] = np.nan
    c["e8_nan_target_dr"] = (dict(nsimu=300, adaptint=100, drscale=2.0, updatesigma=0),
                             dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=0.5, nobs=11, xdata=XDATA, ydata=ynan, lo=[0, 0]), 61)
    # --- npar above 256 (round 5 lifted the engine's limit; the reference allocates any npar, MCMC_init.F90:81-102): AM through four adaptations
    # whose covariance has fewer rows than parameters (qcov_adjust keeps calculate_R's Cholesky alive), RAM with 45 150-element sweeps
    d = 260
    g260 = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=corr_gauss(d))
    c["e9_gauss260_am"] = (dict(nsimu=450, adaptint=100, updatesigma=0), g260, 71)
    c["e10_gauss260_ram"] = (dict(nsimu=400, method="ram", updatesigma=0), g260, 72)
    from mcmcf90_amd.workloads import problem
    ckw, pkw, _ = problem("c5", 250, adaptint=100)
    c["c5_illcond200_scam"] = (ckw, pkw, 51)
    return c


MKL_LOGGED = {"c5_illcond200_scam"}       # fixtures made with MKL's dgesvd + the call log instead of the pinned routine


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])                 # python oracle/gen_golden.py [name ...]: regenerate these fixtures only
    for name, (ckw, pkw, chain_id) in cases().items():
        if only and name not in only:
            continue
        cfg = po.make_cfg(**ckw)
        prob = po.Problem(**pkw)
        logged = name in MKL_LOGGED
        r = rr.run_reference(cfg, prob, chain_id=chain_id, pinned_svd=bool(cfg.usesvd) and not logged, svd_log=logged)
        k = 16
        out = {"chain_id": chain_id, "rng_n": r.rng_n, "chainind": r.chainind, "pinned_svd": int(bool(cfg.usesvd) and not logged),
               "runlen": r.chain[:, -1].astype(np.int32),
               "rows_head": r.chain[:k, :-1], "rows_tail": r.chain[-k:, :-1],
               "ss_head": r.sschain[:k, 0], "ss_tail": r.sschain[-k:, 0],
               "chaincmat": r.chaincmat, "chainmean": r.chainmean}
        if logged:
            # what every dgesvd call returned: [0] MCMC_init's MCMC_calculate_R (cmat0 = 1e-6 I: U = I exactly, not stored),
            # [1..] the adaptations.  Also the rows the chain stood on at the adaptations (scam writes one row per iteration).
            assert all(info == 0 for info, _, _ in r.svd_calls)
            assert np.array_equal(r.svd_calls[0][2], np.eye(prob.npar))
            out["mkl_logged"] = 1
            out["svd_s"] = np.array([sv for _, sv, _ in r.svd_calls])
            out["svd_U"] = np.array([U for _, _, U in r.svd_calls[1:]])
            ticks = [it for it in range(cfg.adaptint, cfg.nsimu + 1, cfg.adaptint)]
            out["svd_ticks"] = np.array(ticks[:len(r.svd_calls) - 1], dtype=np.int32)
            assert r.chainind == cfg.nsimu                      # one row per iteration: row it-1 is the state after iteration it
            out["rows_at_ticks"] = r.chain[[t - 1 for t in out["svd_ticks"]], :-1]
        if cfg.updatesigma:
            out["s2_head"], out["s2_tail"] = r.s2chain[:k], r.s2chain[-k:]
        if r.sschain.shape[1] > 2:                       # nycol > 1: every ss column
            out["ss_head"], out["ss_tail"] = r.sschain[:k, :-1], r.sschain[-k:, :-1]
        for f, _ in po.Cfg._fields_:
            out["cfg_" + f] = getattr(cfg, f)
        for kk, v in pkw.items():
            out["prob_" + kk] = np.asarray(v)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("%-28s chainind=%d rng_n=%d" % (name, r.chainind, r.rng_n))


if __name__ == "__main__":
    main()
