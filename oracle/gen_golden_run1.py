"""oracle/gen_golden_run1.py -- TEST INFRASTRUCTURE. Not part of the product path.

Generates tests/golden/run1/*.npz: the REAL reference's mcmc_main_one (oracle/_ref/mcxref_one = MCMC_run1 / MCMC_run1_er
compiled from /root/reference) driven through K consecutive invocations on the pinned Philox stream (invocation k keyed
(seed0 + k, 0)).  Dev container only; the fixtures are data: the inputs, and per invocation what the program left in
mcmcrun.nml and the mcmc*.dat files.

    python oracle/gen_golden_run1.py

Keys: cfg_* / prob_* as in the chain fixtures; seed0, K; nml int32 [K][4] = drstage, isimu, ieval, nrej; alpha12 [K];
sscrit [K]; accepted [K]; parnew [K][npar]; parf [K][npar]; chainrow [K][npar+1]; ssprev1 [K][nycol]; valid = number of
leading invocations the reference's behaviour is defined for (see case nodr).
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po, refrun as rr  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "run1")
XDATA = np.arange(11.0)
YDATA = np.array([9.33, 9.40, 8.99, 7.06, 7.13, 6.69, 4.69, 4.24, 4.77, 3.86, 4.02])   # testcases/data.dat:2-12


def cases():
    rng = np.random.default_rng(7)
    d = 5
    A = rng.standard_normal((d, d)); lam = A @ A.T + d * np.eye(d)
    g = dict(kind="gauss", npar=d, par0=0.1 * np.arange(1, d + 1), cmat0=0.04 * np.eye(d) + 0.01, mu=np.zeros(d), lam=lam,
             lo=np.array([-0.35] + [-np.inf] * (d - 1)))
    e = dict(kind="expdata", npar=2, par0=[10, 0.1], cmat0=[[0.2, 0], [0, 0.001]], sigma2=1.0, nobs=11, xdata=XDATA, ydata=YDATA, lo=[0, 0])
    c = {}
    # delayed rejection with bounds: out-of-bounds first tries jump to stage 2 (MCMC_run1.F90:192-198)
    c["dram_dr_gauss5"] = (dict(nsimu=40, drscale=2.0, updatesigma=0), g, 40, 1000)
    c["dram_dr_expdata"] = (dict(nsimu=30, drscale=3.0, updatesigma=0), e, 30, 2000)
    # early rejection: the threshold of the next point drawn with its proposal (MCMC_run1_er.F90:162-190)
    c["er_gauss5"] = (dict(nsimu=40, method="er", updatesigma=0), g, 40, 3000)
    # ... with a prior on theta(1) (sigma <= 0: flat, priorfun.f90:78): rejection by the prior alone
    c["er_gauss5_prior"] = (dict(nsimu=40, method="er", updatesigma=0),
                            dict(g, pri_mu=np.array([0.6] + [0.0] * (d - 1)), pri_sig=np.array([0.25] + [0.0] * (d - 1))), 40, 4000)
    # no delayed rejection: the reference proposes from an unset local after its first rejection (MCMC_run1.F90:44-45,
    # 185-189), so only the invocations before that rejection are pinned (`valid`)
    c["dram_nodr_gauss5"] = (dict(nsimu=40, drscale=0.0, updatesigma=0), g, 12, 5000)
    # the SVD factor: MCMC_propose = matmulx(R, z) (MCMC_DRAM.F90:27), iC = dpotri on the factor's upper triangle; dgesvd
    # answered by the pinned Jacobi routine (oracle/_ref/mcxref_one_svd)
    c["dram_dr_svd_gauss5"] = (dict(nsimu=30, drscale=2.0, updatesigma=0, condmax=1e8), dict(g, cmat0=0.02 * (A @ A.T / d + np.eye(d))), 30, 6000)
    # two response columns (mcmcnycol.dat, one sigma2 per column): the sums over columns in MCMC_alpha / MCMC_DR_alpha13
    y2 = np.vstack([YDATA, 0.8 * YDATA + 0.3])
    c["dram_dr_expcols2"] = (dict(nsimu=30, drscale=2.0, updatesigma=0),
                             dict(kind="expdata", npar=3, par0=[10, 0.1, 0.1], cmat0=np.diag([0.2, 0.001, 0.001]), sigma2=[0.5, 0.8], nobs=[11, 11],
                                  xdata=XDATA, ydata=y2, lo=[0, 0, 0]), 30, 7000)
    return c


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, (ckw, pkw, K, seed0) in cases().items():
        cfg = po.make_cfg(**ckw)
        prob = po.Problem(**pkw)
        seeds = [seed0 + k for k in range(K)]
        ref = rr.run_program_one(rr.EXE_ONE_SVD if cfg.usesvd else rr.EXE_ONE, cfg, prob, seeds)
        valid = K
        if not cfg.dodr and cfg.method != po.METHODS["er"]:
            rej = [k for k in range(1, K) if not ref[k]["accepted"]]
            valid = rej[0] if rej else K
        out = {"cfg_" + k: getattr(cfg, k) for k, _ in po.Cfg._fields_}
        for k, v in pkw.items():
            out["prob_" + k] = np.asarray(v) if not isinstance(v, str) else v
        out.update(seed0=seed0, K=K, valid=valid,
                   nml=np.array([[f["drstage"], f["isimu"], f["ieval"], f["nrej"]] for f in ref], dtype=np.int32),
                   alpha12=np.array([f["alpha12"] for f in ref]), sscrit=np.array([f["sscrit"] for f in ref]),
                   accepted=np.array([f["accepted"] for f in ref], dtype=np.uint8),
                   parnew=np.array([f["parnew"] for f in ref]), parf=np.array([f["parf"] for f in ref]),
                   chainrow=np.array([f["chainrow"] for f in ref]), ssprev1=np.array([f["ssprev1"] for f in ref]))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "K", K, "valid", valid, "accepted", int(out["accepted"].sum()), "max drstage", int(out["nml"][:, 0].max()),
              "max nrej", int(out["nml"][:, 3].max()))


if __name__ == "__main__":
    main()
