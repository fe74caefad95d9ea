"""mcmc_main_one on the engine (MCMC_run1 / MCMC_run1_er: one evaluation per program invocation, state in files):
(1) the three arithmetic entry points of the C ABI, 130 chains at once, bit for bit against the oracle's per chain;
(2) the user program of the reference-side driver (oracle/ref/ref_main.F90, `call mcmc_main_one()`), linked against the
    engine's Fortran shim, run invocation by invocation like the protocol's driver script: every file it leaves equals the
    restatement bit for bit and the real reference's fixture (tests/golden/run1) to BLAS / libm rounding."""
import os
import numpy as np
import pytest
import run1_util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("drscale,ny", [(2.0, 1), (0.0, 1), (3.0, 2)])
def test_run1_entry_points_equal_the_oracle_per_chain(drscale, ny, oracle):
    from mcmcf90_amd import engine_from_problem
    from oracle import run1
    N = 130                                          # three tiles, the last one ragged
    rng = np.random.default_rng(5)
    if ny == 1:
        d = 7
        A = rng.standard_normal((d, d)); lam = A @ A.T + d * np.eye(d)
        pkw = dict(kind="gauss", npar=d, par0=0.1 * rng.standard_normal(d), cmat0=0.03 * np.eye(d) + 0.005, mu=np.zeros(d), lam=lam)
    else:
        d = 3
        x = np.arange(11.0)
        y = np.vstack([9.0 * np.exp(-0.1 * x), 9.0 * np.exp(-0.2 * x)]) + 0.1 * rng.standard_normal((2, 11))
        pkw = dict(kind="expdata", npar=d, par0=[9.0, 0.1, 0.2], cmat0=np.diag([0.2, 1e-3, 1e-3]), sigma2=[0.5, 0.7], nobs=[11, 11], xdata=x, ydata=y)
    ckw = dict(nsimu=10, drscale=drscale, updatesigma=0)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=N, external=True)
    e.init()
    with pytest.raises(Exception):
        e.run()                                      # an external target cannot be run by the engine
    ivs = []
    for c in range(N):
        iv = run1._Inv(cfg, prob, prob.par0, 0x6D636D63)
        iv.lc.close()
        iv.lc = oracle.LiveChain(cfg, iv.prob, seed=0x6D636D63, chain_id=c)      # chain c's stream, like the engine's lane c
        ivs.append(iv)
    cur = np.tile(prob.par0, (N, 1))
    ss_cur = np.array([iv.ss(cur[c]) for c, iv in enumerate(ivs)])
    # first try from the current point (stage 1), decided; then, where rejected and DR is on, a second try (stage 2), decided
    p1 = e.run1_propose(1, cur)
    o1 = np.array([iv.propose(1, cur[c]) for c, iv in enumerate(ivs)])
    assert np.array_equal(_bits(p1), _bits(o1))
    ss1 = np.array([iv.ss(p1[c]) for c, iv in enumerate(ivs)])
    pri = 0.01 * np.arange(N)
    a1, r1 = e.run1_decide(1, cur, ss_cur, np.zeros(N), p1, ss1, pri)
    oa1 = [iv.decide(1, cur[c], ss_cur[c], 0.0, cur[c], ss_cur[c], 0.0, 0.0, p1[c], ss1[c], pri[c]) for c, iv in enumerate(ivs)]
    assert np.array_equal(_bits(a1), _bits([a for a, _ in oa1])) and np.array_equal(r1, [r for _, r in oa1])
    assert 0 < r1.sum() < N
    p2 = e.run1_propose(2, cur)
    o2 = np.array([iv.propose(2, cur[c]) for c, iv in enumerate(ivs)])
    assert np.array_equal(_bits(p2), _bits(o2))
    if drscale > 0:
        assert not np.array_equal(p2 - cur, p1 - cur)
    ss2 = np.array([iv.ss(p2[c]) for c, iv in enumerate(ivs)])
    a12 = np.minimum(a1, 0.95)                       # alpha12 = 1 divides by zero in MCMC_DR_alpha13 (a NaN's sign is not pinned)
    a2, r2 = e.run1_decide(2, p1, ss1, pri, p2, ss2, 2 * pri, oldpar2=cur, ssprev2=ss_cur, sspri2=np.zeros(N), alpha12=a12)
    oa2 = [iv.decide(2, cur[c], ss_cur[c], 0.0, p1[c], ss1[c], pri[c], a12[c], p2[c], ss2[c], 2 * pri[c]) for c, iv in enumerate(ivs)]
    assert np.array_equal(_bits(a2), _bits([a for a, _ in oa2])) and np.array_equal(r2, [r for _, r in oa2])
    # the out-of-bounds first try of MCMC_run1.F90:192-198: ssprev1 = huge, alpha12 = 0
    a3, r3 = e.run1_decide(2, p1, np.full((N, ny), np.finfo(np.float64).max), pri, p2, ss2, pri, oldpar2=cur, ssprev2=ss_cur, sspri2=np.zeros(N), alpha12=np.zeros(N))
    oa3 = [iv.decide(2, cur[c], ss_cur[c], 0.0, p1[c], np.full(ny, np.finfo(np.float64).max), pri[c], 0.0, p2[c], ss2[c], pri[c]) for c, iv in enumerate(ivs)]
    assert np.array_equal(_bits(a3), _bits([a for a, _ in oa3])) and np.array_equal(r3, [r for _, r in oa3])
    sc = e.run1_sscrit(ss_cur, pri)
    osc = [iv.sscrit(ss_cur[c], pri[c]) for c, iv in enumerate(ivs)]
    assert np.array_equal(_bits(sc), _bits(osc))
    for c in (0, 64, 129):                           # the streams are where the oracle's are
        assert e.rng(c)[0] == ivs[c].lc.ch.contents.rng.n
    for iv in ivs:
        iv.close()
    e.close()


@pytest.mark.parametrize("name", run1_util.names())
def test_shim_mcmc_main_one_equals_restatement_and_reference(name, oracle):
    from oracle import refrun, run1
    exe = os.path.join(ROOT, "oracle", "_ref", "one_shim")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/one_shim not built (make -C oracle testcases)")
    z, cfg, prob = run1_util.load(name, oracle)
    K = int(z["K"])
    seeds = [int(z["seed0"]) + k for k in range(K)]
    got = refrun.run_program_one(exe, cfg, prob, seeds, seedfile=True)
    f = run1.new_files(prob.par0)
    for k in range(K):
        f = run1.invoke(f, cfg, prob, seeds[k])
        g = got[k]
        assert [g[x] for x in ("drstage", "isimu", "ieval", "nrej", "accepted", "done")] == [f[x] for x in ("drstage", "isimu", "ieval", "nrej", "accepted", "done")], (name, k)
        for key in ("parnew", "parf", "mean", "ssprev1", "oldpar1", "oldpar2", "ssprev2"):
            if f[key] is None or (key in ("oldpar1", "oldpar2", "ssprev2") and g[key] is None):
                continue
            assert np.array_equal(_bits(g[key]), _bits(f[key])), (name, k, key, g[key], f[key])
        assert g["alpha12"] == f["alpha12"] and g["sscrit"] == f["sscrit"], (name, k)
        if f["sscritfile"] is not None:
            assert g["sscritfile"] == f["sscritfile"]
        assert np.array_equal(_bits(g["chainrow"]), _bits(f["chainrow"]))
        if k < int(z["valid"]):
            run1_util.check_invocation(z, k, g, 1e-9, name + " (shim)")
        f["par"] = f["parnew"].copy()
