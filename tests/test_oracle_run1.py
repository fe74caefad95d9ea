"""MCMC_run1 / MCMC_run1_er (mcmc_main_one: one evaluation per program invocation, state in files between invocations):
the restatement oracle/run1.py against the real reference -- the committed fixtures tests/golden/run1/*.npz, and, in
the dev container, oracle/_ref/mcxref_one run live on other seeds."""
import os
import numpy as np
import pytest
import run1_util


@pytest.mark.parametrize("name", run1_util.names())
def test_run1_restatement_equals_reference_fixture(name, oracle):
    from oracle import run1
    z, cfg, prob = run1_util.load(name, oracle)
    f = run1.new_files(prob.par0)
    for k in range(int(z["valid"])):
        f = run1.invoke(f, cfg, prob, int(z["seed0"]) + k)
        run1_util.check_invocation(z, k, f, 1e-9, name)
        f["par"] = f["parnew"].copy()            # the protocol's driver: mcmcparnew.dat becomes the next mcmcpar.dat
    assert f["done"] == (f["ieval"] >= cfg.nsimu)


def test_run1_nodr_after_a_rejection_is_undefined_in_the_reference(oracle):
    """Without delayed rejection the reference proposes from an unset local once a point has been rejected
    (MCMC_run1.F90:44-45,185-189): the fixture records where that starts; the restatement (and the engine's shim) go on
    from the last accepted point, so the two part there and not before."""
    z, cfg, prob = run1_util.load("dram_nodr_gauss5", oracle)
    assert 1 < int(z["valid"]) < int(z["K"]) and not bool(z["accepted"][int(z["valid"])])


@pytest.mark.parametrize("method,drscale", [("dram", 2.5), ("er", 0.0), ("dram", 1.5)])
def test_run1_restatement_equals_live_reference(method, drscale, oracle):
    from oracle import refrun, run1
    if not os.path.exists(refrun.EXE_ONE):
        pytest.skip("oracle/_ref/mcxref_one not built (dev container only)")
    rng = np.random.default_rng(int(drscale * 10) + len(method))
    d = 4
    A = rng.standard_normal((d, d)); lam = A @ A.T + d * np.eye(d)
    cfg = oracle.make_cfg(nsimu=25, drscale=drscale, updatesigma=0, method=method)
    prob = oracle.Problem(kind="gauss", npar=d, par0=rng.standard_normal(d) * 0.2, cmat0=0.05 * np.eye(d), mu=np.zeros(d), lam=lam,
                          hi=np.array([np.inf, 0.4, np.inf, np.inf]))
    seeds = [77000 + 13 * k for k in range(25)]
    ref = refrun.run_program_one(refrun.EXE_ONE, cfg, prob, seeds)
    f = run1.new_files(prob.par0)
    for k, s in enumerate(seeds):
        f = run1.invoke(f, cfg, prob, s)
        r = ref[k]
        assert [f[x] for x in ("drstage", "isimu", "ieval", "nrej", "accepted", "done")] == [r[x] for x in ("drstage", "isimu", "ieval", "nrej", "accepted", "done")], k
        np.testing.assert_allclose(f["parnew"], r["parnew"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(f["mean"], r["mean"], rtol=1e-12)
        f["par"] = f["parnew"].copy()
