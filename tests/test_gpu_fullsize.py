"""BASELINE.json's configurations at their full chain counts, checked through size-independent
properties: (1) a few chains picked from the whole range are still bit-identical to the oracle run with
the same stream, (2) every chain's counters are consistent with its accept ballots, (3) pooled posterior
moments agree with the analytic target within Monte-Carlo error, (4) no chain raised a status flag."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _check_chains_vs_oracle(oracle, e, ckw, pkw, chains):
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in chains:
        o = oracle.run_chain(cfg, prob, chain_id=c)
        assert o.rc == 0
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["chainind"], cnt["draccepted"], cnt["drtries"]) == \
               (o.stayed, o.bndstayed, o.chainind, o.draccepted, o.drtries)
        assert e.rng(c)[0] == o.rng_n
        np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))


def test_c2_gauss10_am_65536_chains(oracle):
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd import dist as mdist
    d, n, nsimu = 10, 65536, 5000
    ckw = dict(nsimu=nsimu, adaptint=100, updatesigma=0)
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=np.eye(d))
    e = engine_from_problem(ckw, pkw, nchains=n, record_accept=1)
    e.init(); e.run()
    _check_chains_vs_oracle(oracle, e, ckw, pkw, [0, 1, 4097, 65535])
    tot = e.totals()
    masks = e.accept_masks()                                  # [nsimu][ntiles] wave ballots
    pop = int(np.unpackbits(masks.view(np.uint8)).sum())
    assert pop == n * nsimu - tot["stayed"]                   # accepted (incl. row 1) + stayed = nsimu per chain
    assert tot["proposals"] == n * (nsimu - 1)
    mean, cov = mdist.finalize_moments(e.pooled_moments(), d, np.zeros(d))
    assert np.max(np.abs(mean)) < 0.03                        # N(0, I): se = 1/sqrt(65536) = 0.004
    assert np.max(np.abs(cov - np.eye(d))) < 0.05
    acc = 1.0 - tot["stayed"] / (n * (nsimu - 1.0))
    assert 0.15 < acc < 0.45                                  # AM at d=10 settles near 0.25-0.3
    e.close()


def test_c3_banana20_dram_262144_chains(oracle):
    from mcmcf90_amd import engine_from_problem
    d, n, nsimu = 20, 262144, 600
    ckw = dict(nsimu=nsimu, adaptint=100, updatesigma=0, drscale=2.0)
    pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), b=0.1)
    e = engine_from_problem(ckw, pkw, nchains=n)
    e.init(); e.run()
    _check_chains_vs_oracle(oracle, e, ckw, pkw, [0, 131071, 262143])
    tot = e.totals()
    assert tot["proposals"] == n * (nsimu - 1) + tot["drtries"]
    assert 0 < tot["draccepted"] < tot["drtries"] <= tot["proposals"]
    assert tot["stayed"] + tot["draccepted"] <= n * (nsimu - 1)
    e.close()


def test_c4_gauss50_ram_131072_chains(oracle):
    from mcmcf90_amd import engine_from_problem
    d, n, nsimu = 50, 131072, 400
    S = 0.5 ** np.abs(np.subtract.outer(np.arange(d), np.arange(d)))
    ckw = dict(nsimu=nsimu, method="ram", updatesigma=0)
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=np.linalg.inv(S))
    e = engine_from_problem(ckw, pkw, nchains=n)
    e.init(); e.run(137); e.run()                             # a launch boundary in the middle of the run
    _check_chains_vs_oracle(oracle, e, ckw, pkw, [0, 77, 65536, 131071])
    for c in (0, 131071):
        assert e.counters(c)["status"] == 0                   # no failed downdate
    R = e.R(5)
    assert np.all(np.diag(R) > 0) and np.allclose(np.tril(R, -1), 0)
    e.close()


def test_c4_gauss50_ram_1048576_chains(oracle):
    """BASELINE config 4 as written -- ALL 1 048 576 chains on one GPU, what bench.py times at N = 1 (10.7 GB of per-chain factors, byte
    offsets past 2**32): 130 iterations with a launch boundary in the middle; chains from both ends and the middle of the range against
    oracle.run_chain (state, factor, counters, stream position), every chain's counters against the accept ballots, no status flag."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    n, nsimu = 1048576, 130
    ckw, pkw, _ = problem("c4", nsimu)
    e = engine_from_problem(ckw, pkw, nchains=n, record_accept=1)
    e.init(); e.run(67); e.run()
    assert e.last_kernel() == "step_kernel_ram_wide", e.last_kernel()
    _check_chains_vs_oracle(oracle, e, ckw, pkw, [0, 524288, 524289 + 63, 1048575])
    tot = e.totals()
    masks = e.accept_masks()                                  # [nsimu][ntiles] wave ballots
    pop = int(np.unpackbits(masks.view(np.uint8)).sum())
    assert pop == n * nsimu - tot["stayed"]                   # accepted (incl. row 1) + stayed = nsimu per chain
    assert tot["proposals"] == n * (nsimu - 1)
    assert tot["status"] == 0                                 # the OR of all chains' status bits: no failed downdate anywhere
    e.close()


def test_c4_gauss50_pooled_ram_1048576_chains(oracle, monkeypatch):
    """... and its pooled twin (bench.py's `c4_pooled`: one shared factor, the RAM statistic of all chains folded in every adaptint
    iterations): at 16384 tiles the engine takes pooled_mfma_ks_kernel (two waves per SIMD, the LDS vector in two pieces of forty rows: eight
    tiles per CU) by itself, and pooled_mfma_kernel<false, true> when that form is switched off.  Chains from both ends and
    the middle of the range are the single-chain oracle with the factor fixed (up to the tick at 100) and, carried on with the engine's
    pooled factor, to the end; the whole-vector and the one-wave-per-SIMD instances give the same 1 048 576 states, ballots and pooled factor
    bit for bit."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    n, nsimu, tick = 1048576, 130, 100
    ckw, pkw, _ = problem("c4", nsimu, adaptint=tick)
    monkeypatch.delenv("MCMCX_POOLED_WAVES", raising=False)
    monkeypatch.delenv("MCMCX_POOLED_KS", raising=False)
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
    e.init(); e.run(tick)
    assert e.last_kernel() == "pooled_mfma_ks_kernel", e.last_kernel()
    plain = oracle.make_cfg(**dict(ckw, doadapt=0, method="dram"))
    prob = oracle.Problem(**pkw)
    picks = [0, 63, 524288, 1048575]
    live = {c: oracle.LiveChain(plain, prob, chain_id=c) for c in picks}
    th = e.theta()
    for c in picks:
        live[c].run(tick)
        np.testing.assert_array_equal(_bits(th[c]), _bits(live[c].theta))
    R = e.pooled()[3]
    assert not np.array_equal(np.triu(R), 0.1 * 2.4 / np.sqrt(50.0) * np.eye(50))     # the tick did refactor
    e.run()
    th = e.theta(); masks = e.accept_masks(); tot = e.totals()
    for c in picks:
        live[c].set_R(R)
        live[c].run(nsimu)
        np.testing.assert_array_equal(_bits(th[c]), _bits(live[c].theta))
        assert e.rng(c)[0] == live[c].ch.contents.rng.n
        live[c].close()
    pop = int(np.unpackbits(masks.view(np.uint8)).sum())
    assert pop == n * nsimu - tot["stayed"] and tot["proposals"] == n * (nsimu - 1) and tot["status"] == 0
    e.close()
    for switch, value, kernel in (("MCMCX_POOLED_KS", "0", "pooled_mfma_kernel<false, true>"), ("MCMCX_POOLED_WAVES", "1", "pooled_mfma_kernel<false>")):
        monkeypatch.setenv(switch, value)
        e1 = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
        e1.init(); e1.run()
        assert e1.last_kernel() == kernel, e1.last_kernel()
        assert np.array_equal(_bits(e1.theta()), _bits(th)) and np.array_equal(e1.accept_masks(), masks)
        np.testing.assert_array_equal(_bits(e1.pooled()[3]), _bits(R))
        e1.close()


def test_c5_illcond200_scam_pooled_65536_chains(oracle):
    """BASELINE config 5 at full size in the pooled mode (one rotation for all chains, scam_pooled_kernel on the f64
    matrix cores).  (1) up to the first tick every chain is the single-chain oracle with adaptation off; (2) the shared
    rotation is an orthogonal eigenbasis of the pooled covariance of the states at the tick; (3) carried on with the
    engine's U / qcovstd, the oracle chains stay bit-identical to the engine to the end."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    n, nsimu, tick = 65536, 16, 10
    ckw, pkw, _ = problem("c5", nsimu, adaptint=tick)
    d = pkw["npar"]
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1)
    e.init(); e.run(tick)
    cfg = oracle.make_cfg(**dict(ckw, doadapt=0))
    prob = oracle.Problem(**pkw)
    picks = [0, 4097, 65535]
    live = {c: oracle.LiveChain(cfg, prob, chain_id=c) for c in picks}
    th = e.theta()
    for c in picks:
        live[c].run(tick)
        np.testing.assert_array_equal(_bits(th[c]), _bits(live[c].theta))
    cm, mean, W, U = e.pooled()
    std = e.qcovstd(0)
    assert W == n
    np.testing.assert_allclose(mean, th.mean(axis=0), rtol=0, atol=1e-12)
    np.testing.assert_allclose(cm, np.cov(th.T), rtol=1e-9, atol=1e-18)
    np.testing.assert_allclose(U.T @ U, np.eye(d), atol=1e-12)
    floor = std[0] ** 2 / cfg.condmax
    np.testing.assert_allclose((U * np.maximum(std ** 2, floor)) @ U.T, cm, rtol=0, atol=1e-10 * std[0] ** 2)
    assert np.all(np.diff(std) <= 0)
    # ... and bit for bit the oracle's pinned SVD (scam_svd, matutils.F90:583-653: dgesvd + the condmax floor) of that covariance
    import ctypes as C
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    G = np.asfortranarray(cm.copy()); V = np.zeros((d, d), order="F"); sv = np.zeros(d)
    oracle.lib().mcxo_symsvd(d, dp(G), dp(V), dp(sv))
    tol = sv[0] / cfg.condmax
    if sv[-1] <= tol:
        sv[sv < tol] = tol
    np.testing.assert_array_equal(_bits(U), _bits(V))
    np.testing.assert_array_equal(_bits(std), _bits(np.sqrt(sv)))
    e.run()
    th = e.theta()
    for c in picks:
        live[c].set_R(U); live[c].set_qcovstd(std)
        live[c].run(nsimu)
        np.testing.assert_array_equal(_bits(th[c]), _bits(live[c].theta))
        live[c].close()
    tot = e.totals()
    assert tot["proposals"] == n * (nsimu - 1) * d
    e.close()


@pytest.mark.slow
@pytest.mark.parametrize("lane_svd", [0, pytest.param(1, marks=pytest.mark.extended)], ids=["blocked_svd", "lane_svd"])   # (the lane form is production below npar 48 only)
def test_c5_illcond200_scam_replicas_two_ticks(oracle, lane_svd, monkeypatch):
    """BASELINE config 5's target with per-chain rotations (the reference's semantics) THROUGH two adaptations: 70 chains
    (a ragged second tile), adaptint = 10, 25 iterations = 5000 componentwise proposals per chain, the adaptation's pinned
    SVD in both device forms (a workgroup per chain through LDS -- thirteen column blocks and eight-lane partial chains of
    25 rows at d = 200 -- and one lane per chain) against oracle.run_chain: state, rotation, qcovstd, covariance, stream
    position, bit for bit."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    if lane_svd:
        monkeypatch.setenv("MCMCX_SVD_LANE", "1")
    ckw, pkw, _ = problem("c5", 25, adaptint=10)
    e = engine_from_problem(ckw, pkw, nchains=70, record_accept=1)
    e.init(); e.run()
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in (0, 63, 69):
        o = oracle.run_chain(cfg, prob, chain_id=c)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R))
        np.testing.assert_array_equal(_bits(e.qcovstd(c)), _bits(o.qcovstd))
        cm, mean, wsum = e.chaincov(c)
        np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)))
        np.testing.assert_array_equal(_bits(mean), _bits(o.chainmean))
        assert e.rng(c)[0] == o.rng_n
    assert e.totals()["proposals"] == 70 * 24 * 200
    e.close()
