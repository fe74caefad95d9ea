"""The HIP engine (through the C ABI) against the oracle and against the fixtures of the real
Fortran reference.  Integer/index work (accept sequence, counters, stream position) must be
identical; by construction of the arithmetic the floating-point state is
identical to the oracle's too, so it is compared bit for bit, and to the reference fixture at
the BLAS/libm rounding level."""
import numpy as np
import pytest
from golden_util import names, load, accepted_from_runlen, GOLDEN

pytestmark = pytest.mark.gpu

# every fixture: AM, DRAM, RAM, burn-in scaling, greedy burn-in, AP window, priors, bounds, sigma2 update
# (the d=200 fixture has its own, truncated test below: 250 x 200 componentwise proposals of a handful of chains are
# minutes of latency-bound work for a one-wave launch)
# (the nycol = 2 fixtures m* need the host-callback target: tests/test_gpu_host_callbacks.py)
SUPPORTED = [n for n in names() if n != "c5_illcond200_scam" and not n.startswith("m")]


def _kw(z):
    ckw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_") and k[4:] not in ("dodr", "doscam", "usesvd")}
    pkw = {}
    for k in z.files:
        if k.startswith("prob_"):
            v = z[k]
            pkw[k[5:]] = v.item() if v.ndim == 0 else v
    return ckw, pkw


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


# the lane-per-chain kernel each fixture runs at 130 chains (mcx_api.hip's STEP_TABLE / SCAM_TABLE; tests/test_kernel_table.py wants every
# entry of those tables asserted by some parity test)
LANE_KERNEL = {
    "c1_expdata_dram": "step_kernel_dr", "c1_priors_ap": "step_kernel_dr", "c1_shipped_nml": "step_kernel_ldsr",
    "c2_gauss10_am": "step_kernel_ldsr", "c2_gauss10_am_initcmatn": "step_kernel_ldsr", "c3_banana20_dram": "step_kernel_dr",
    "c4_gauss50_am": "step_kernel_ldsv", "c4_gauss50_ram": "step_kernel_ram_wide", "e1_expdata_ram_bounds_s2": "step_kernel_ram_ldsr",
    "e2_gauss1_am": "step_kernel_ldsr", "e3_gauss3_ram_burnin": "step_kernel_ram_ldsr", "e4_banana9_dram_noadapt": "step_kernel_dr",
    "e5_gauss17_am_short": "step_kernel_ldsv", "e6_expdata_er_priors": "step_kernel_ldsr", "e7_gauss10_er": "step_kernel_ldsr",
    "e8_nan_target_dr": "step_kernel_dr", "s1_gauss6_scam": "scam_mw_kernel<8>", "s2_gauss6_dram_svd": "step_kernel_ldsv",
    "s3_gauss6_dram_svd_dr": "step_kernel_dr", "s4_expdata_scam_s2": "scam_mw_kernel<8>", "s5_banana20_scam": "scam_mw_kernel<8>",
    "e9_gauss260_am": "step_kernel<false, false, false>", "e10_gauss260_ram": "step_kernel_ram_wide",       # npar above 256 (round 5), from the real reference
}


@pytest.mark.parametrize("kernels", ["lane", "auto"])          # the lane-per-chain kernels / whatever the engine picks (tests/conftest.py)
@pytest.mark.parametrize("name", SUPPORTED)
def test_engine_matches_oracle_and_reference(oracle, name, kernels):
    from mcmcf90_amd import engine_from_problem
    z, cfg, prob = load(name, oracle)
    ckw, pkw = _kw(z)
    cid = int(z["chain_id"])
    nch = 130                                  # 3 tiles, last one ragged
    e = engine_from_problem(ckw, pkw, nchains=nch, chain_id0=cid - 1 if cid > 0 else 0, record_accept=1, record_chain=1)
    off = 1 if cid > 0 else 0                  # engine chain `off` has the fixture's stream
    e.init(); e.run()
    assert e.simuind == cfg.nsimu
    if kernels == "lane":
        assert e.last_kernel() == LANE_KERNEL[name], e.last_kernel()
    # --- against the real reference (fixture)
    acc = e.accepted(off)
    np.testing.assert_array_equal(acc, accepted_from_runlen(z["runlen"]))
    ch, ss, s2 = e.chain(off)
    np.testing.assert_array_equal(ch[:, -1].astype(np.int32), z["runlen"])
    assert e.rng(off)[0] == int(z["rng_n"])
    k = z["rows_head"].shape[0]
    scale = np.maximum(np.abs(z["rows_tail"]).max(axis=0), 1e-3)
    assert np.max(np.abs(ch[-z["rows_tail"].shape[0]:, :-1] - z["rows_tail"]) / scale) < 1e-7
    # posterior moments the reference wrote (mcmccovf.dat, mcmcmean.dat): north_star asks for 1e-6 relative
    cmf, meanf, _ = e.chaincov(off)
    assert np.max(np.abs(np.triu(cmf) - np.triu(z["chaincmat"]))) / max(np.max(np.abs(z["chaincmat"])), 1e-300) < 1e-6    # (fixture e8: covariance 0)
    assert np.max(np.abs(meanf - z["chainmean"])) / max(np.max(np.abs(z["chainmean"])), 1e-300) < 1e-6 \
        or np.max(np.abs(meanf - z["chainmean"])) < 1e-9 * np.sqrt(np.max(np.abs(z["chaincmat"])))
    # --- against the oracle, several chains incl. the ragged tile: bit for bit
    for c in (0, off, 63, 64, 129):
        o = oracle.run_chain(cfg, prob, chain_id=(cid - off) + c, continue_on_downdate_fail=True)
        # a failed RAM downdate (DCHDD INFO=-1) stops the reference; the engine flags the chain and goes on
        assert bool(e.counters(c)["status"] & 1) == (o.ram_downdate_fail != 0)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        chc, ssc, s2c = e.chain(c)
        np.testing.assert_array_equal(_bits(chc), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ssc), _bits(o.sschain))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2c), _bits(o.s2chain))
        n, saved, saved_y = e.rng(c)
        assert (n, saved) == (o.rng_n, o.rng_saved)
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["chainind"]) == (o.stayed, o.bndstayed, o.chainind)
        assert (cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == (o.draccepted, o.drtries, o.erstayed)
        if cfg.dodr:
            r2, ic = e.dr_state(c)
            if cfg.usesvd:
                np.testing.assert_array_equal(_bits(r2), _bits(o.R2))
            else:
                np.testing.assert_array_equal(_bits(np.triu(r2)), _bits(np.triu(o.R2)))
            np.testing.assert_array_equal(_bits(np.triu(ic)), _bits(np.triu(o.iC)))
        if cfg.usesvd:                                   # full factor U sqrt(s) 2.4/sqrt(d), or U for scam
            np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R))
            if cfg.doscam:
                np.testing.assert_array_equal(_bits(e.qcovstd(c)), _bits(o.qcovstd))
        else:
            np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))
        cm, mean, wsum = e.chaincov(c)
        np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)))
        np.testing.assert_array_equal(_bits(mean), _bits(o.chainmean))
        assert wsum == o.chainwsum
    th = e.theta(); sc = e.scalars()
    o = oracle.run_chain(cfg, prob, chain_id=(cid - off) + 129)
    np.testing.assert_array_equal(_bits(th[129]), _bits(o.theta))
    assert _bits(sc[129, 0]) == _bits(np.float64(o.ss1)) and sc[129, 2] == o.sigma2       # (bits: fixture e8's ss is NaN)
    e.close()


@pytest.mark.slow
def test_c5_fixture_all_iterations_with_the_logged_factors(oracle):
    """BASELINE config 5's fixture (d=200 SCAM, the MKL-linked reference through the adaptations at iterations 100 and
    200): ALL 250 iterations = 50000 componentwise proposals per chain on the device.  At each adaptation the engine runs
    its own tick (covariance update + the pinned SVD) and the test then replaces the factor by what MKL's dgesvd returned
    to the reference there (mcmcx_debug_set_factor) -- past an adaptation the rotation is not a function of the inputs
    alone (rank-deficient covariance), see oracle/gen_golden.py.  Chain 1 (the fixture's stream) against the reference:
    run-length column, stream position, the rows at the adaptations and at both ends; three chains incl. the ragged tile
    bit for bit against the oracle fed the same factors."""
    from mcmcf90_amd import engine_from_problem
    from golden_util import logged_factors
    z, cfg, prob = load("c5_illcond200_scam", oracle)
    ckw, pkw = _kw(z)
    cid = int(z["chain_id"])
    picks = (1, 65)                                       # the fixture's stream and the ragged tile (an oracle chain is ~25 s of this test)
    e = engine_from_problem(ckw, pkw, nchains=66, chain_id0=cid - 1, record_accept=1, record_chain=1)
    e.init()
    live = {c: oracle.LiveChain(cfg, prob, chain_id=cid - 1 + c) for c in picks}
    scale = np.maximum(np.abs(z["rows_tail"]).max(axis=0), 1e-3)
    for k, (it, U, sd) in enumerate(logged_factors(z, cfg)):
        if it > 0:
            e.run(it)
            th = e.theta()
            assert np.max(np.abs(th[1] - z["rows_at_ticks"][k - 1]) / scale) < 1e-7
            for c in picks:
                live[c].run(it)
                np.testing.assert_array_equal(_bits(th[c]), _bits(live[c].theta))
            # the tick's own products, before they are replaced: covariance, and the pinned SVD of it (blocked form at d=200)
            o = live[1].ch.contents
            cm, mean, wsum = e.chaincov(1)
            np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(np.ctypeslib.as_array(o.chaincmat, shape=(200, 200)).T)))
            np.testing.assert_array_equal(_bits(e.R(1)), _bits(np.ctypeslib.as_array(o.R, shape=(200, 200)).T))
            np.testing.assert_array_equal(_bits(e.qcovstd(1)), _bits(np.ctypeslib.as_array(o.qcovstd, shape=(200,))))
        e.debug_set_factor(U, sd)
        for c in picks:
            live[c].set_R(U); live[c].set_qcovstd(sd)
    e.run()
    assert e.simuind == cfg.nsimu
    np.testing.assert_array_equal(e.accepted(1), accepted_from_runlen(z["runlen"]))
    ch, ss, s2 = e.chain(1)
    np.testing.assert_array_equal(ch[:, -1].astype(np.int32), z["runlen"])
    assert e.rng(1)[0] == int(z["rng_n"])
    k = z["rows_head"].shape[0]
    assert np.max(np.abs(ch[:k, :-1] - z["rows_head"]) / scale) < 1e-7
    assert np.max(np.abs(ch[-k:, :-1] - z["rows_tail"]) / scale) < 1e-7
    np.testing.assert_allclose(ss[-k:, 0], z["ss_tail"], rtol=1e-7, atol=1e-9)
    cmf, meanf, _ = e.chaincov(1)
    assert np.max(np.abs(np.triu(cmf) - np.triu(z["chaincmat"]))) / max(np.max(np.abs(z["chaincmat"])), 1e-300) < 1e-6    # (fixture e8: covariance 0)
    for c in picks:
        live[c].run(cfg.nsimu)
        o = live[c].ch.contents
        chc, ssc, _ = e.chain(c)
        np.testing.assert_array_equal(_bits(chc), _bits(np.ctypeslib.as_array(o.chain, shape=(cfg.nsimu, 201))[:o.chainind]))
        assert e.rng(c)[0] == o.rng.n
        live[c].close()
    e.close()


def test_incremental_runs_equal_one_shot(oracle):
    """mcmcx_run(upto) in pieces (incl. pieces that end between adaptation ticks) == one call."""
    from mcmcf90_amd import engine_from_problem
    z, cfg, prob = load("c2_gauss10_am", oracle)
    ckw, pkw = _kw(z)
    e1 = engine_from_problem(ckw, pkw, nchains=64, record_accept=1)
    e1.init(); e1.run()
    e2 = engine_from_problem(ckw, pkw, nchains=64, record_accept=1)
    e2.init()
    for upto in (57, 100, 101, 777, 1234, cfg.nsimu):
        e2.run(upto)
    np.testing.assert_array_equal(e1.accept_masks(), e2.accept_masks())
    np.testing.assert_array_equal(_bits(e1.theta()), _bits(e2.theta()))
    e1.close(); e2.close()


def test_unsupported_and_bad_configs_fail_loudly():
    from mcmcf90_amd import make_config, Engine, McmcError
    with pytest.raises(McmcError):
        Engine(make_config(2, 1, nsimu=10, method=7))
    with pytest.raises(McmcError):
        Engine(make_config(2, 1, nsimu=10, scalelimit=0.9))          # mcmcinit.F90:260-263
    with pytest.raises(McmcError):
        Engine(make_config(2, 1, nsimu=0))                           # mcmc_main.F90:22-25
    e = Engine(make_config(2, 1, nsimu=10))
    with pytest.raises(McmcError):
        e.run()                                                      # 'we have not inited'
    e.setpar0([1.0, 1.0])
    e.setcmat0(np.array([[1.0, 2.0], [2.0, 1.0]]))                   # not positive definite
    e.set_target("banana")
    with pytest.raises(McmcError):
        e.init()                                                     # 'could not factor the initial covariance'
    e.close()


def test_pooled_moments_and_shard_invariance(oracle):
    """Pooled moments of the current states: equal to the host sum, and bit-identical whether 256 chains run on one
    engine or as two 128-chain shards (chain_id0 offset) combined in the fixed tree -- the multi-GPU form."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd import dist as mdist
    z, cfg, prob = load("c2_gauss10_am", oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = 250
    d = 10

    def run(n, c0):
        e = engine_from_problem(ckw, pkw, nchains=n, chain_id0=c0)
        e.init(); e.run()
        m, th = e.pooled_moments(), e.theta()
        e.close()
        return m, th

    m_all, th_all = run(256, 0)
    m_a, th_a = run(128, 0)
    m_b, th_b = run(128, 128)
    np.testing.assert_array_equal(_bits(np.vstack([th_a, th_b])), _bits(th_all))      # streams keyed by chain id
    np.testing.assert_array_equal(_bits(m_a + m_b), _bits(m_all))                     # same tree, any shard count
    mean, cov = mdist.finalize_moments(m_all, d, np.zeros(d))
    np.testing.assert_allclose(mean, th_all.mean(axis=0), rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(cov, np.cov(th_all.T), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("method,d,extra", [("dram", 256, {}), ("ram", 256, {}), ("dram", 150, {"drscale": 2.0}), ("dram", 1, {"drscale": 3.0}),
                                            ("dram", 160, {"drscale": 2.0}), ("dram", 161, {"drscale": 2.0}), ("dram", 256, {"drscale": 2.0})])
def test_extreme_dimensions(oracle, method, d, extra, monkeypatch):
    """npar at the engine's limits: 256 (AM and RAM; one adaptation tick), delayed rejection on both sides of the size
    whose two second-stage vectors still fit the LDS (150, 160: step_kernel_dr, asked for -- above npar 20 the engine would
    take the other form by itself; 161 and 256: step_kernel_dr_big, the vectors in global scratch), and npar = 1 with DR."""
    from mcmcf90_amd import engine_from_problem
    if d in (150, 160):
        monkeypatch.setenv("MCMCX_DR_BIG", "0")
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    ckw = dict(nsimu=45, adaptint=20, updatesigma=0, method=method, **extra)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=lam)
    e = engine_from_problem(ckw, pkw, nchains=66, record_accept=1)
    e.init(); e.run()
    want = {("dram", 256, 0): "step_kernel<false, false, false>", ("ram", 256, 0): "step_kernel_ram_wide", ("dram", 1, 1): "step_kernel_dr",
            ("dram", 150, 1): "step_kernel_dr", ("dram", 160, 1): "step_kernel_dr", ("dram", 161, 1): "step_kernel_dr_big", ("dram", 256, 1): "step_kernel_dr_big"}
    assert e.last_kernel() == want[(method, d, 1 if extra else 0)], e.last_kernel()
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in (0, 65):
        o = oracle.run_chain(cfg, prob, chain_id=c, continue_on_downdate_fail=True)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))
    e.close()


@pytest.mark.parametrize("method,d,extra", [("dram", 300, {}), ("ram", 300, {}), ("dram", 330, {"drscale": 2.0}), ("ram", 330, {}),
                                            # one step past the blocked SVD's last instantiation (npar 256): the lane-per-chain SVD, in the DEFAULT suite
                                            # -- initcmatn = npar gives the one adaptation a full-rank covariance, which the Jacobi finishes in a few sweeps
                                            ("dram", 257, {"condmax": 1e8, "initcmatn": 257}),
                                            # (... and with the rank-deficient covariance of 20 rows, the routine at its 60-sweep cap: minutes -- MCMCX_EXTENDED=1)
                                            pytest.param("dram", 257, {"condmax": 1e8}, marks=pytest.mark.extended), pytest.param("scam", 257, {}, marks=pytest.mark.extended)])
def test_npar_above_256(oracle, method, d, extra):
    """The reference allocates whatever npar the namelist says (MCMC_init.F90:81-102); up to round 4 the engine stopped at 256.  Above it the
    forms that keep an npar-vector per lane in LDS give way to global scratch -- the adaptation's work vector above npar 320
    (adapt_post_kernel<.., true>), the pooled-moment kernel above 317 (moments_kernel<true>), the blocked SVD to the lane-per-chain one above 256:
    slower, bit-equal.  AM, RAM (update and downdate sweeps over 45 150 elements), delayed rejection with its three factors, the SVD
    factor and SCAM, 70 chains incl. a ragged tile, one adaptation each, against the oracle -- and the pooled moments against numpy."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd import dist as mdist
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    nsimu, adaptint = (12, 5) if method == "scam" else (45, 20)
    if extra.get("initcmatn"):                          # the default-suite lane-SVD case: one adaptation (a 257 x 257 Jacobi SVD on one lane is ~20 s)
        nsimu = 25
    ckw = dict(nsimu=nsimu, adaptint=adaptint, updatesigma=0, method=method, **extra)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=lam)
    e = engine_from_problem(ckw, pkw, nchains=70, record_accept=1)
    e.init(); e.run()
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in (0, 69):
        o = oracle.run_chain(cfg, prob, chain_id=c, continue_on_downdate_fail=True)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        if cfg.usesvd:
            np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R))
        else:
            np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))
        if method != "ram":
            cm, mean, wsum = e.chaincov(c)
            np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)))
        if extra.get("drscale"):
            r2, ic = e.dr_state(c)
            np.testing.assert_array_equal(_bits(np.triu(ic)), _bits(np.triu(o.iC)))
        assert e.rng(c)[0] == o.rng_n
    mean, cov = mdist.finalize_moments(e.pooled_moments(), d, np.asarray(pkw["par0"]))
    np.testing.assert_allclose(mean, th.mean(axis=0), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cov, np.cov(th.T), rtol=1e-8, atol=1e-12)
    e.close()


@pytest.mark.parametrize("d", [315, 316])
def test_pooled_moments_at_the_lds_switch(oracle, d):
    """moments_kernel keeps a tile's 64 state vectors in LDS while (64 (npar | 1) + 320) doubles fit 160 KiB -- up to npar 315 -- and forms every
    value from global memory from 316 on (moments_kernel<true>; ADVICE round 5: the comments said 317 / 318).  Both sides of the switch: the
    pooled moment vector of 70 chains after a few Metropolis iterations against numpy, and chain 0 / 69 against the oracle."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd import dist as mdist
    assert (64 * (315 | 1) + 320) * 8 <= 160 * 1024 < (64 * (316 | 1) + 320) * 8
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    ckw = dict(nsimu=8, doadapt=0, updatesigma=0)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=A @ A.T + np.eye(d))
    e = engine_from_problem(ckw, pkw, nchains=70, record_accept=1)
    e.init(); e.run()
    th = e.theta()
    mean, cov = mdist.finalize_moments(e.pooled_moments(), d, np.asarray(pkw["par0"]))
    np.testing.assert_allclose(mean, th.mean(axis=0), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(cov, np.cov(th.T), rtol=1e-8, atol=1e-12)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for c in (0, 69):
        o = oracle.run_chain(cfg, prob, chain_id=c)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
    e.close()


@pytest.mark.parametrize("d", [7, 20, 23, 37, 64])
def test_delayed_rejection_vectors_in_lds_or_global(oracle, d, monkeypatch):
    """step_kernel_dr (second-stage vectors in LDS: the engine's choice up to npar 20) and step_kernel_dr_big (in global scratch:
    its choice above) are the same chain bit for bit, and the oracle's -- both forms at every size, a ragged tile, two adaptations."""
    from mcmcf90_amd import engine_from_problem
    rng = np.random.default_rng(100 + d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    ckw = dict(nsimu=130, adaptint=50, updatesigma=0, drscale=2.0)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.5 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
    res = []
    for big in ("0", "1"):
        monkeypatch.setenv("MCMCX_DR_BIG", big)
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=2, record_accept=1)
        e.init(); e.run()
        assert e.last_kernel() == {"0": "step_kernel_dr", "1": "step_kernel_dr_big"}[big]
        res.append((e.theta().copy(), e.accept_masks(), [e.R(c).copy() for c in (0, 69)], [e.rng(c)[0] for c in (0, 69)], [e.counters(c)["draccepted"] for c in (0, 69)]))
        e.close()
    a = res[0]
    for b in res[1:]:
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[4] == b[4]
        for x, y in zip(a[2], b[2]):
            np.testing.assert_array_equal(_bits(x), _bits(y))
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate((0, 69)):
        o = oracle.run_chain(cfg, prob, chain_id=2 + c)
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        assert a[3][i] == o.rng_n


def test_sizes_beyond_the_limits_fail_loudly():
    from mcmcf90_amd import make_config, Engine, McmcError
    with pytest.raises(McmcError):
        Engine(make_config(8193, 1, nsimu=10))                         # (npar itself is unlimited like the reference's up to int-sized packed indices)
    e = Engine(make_config(200, 4, nsimu=10, drscale=2.0, pooled=1))   # pooled delayed rejection beyond 160 (where its two vectors no longer fit
    e.setpar0(np.zeros(200)); e.set_target("banana", b=0.1)           # the LDS) runs on global scratch since round 3: no limit of its own
    e.init(); e.run()
    assert e.last_kernel() == "step_kernel_pooled_dr_big"      # (the matrix-core form stops where its LDS does, below npar 200)
    e.close()


def test_caller_supplied_stream(oracle):
    """mcmcx_set_stream: the engine enqueues on the host program's stream (ordering with the caller's own kernels and
    copies); same chain as on its private stream."""
    import torch
    from mcmcf90_amd import engine_from_problem
    z, cfg, prob = load("c2_gauss10_am", oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = 350
    ref = engine_from_problem(ckw, pkw, nchains=70)
    ref.init(); ref.run()
    st = torch.cuda.Stream()
    e = engine_from_problem(ckw, pkw, nchains=70)
    e.set_stream(st.cuda_stream)
    e.init(); e.run(); e.sync()
    st.synchronize()
    np.testing.assert_array_equal(_bits(e.theta()), _bits(ref.theta()))
    e.close(); ref.close()


def test_engines_release_their_device_memory(oracle):
    """Create / run / destroy in a loop: device memory returns to where it was (allocations are owned by the handle)."""
    import torch
    from mcmcf90_amd import engine_from_problem
    z, cfg, prob = load("c3_banana20_dram", oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = 120
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        e = engine_from_problem(ckw, pkw, nchains=4096, record_chain=1)
        e.init(); e.run(); e.sync()
        e.close()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)


@pytest.mark.parametrize("kw", [
    dict(burnintime=50, adaptint=100, doburnin=0),                  # first AM tick at 200, not at the threshold 150
    dict(burnintime=130, adaptint=100, doburnin=1, badaptint=40),   # burn-in ticks only rescale; first AM tick at 240 (threshold 230)
    dict(burnintime=37, adaptint=64, adapthist=1, doburnin=0),      # adapthist = 1: threshold 102, tick at 128
    dict(burnintime=0, adaptint=100, doburnin=0),                   # aligned (the case every fixture has)
], ids=["burn50", "burn130_b40", "hist1", "aligned"])
@pytest.mark.parametrize("dr", [0.0, 2.0], ids=["am", "dram"])
@pytest.mark.parametrize("kernels", ["lane", "auto"])          # auto: the lane-group kernels (accept bytes + group_pack_kernel feed the same ring)
def test_history_ring_without_record_chain(oracle, kw, dr, kernels):
    """record_chain = 0 (what bench.py runs): the history ring only holds the adaptation window.  Its size must cover
    the FIRST window, which is longer than burnintime + adaptint + adapthist whenever that threshold is not a multiple
    of adaptint: final state, factor and covariance bit for bit against the oracle."""
    from mcmcf90_amd import engine_from_problem
    d = 6
    rng = np.random.default_rng(11)
    A = rng.standard_normal((d, d)); lam = A @ A.T + d * np.eye(d)
    ckw = dict(nsimu=700, updatesigma=0, drscale=dr, **kw)
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.05 * np.eye(d), mu=np.zeros(d), lam=lam)
    e = engine_from_problem(ckw, pkw, nchains=70, record_accept=1, record_chain=0)
    e.init(); e.run()
    th = e.theta()
    for c in (0, 5, 64, 69):
        o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=c)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))
        cm, mean, wsum = e.chaincov(c)
        np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)))
        np.testing.assert_array_equal(_bits(mean), _bits(o.chainmean))
        assert wsum == o.chainwsum
    e.close()


@pytest.mark.parametrize("d,method,extra", [(48, "scam", {}), (49, "scam", {}), (70, "scam", {}), (64, "dram", dict(condmax=1e6)), (65, "dram", dict(condmax=1e6)),
                                            (100, "dram", dict(condmax=50.0, drscale=2.0)), (128, "scam", {}), (129, "scam", {}), (200, "dram", dict(condmax=1e6)),
                                            (200, "scam", {}), (209, "scam", {}), (225, "scam", {}),
                                            pytest.param(255, "scam", {}, marks=pytest.mark.extended), pytest.param(256, "scam", {}, marks=pytest.mark.extended)])
def test_blocked_svd_equals_lane_svd_and_oracle(oracle, d, method, extra, monkeypatch):
    """npar 48..256 with an SVD factor: MCMC_adapt's factorisation runs one workgroup per chain (mcx_svd.hpp: the pinned routine's pairs in a
    reordered but equivalent sequence, every later column streamed past a block's pair-lanes -- svd_sweep_stream32_kernel up to npar 200,
    svd_sweep_stream_kernel<26> up to 208 and <32> up to 256 (209, 225: both instances in the default suite), svd_applyv_stream32_kernel for V).
    Bit for bit the one-lane-per-chain routine (MCMCX_SVD_LANE=1: the engine's own form below npar 48 and above 256; compared up to npar = 128)
    -- factor, singular values, states after several adaptations, a ragged tile -- and the oracle on two chains.  (The four superseded
    generations of these kernels and the shared-rotation negative were compared here up to round 5: tools/variants/README.md.)"""
    from mcmcf90_amd import engine_from_problem
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.diag(10.0 ** np.linspace(-1, 2, d))
    nsimu = 45 if method == "scam" else 160
    adaptint = 14 if method == "scam" else 50
    if d >= 128:                                          # two adaptations: a lane-per-chain SVD of this size takes ~10-25 s
        adaptint = 9 if method == "scam" else 40
        nsimu = 2 * adaptint + 3
    ckw = dict(nsimu=nsimu, method=method, adaptint=adaptint, updatesigma=0, **extra)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=lam)
    res = []
    # (the lane-per-chain form takes 10-25 s per adaptation at npar 200: there it is compared with the oracle in
    #  test_gpu_fullsize.py::test_c5_illcond200_scam_replicas_two_ticks)
    # (odd npar: the V replay's extra row and unaligned columns; 65 / 129 / 209: one row past a register-count instantiation)
    paths = ("stream", "lane") if d <= 128 else ("stream",)
    for path in paths:
        monkeypatch.delenv("MCMCX_SVD_LANE", raising=False)
        if path == "lane":
            monkeypatch.setenv("MCMCX_SVD_LANE", "1")
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=5, record_accept=1)
        e.init(); e.run()
        res.append((e.theta(), e.accept_masks(), [e.R(c) for c in (0, 63, 64, 69)],
                    [e.qcovstd(c) for c in (0, 69)], [e.chaincov(c)[0] for c in (0, 69)], [e.rng(c)[0] for c in (0, 69)]))
        e.close()
    a = res[0]
    for b in res[1:]:
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1])
        for x, y in zip(a[2] + a[3] + a[4], b[2] + b[3] + b[4]):
            np.testing.assert_array_equal(_bits(x), _bits(y))
        assert a[5] == b[5]
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate((0, 69)):
        if d >= 200 and c == 0:                           # (the forms above agree on all 70 chains; one oracle chain at this size: the ragged tile's)
            continue
        o = oracle.run_chain(cfg, prob, chain_id=5 + c)
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(a[2][0 if c == 0 else 3]), _bits(o.R))
        assert a[5][i] == o.rng_n


@pytest.mark.parametrize("nw", [1, 2, 4, 8])
@pytest.mark.parametrize("kind,d", [("gauss", 37), ("banana", 20), ("expdata", 2)])
def test_scam_waves_per_tile(oracle, nw, kind, d, monkeypatch):
    """Per-chain SCAM with 1, 2, 4, 8 waves per tile (scam_kernel / scam_mw_kernel: the waves share the panels of the two
    rotations and the Gaussian target's row blocks): the same chain bit for bit, whatever the split -- ragged panels
    (d = 37: five panels, three row blocks), fewer panels than waves (d = 2), bounds, the sigma2 update, an adaptation."""
    from mcmcf90_amd import engine_from_problem
    monkeypatch.setenv("MCMCX_SCAM_WAVES", str(nw))
    rng = np.random.default_rng(d)
    if kind == "gauss":
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        ckw = dict(nsimu=33, method="scam", adaptint=15, updatesigma=0)
    elif kind == "banana":
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), b=0.1)
        ckw = dict(nsimu=60, method="scam", adaptint=25, updatesigma=0)
    else:
        z, _, _ = load("s4_expdata_scam_s2", oracle)
        ckw, pkw = _kw(z)
        ckw["nsimu"] = 250
    e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=3, record_accept=1, record_chain=1)
    e.init(); e.run()
    assert e.last_kernel() == {1: "scam_kernel", 2: "scam_mw_kernel<2>", 4: "scam_mw_kernel<4>", 8: "scam_mw_kernel<8>"}[nw], e.last_kernel()
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in (0, 64, 69):
        o = oracle.run_chain(cfg, prob, chain_id=3 + c)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta))
        chc, ssc, s2c = e.chain(c)
        np.testing.assert_array_equal(_bits(chc), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ssc), _bits(o.sschain))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2c), _bits(o.s2chain))
        np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R))
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["chainind"]) == (o.stayed, o.bndstayed, o.chainind)
        assert e.rng(c)[0] == o.rng_n
    e.close()


@pytest.mark.parametrize("case", ["am_first", "ap", "greedy", "ap_ragged"])
def test_batch_branch_in_blocks_equals_row_form_and_oracle(oracle, case, monkeypatch):
    """covmat's two-pass batch branch (matutils.F90:311-338) -- the first AM adaptation with initcmatn = 0, every adaptation of
    an AP window (adapthist > 1), the greedy restart of the burn-in -- runs in register blocks (adapt_covb_*); the row-by-row
    form (covmat_rows, MCMCX_COV_BATCH_ROWS=1) and the oracle must give the same bits: covariance, mean, weight, factor, state."""
    from mcmcf90_amd import engine_from_problem
    if case == "am_first":
        z, _, _ = load("c4_gauss50_am", oracle); ckw, pkw = _kw(z)
    elif case == "ap":
        z, _, _ = load("c1_priors_ap", oracle); ckw, pkw = _kw(z)
    elif case == "greedy":
        rng = np.random.default_rng(5)
        d = 13
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.3), cmat0=0.05 * np.eye(d), mu=np.zeros(d), lam=A @ A.T + np.eye(d))
        ckw = dict(nsimu=700, method="dram", drscale=0.0, adaptint=100, doburnin=1, burnintime=400, badaptint=50, greedy=1,
                   scalelimit=0.05, scalefactor=2.5, updatesigma=0)
    else:
        rng = np.random.default_rng(9)
        d = 37                                                    # four blocks of ten, the last one ragged
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.1), cmat0=(0.5 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        ckw = dict(nsimu=650, method="dram", drscale=0.0, adaptint=100, adapthist=230, updatesigma=0)
    res = []
    for rows in (0, 1):
        if rows:
            monkeypatch.setenv("MCMCX_COV_BATCH_ROWS", "1")
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=11, record_accept=1)
        e.init(); e.run()
        res.append((e.theta().copy(), [e.chaincov(c) for c in (0, 63, 69)], [e.R(c).copy() for c in (0, 63, 69)], [e.rng(c)[0] for c in (0, 63, 69)]))
        e.close()
    a, b = res
    np.testing.assert_array_equal(_bits(a[0]), _bits(b[0]))
    for (ca, ma, wa, *_), (cb, mb, wb, *_) in zip(a[1], b[1]):
        np.testing.assert_array_equal(_bits(ca), _bits(cb)); np.testing.assert_array_equal(_bits(ma), _bits(mb)); assert wa == wb
    for ra, rb in zip(a[2], b[2]):
        np.testing.assert_array_equal(_bits(ra), _bits(rb))
    assert a[3] == b[3]
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate((0, 63, 69)):
        o = oracle.run_chain(cfg, prob, chain_id=11 + c)
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(a[1][i][0])), _bits(np.triu(o.chaincmat)))
        np.testing.assert_array_equal(_bits(a[1][i][1]), _bits(o.chainmean))
        assert a[3][i] == o.rng_n


@pytest.mark.parametrize("d", [21, 34, 50, 64, 100])
@pytest.mark.parametrize("start", ["default", "target"])
def test_ram_wide_panels(oracle, d, start, monkeypatch):
    """method='ram' above npar 20: step_kernel_ram_wide sweeps the factor in equal column panels up to 17 wide (21: 11 + 10, 34: 17 + 17,
    50: 17 + 17 + 16, 64: four of 16, 100: six of 17 / 15) instead of panels of ten.  Same chain bit for bit as step_kernel<true, false, false>
    (MCMCX_RAM_WIDE=0) and as the oracle -- update lanes alone (default start) and update and downdate lanes mixed (start = target), a ragged
    tile, several launches, the sigma2 update and bounds at one size each."""
    from mcmcf90_amd import engine_from_problem
    rng = np.random.default_rng(140 + d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    cm0 = np.linalg.inv(lam) if start == "target" else 0.01 * np.eye(d)
    ckw = dict(nsimu=130, method="ram", adaptint=100, updatesigma=1 if d == 34 else 0)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.1), cmat0=cm0, mu=np.linspace(-0.5, 0.5, d), lam=lam)
    if d == 21:
        pkw.update(lo=np.full(d, -2.5), hi=np.full(d, 2.5))
    if d == 34:
        pkw.update(sigma2=0.9, nobs=20)
    res = []
    for narrow in (0, 1):
        if narrow:
            monkeypatch.setenv("MCMCX_RAM_WIDE", "0")
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=4, record_accept=1)
        e.init(); e.run(60); e.run(61); e.run()
        assert e.last_kernel() == ("step_kernel<true, false, false>" if narrow else "step_kernel_ram_wide")
        res.append((e.theta().copy(), e.accept_masks().copy(), [e.R(c).copy() for c in (0, 63, 69)], [e.rng(c)[0] for c in (0, 63, 69)], e.totals()["downdates"]))
        e.close()
    a, b = res
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[4] == b[4]
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(_bits(x), _bits(y))
    if start == "target":
        assert a[4] > 0
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate((0, 69)):
        o = oracle.run_chain(cfg, prob, chain_id=4 + c)
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(a[2][0 if c == 0 else 2])), _bits(np.triu(o.R)))
        assert a[3][0 if c == 0 else 2] == o.rng_n


@pytest.mark.parametrize("d", [1, 3, 7, 10])
@pytest.mark.parametrize("start", ["default", "target"])
def test_ram_factor_in_lds(oracle, d, start, monkeypatch):
    """method='ram' at npar <= 10 (one column panel) with all tiles resident: step_kernel_ram_ldsr keeps the packed factor that
    DCHUD / DCHDD rewrite at every iteration, and their rotations, in LDS for the launch.  Same chain bit for bit as the
    global-memory form (MCMCX_LDS_SCRATCH=0) and as the oracle -- update and downdate lanes mixed (start = target: about half of
    the iterations downdate), bounds, the sigma2 update, several launches."""
    from mcmcf90_amd import engine_from_problem
    rng = np.random.default_rng(40 + d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    cm0 = np.linalg.inv(lam) if start == "target" else 0.01 * np.eye(d)
    ckw = dict(nsimu=260, method="ram", adaptint=100, updatesigma=1 if d == 3 else 0)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.1), cmat0=cm0, mu=np.linspace(-0.5, 0.5, d), lam=lam)
    if d == 7:
        pkw.update(lo=np.full(d, -1.5), hi=np.full(d, 1.5))
    if d == 3:
        pkw.update(sigma2=0.9, nobs=20)
    res = []
    for off in (0, 1):
        if off:
            monkeypatch.setenv("MCMCX_LDS_SCRATCH", "0")
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=4, record_accept=1)
        e.init(); e.run(90); e.run(91); e.run()
        assert e.last_kernel() == ("step_kernel<true, false, false>" if off else "step_kernel_ram_ldsr")
        res.append((e.theta().copy(), e.accept_masks().copy(), [e.R(c).copy() for c in (0, 63, 69)], [e.rng(c)[0] for c in (0, 63, 69)], e.totals()["downdates"]))
        e.close()
    a, b = res
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[4] == b[4]
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(_bits(x), _bits(y))
    if start == "target" and d > 1:
        assert a[4] > 0                                    # downdates happened: both kinds of lanes were in the waves
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate((0, 63, 69)):
        o = oracle.run_chain(cfg, prob, chain_id=4 + c, continue_on_downdate_fail=True)
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(a[2][i])), _bits(np.triu(o.R)))
        assert a[3][i] == o.rng_n
