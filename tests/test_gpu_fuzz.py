"""Randomised configurations: the engine against the oracle, bit for bit, over the namelist's option space at small
sizes -- method x delayed rejection x burn-in scaling x greedy x AP window x adaptend x initcmatn x sigma2 update x
bounds x priors x SVD factor (condmax) x target, with seeds fixed so a failure reproduces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _draw(seed):
    r = np.random.default_rng(1000 + seed)
    method = r.choice(["dram", "dram", "ram", "scam", "er"])
    d = int(r.integers(1, 13))
    kind = r.choice(["gauss", "gauss", "banana", "expdata"])
    if kind == "banana" and d < 2:
        d = 2
    if kind == "expdata":
        d = 2
    nsimu = int(r.integers(60, 260))
    ckw = dict(nsimu=nsimu, method=str(method), adaptint=int(r.choice([7, 20, 50])), updatesigma=int(r.integers(0, 2)))
    if method == "dram":
        if r.random() < 0.5:
            ckw["drscale"] = float(r.choice([2.0, 3.0, 5.0]))
        if r.random() < 0.4:
            ckw.update(doburnin=1, burnintime=int(r.integers(20, 120)), scalelimit=float(r.choice([0.05, 0.3])),
                       greedy=int(r.integers(0, 2)))
        if r.random() < 0.25:
            ckw["adapthist"] = int(r.choice([30, 80]))
        if r.random() < 0.3:
            ckw["condmax"] = float(r.choice([1e8, 1e12]))
    if method in ("dram", "er") and r.random() < 0.3:
        ckw["adaptend"] = int(nsimu * 0.6)
    if r.random() < 0.3:
        ckw["initcmatn"] = int(r.integers(1, 60))
    if method == "ram":
        ckw.update(alphatarget=float(r.choice([0.234, 0.4])), nuparam=float(r.choice([0.6, 0.7, 0.9])))
        if r.random() < 0.3:
            ckw.update(doburnin=1, burnintime=int(r.integers(10, 60)))
    if kind == "gauss":
        A = r.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=r.standard_normal(d) * 0.3, cmat0=np.diag(r.uniform(0.05, 0.6, d)),
                   mu=r.standard_normal(d) * 0.2, lam=A @ A.T + np.diag(r.uniform(0.5, 2.0, d)))
    elif kind == "banana":
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=float(r.uniform(0.05, 0.5)) * np.eye(d), b=float(r.choice([0.03, 0.1])))
    else:
        x = np.arange(11.0)
        y = 9.0 * np.exp(-0.1 * x) + r.standard_normal(11) * 0.3
        pkw = dict(kind="expdata", npar=2, par0=[9.0, 0.12], cmat0=[[0.2, 0], [0, 0.001]], xdata=x, ydata=y, lo=[0, 0])
    if ckw["updatesigma"]:
        pkw.update(sigma2=float(r.uniform(0.3, 1.5)), nobs=int(r.integers(5, 40)))
        ckw.update(N0=float(r.choice([1.0, 4.0])), S02=float(r.choice([0.0, 0.8])))
    if kind != "expdata" and r.random() < 0.3:
        pkw.update(lo=np.full(d, -1.5), hi=np.full(d, 1.8))
    if r.random() < 0.3:
        pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(r.random(d) < 0.5, 0.0, 1.0))
    return ckw, pkw


def _check_against_oracle(oracle, ckw, pkw, seed, want_kernel=None):
    from mcmcf90_amd import engine_from_problem
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    from mcmcf90_amd import McmcError
    e = engine_from_problem(ckw, pkw, nchains=67, chain_id0=3 * seed, record_accept=1, record_chain=1)
    try:
        oracle.run_chain(oracle.make_cfg(**dict(ckw, nsimu=2)), prob, chain_id=3 * seed)
    except RuntimeError:                                      # the reference stops in MCMC_init (e.g. dpotri of an SVD factor
        with pytest.raises(McmcError):                        # whose upper triangle is singular): so must the engine
            e.init()
        e.close()
        return
    e.init(); e.run()
    if want_kernel:
        assert e.last_kernel() == want_kernel, e.last_kernel()
    for c in (0, 63, 66):
        o = oracle.run_chain(cfg, prob, chain_id=3 * seed + c, continue_on_downdate_fail=True)
        if o.rc <= -2001:
            # the reference STOPS inside MCMC_adapt here ("cannot invert cmat", MCMC_adapt.F90:220-223: dpotri on the upper triangle of an SVD
            # factor U sqrt(s) that has an exact zero on its diagonal -- two of 85 000 random configurations, tools/bigfuzz.py seeds 52160 / 61152).
            # The engine cannot stop one chain of a launch: it raises the chain's status bit 4 (ST_POTRI_FAIL), keeps the inverse it had and goes
            # on; up to the iteration the reference died at the two are the same chain.
            ns = o.simuind
            assert e.counters(c)["status"] & 4, (ckw, c)
            np.testing.assert_array_equal(e.accepted(c)[:ns], o.accepted[:ns], err_msg=str(ckw))
            ch, ss, s2 = e.chain(c)
            np.testing.assert_array_equal(_bits(ch[:o.chainind - 1]), _bits(o.chain[:o.chainind - 1]), err_msg=str(ckw))
            continue
        assert o.rc == 0, (ckw, o.rc)
        np.testing.assert_array_equal(e.accepted(c), o.accepted, err_msg=str(ckw))
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain), err_msg=str(ckw))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain), err_msg=str(ckw))
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == \
               (o.stayed, o.bndstayed, o.draccepted, o.drtries, o.erstayed), ckw
        assert e.rng(c)[0] == o.rng_n, ckw
        if cfg.usesvd:
            np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R), err_msg=str(ckw))
        else:
            np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)), err_msg=str(ckw))
        cm, mean, wsum = e.chaincov(c)
        if cfg.method != 1:                                   # RAM never touches chaincmat
            np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)), err_msg=str(ckw))
            assert wsum == o.chainwsum, ckw
    e.close()


# (the lane-per-chain kernels on every seed, the engine's own choice -- the lane-group kernels wherever they cover the draw -- on every other one:
#  tools/bigfuzz.py runs thousands of seeds on each family once per round)
@pytest.mark.parametrize("seed,kernels", [(s, "lane") for s in range(300)] + [(s, "auto") for s in range(0, 300, 2)])
def test_random_configuration(oracle, seed, kernels):
    ckw, pkw = _draw(seed)
    _check_against_oracle(oracle, ckw, pkw, seed)


def _check_host_callbacks_against_oracle(oracle, seed, nch=3):
    """The draw of `seed` with the target on the HOST: ssfunction / priorfun / checkbounds are the oracle's own C target functions behind the
    engine's callback interface (external_inc.h:4-33), every method incl. SCAM's componentwise loop and early rejection's two evaluations."""
    import ctypes as C
    from mcmcf90_amd import Engine, make_config
    ckw, pkw = _draw(seed)
    ckw["nsimu"] = min(int(ckw["nsimu"]), 140)
    if "adaptend" in ckw:
        ckw["adaptend"] = min(int(ckw["adaptend"]), 100)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    try:
        oracle.run_chain(oracle.make_cfg(**dict(ckw, nsimu=2)), prob, chain_id=7 * seed)
    except RuntimeError:
        return "init stops"
    L = oracle.lib(); tgt = prob.ctarget()
    L.mcxo_ssfun.restype = C.c_double; L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)
    npar = int(pkw["npar"])
    e = Engine(make_config(npar, nch, record_chain=1, chain_id0=7 * seed, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(float(pkw.get("sigma2", 1.0)), int(pkw.get("nobs", 1)))
    e.set_target_host(lambda th: L.mcxo_ssfun(C.byref(tgt), th.ctypes.data_as(dp)), lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))))
    e.init(); e.run(int(ckw["nsimu"]) // 3); e.run()
    try:
        for c in range(nch):
            o = oracle.run_chain(cfg, prob, chain_id=7 * seed + c, continue_on_downdate_fail=True)
            if o.rc != 0:
                assert o.rc <= -2001 and (e.counters(c)["status"] & 4), (ckw, o.rc)
                continue
            ch, ss, s2 = e.chain(c)
            np.testing.assert_array_equal(_bits(ch), _bits(o.chain), err_msg=str(ckw))
            np.testing.assert_array_equal(_bits(ss), _bits(o.sschain), err_msg=str(ckw))
            if cfg.updatesigma:
                np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain), err_msg=str(ckw))
            cnt = e.counters(c)
            assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == (o.stayed, o.bndstayed, o.draccepted, o.drtries, o.erstayed), ckw
            assert e.rng(c)[0] == o.rng_n, ckw
    finally:
        e.close()
    return str(ckw["method"])


@pytest.mark.parametrize("plumbing", ["fused_mapped", "phase_launches_and_copies"])
@pytest.mark.parametrize("seed", range(24))
def test_random_configuration_host_callbacks(oracle, seed, plumbing, monkeypatch):
    """Random configurations with the user's functions on the host, in the two plumbing forms (round 5: mapped host memory and fused phases, or
    device buffers and one launch per phase); tools/bigfuzz.py-style loops: tools/host_fuzz.py."""
    if plumbing != "fused_mapped":
        monkeypatch.setenv("MCMCX_HOST_MAPPED", "0"); monkeypatch.setenv("MCMCX_HOST_FUSE", "0")
    _check_host_callbacks_against_oracle(oracle, seed)


@pytest.mark.parametrize("seed", [52160, 61152])
def test_reference_stops_inside_an_adaptation(oracle, seed):
    """Two draws of tools/bigfuzz.py (round 5, 85 000 configurations) in which the REFERENCE terminates inside MCMC_adapt for one of the checked
    chains: delayed rejection with condmax > 0 and an SVD factor whose upper triangle has an exact zero on the diagonal, so dpotri fails
    ("cannot invert cmat", MCMC_adapt.F90:220-223).  The engine flags that chain (status bit 4) and continues; up to there the chains agree,
    and the other chains of the run are compared in full."""
    ckw, pkw = _draw(seed)
    _check_against_oracle(oracle, ckw, pkw, seed)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    assert any(oracle.run_chain(cfg, prob, chain_id=3 * seed + c, continue_on_downdate_fail=True).rc <= -2001 for c in (0, 63, 66))


def _draw_ram_svd(seed):
    """method='ram' with condmax > 0: the factor is covtor's full SVD factor, proposals are matmulx(R,u) and
    cholupdate / choldowndate work on its upper triangle (MCMC_run_ram.F90:96-97, 168-172)."""
    ckw, pkw = _draw(20000 + seed)
    r = np.random.default_rng(21000 + seed)
    for k in ("drscale", "greedy", "adapthist", "adaptend", "scalelimit", "doburnin", "burnintime"):
        ckw.pop(k, None)
    ckw.update(method="ram", condmax=float(r.choice([1e8, 1e12])), alphatarget=float(r.choice([0.234, 0.4])),
               nuparam=float(r.choice([0.6, 0.7, 0.9])))
    if r.random() < 0.3:
        ckw.update(doburnin=1, burnintime=int(r.integers(10, 60)))
    d = int(pkw["npar"])
    if pkw["kind"] != "expdata" and d > 1:                    # a dense initial covariance: a factor with a full lower triangle
        A = r.standard_normal((d, d)) / np.sqrt(d)
        pkw["cmat0"] = (A @ A.T + np.eye(d)) * float(r.uniform(0.02, 0.2))
    return ckw, pkw


@pytest.mark.parametrize("seed", range(60))
def test_random_configuration_ram_with_svd_factor(oracle, seed):
    ckw, pkw = _draw_ram_svd(seed)
    _check_against_oracle(oracle, ckw, pkw, seed, want_kernel="step_kernel_ram_fullr")


@pytest.mark.parametrize("kernels", ["lane", "auto"])
@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_in_pieces(oracle, seed, kernels):
    """mcmcx_run(upto) called in random pieces (launch boundaries anywhere relative to the adaptation ticks) ends in the
    same state as one call, for random configurations."""
    from mcmcf90_amd import engine_from_problem, McmcError
    ckw, pkw = _draw(5000 + seed)
    r = np.random.default_rng(seed)
    e1 = engine_from_problem(ckw, pkw, nchains=65, record_accept=1)
    try:
        e1.init()
    except McmcError:
        e1.close()
        return
    e1.run()
    e2 = engine_from_problem(ckw, pkw, nchains=65, record_accept=1)
    e2.init()
    cuts = sorted(set(int(v) for v in r.integers(2, ckw["nsimu"], size=6))) + [ckw["nsimu"]]
    for upto in cuts:
        e2.run(upto)
    np.testing.assert_array_equal(e1.accept_masks(), e2.accept_masks(), err_msg=str((ckw, cuts)))
    np.testing.assert_array_equal(_bits(e1.theta()), _bits(e2.theta()), err_msg=str((ckw, cuts)))
    e1.close(); e2.close()


@pytest.mark.parametrize("seed", range(60))
def test_random_configuration_host_callbacks(oracle, seed):
    """The same option space through the host-callback path (the user's ssfunction / priorfun / checkbounds /
    ssfunction_er): identical to the device-resident target."""
    ckw, pkw = _draw(9000 + seed)
    _check_host_callbacks(oracle, ckw, pkw)


@pytest.mark.parametrize("seed", range(12))
def test_random_configuration_ram_with_svd_factor_host_callbacks(oracle, seed):
    ckw, pkw = _draw_ram_svd(500 + seed)
    _check_host_callbacks(oracle, ckw, pkw)


def _check_host_callbacks(oracle, ckw, pkw):
    import ctypes as C
    from mcmcf90_amd import Engine, make_config, engine_from_problem, McmcError
    ckw["nsimu"] = min(ckw["nsimu"], 120 if ckw["method"] != "scam" else 40)
    prob = oracle.Problem(**pkw)
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_ssfun.restype = C.c_double; L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)
    ref = engine_from_problem(ckw, pkw, nchains=3, chain_id0=7, record_accept=1)
    try:
        ref.init()
    except McmcError:
        ref.close()
        return
    ref.run()
    npar = int(pkw["npar"])
    e = Engine(make_config(npar, 3, chain_id0=7, record_accept=1, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(float(pkw.get("sigma2", 1.0)), int(pkw.get("nobs", 1)))
    e.set_target_host(lambda th: L.mcxo_ssfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))))
    e.init(); e.run()
    np.testing.assert_array_equal(_bits(e.theta()), _bits(ref.theta()), err_msg=str(ckw))
    for c in range(3):                                        # (the ballots' bits of the tile's unused lanes differ: no callbacks there)
        np.testing.assert_array_equal(e.accepted(c), ref.accepted(c), err_msg=str(ckw))
        assert e.rng(c)[0] == ref.rng(c)[0], ckw
    e.close(); ref.close()


@pytest.mark.parametrize("kernels", ["lane", "auto"])
@pytest.mark.parametrize("seed", range(72))
def test_random_configuration_larger_npar(oracle, seed, kernels):
    """npar 13..64: two to seven column panels of the RAM sweeps (panel width 10), several 8 x 8 blocks of the
    covariance update and of the Cholesky factorisation, ragged last panels.  RAM is drawn half of the time, half of
    those started at the target's own covariance -- waves that mix update and downdate lanes (whole-segment stores,
    pipelined sweeps) -- with burn-in, bounds, priors and the sigma2 update mixed in."""
    _check_larger_npar(oracle, seed)


@pytest.mark.parametrize("seed", range(500000, 500012))
def test_random_configuration_npar_65_to_300(oracle, seed):
    """The same draws at npar 65..300 (round 5: beyond the group kernels, beyond the LDS-resident forms, beyond the old npar limit), one in ten with
    an SVD factor up to npar 130; tools/bignpar_fuzz.py runs hundreds."""
    _check_larger_npar(oracle, seed, 65, 301)


def _check_scam_npar(oracle, seed, dlo=13, dhi=121):
    """method = 'scam' (MCMC_run_scam.F90:38-138) at npar 13..120: the componentwise loop on the per-chain rotation, adaptations through the
    lane-per-chain SVD (npar < 48) and the blocked Jacobi (from 48 on), Gaussian and banana targets, bounds, priors, the sigma2 update."""
    from mcmcf90_amd import engine_from_problem
    r = np.random.default_rng(8000 + seed)
    d = int(r.integers(dlo, dhi))
    ckw = dict(nsimu=int(r.integers(25, 60)), method="scam", adaptint=int(r.choice([10, 20])), updatesigma=int(r.random() < 0.3))
    if r.random() < 0.3:
        ckw["condmax"] = float(r.choice([1e6, 50.0, 1e10]))
    kind = "banana" if r.random() < 0.3 else "gauss"
    if kind == "gauss":
        A = r.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=r.standard_normal(d) * 0.1, cmat0=np.diag(r.uniform(0.2, 1.0, d)) / d, mu=np.zeros(d), lam=A @ A.T + np.eye(d))
    else:
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=(0.5 / d) * np.eye(d), b=0.1)
    if ckw["updatesigma"]:
        pkw.update(sigma2=float(r.uniform(0.5, 1.5)), nobs=int(r.integers(5, 40)))
    if r.random() < 0.25:
        pkw.update(lo=np.full(d, -2.5), hi=np.full(d, 2.5))
    if r.random() < 0.25:
        pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(r.random(d) < 0.5, 0.0, 2.0))
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=66, chain_id0=seed, record_accept=1)
    e.init(); e.run(int(ckw["nsimu"]) // 2); e.run()
    th = e.theta()
    try:
        for c in (0, 65):
            o = oracle.run_chain(cfg, prob, chain_id=seed + c)
            assert o.rc == 0, (ckw, o.rc)
            np.testing.assert_array_equal(e.accepted(c), o.accepted, err_msg=str(ckw))
            np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta), err_msg=str(ckw))
            np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R), err_msg=str(ckw))
            np.testing.assert_array_equal(_bits(e.qcovstd(c)), _bits(o.qcovstd), err_msg=str(ckw))
            assert e.rng(c)[0] == o.rng_n, ckw
    finally:
        e.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_scam_configuration_npar_13_to_120(oracle, seed):
    _check_scam_npar(oracle, seed)


def _check_larger_npar(oracle, seed, dlo=13, dhi=65):
    from mcmcf90_amd import engine_from_problem
    r = np.random.default_rng(7000 + seed)
    d = int(r.integers(dlo, dhi))
    method = str(r.choice(["ram", "ram", "dram", "er"]))
    ckw = dict(nsimu=int(r.integers(50, 130)), method=method, adaptint=int(r.choice([15, 40])), updatesigma=int(r.random() < 0.3))
    if dlo > 64 and method == "dram" and d <= 130 and r.random() < 0.3:
        ckw.update(condmax=float(r.choice([1e6, 50.0])), adaptint=40)
    if method == "dram" and r.random() < 0.5:
        ckw["drscale"] = 2.0
    if method == "dram" and r.random() < 0.3:
        ckw.update(doburnin=1, burnintime=int(r.integers(20, 60)), greedy=int(r.integers(0, 2)), scalelimit=0.3)
    if method == "ram":
        ckw.update(alphatarget=float(r.choice([0.234, 0.4])), nuparam=float(r.choice([0.6, 0.7, 0.9])))
    A = r.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    cmat0 = np.diag(r.uniform(0.2, 1.0, d)) / d
    if method == "ram" and r.random() < 0.5:
        cmat0 = np.linalg.inv(lam)                 # at the target acceptance rate: most iterations downdate
    pkw = dict(kind="gauss", npar=d, par0=r.standard_normal(d) * 0.1, cmat0=cmat0, mu=np.zeros(d), lam=lam)
    if ckw["updatesigma"]:
        pkw.update(sigma2=float(r.uniform(0.5, 1.5)), nobs=int(r.integers(5, 40)))
    if r.random() < 0.25:
        pkw.update(lo=np.full(d, -2.5), hi=np.full(d, 2.5))
    if r.random() < 0.25:
        pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(r.random(d) < 0.5, 0.0, 2.0))
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=66, chain_id0=seed, record_accept=1)
    try:
        oracle.run_chain(oracle.make_cfg(**dict(ckw, nsimu=2)), prob, chain_id=seed)
    except RuntimeError:                                      # delayed rejection + condmax with a diagonal cmat0 that is not sorted: U is a permutation,
        from mcmcf90_amd import McmcError                     # the factor's upper triangle singular -- the reference stops in MCMC_init ("cannot invert
        with pytest.raises(McmcError):                        # cmat"), and so must the engine
            e.init()
        e.close()
        return
    e.init(); e.run()
    th = e.theta()
    for c in (0, 65):
        o = oracle.run_chain(cfg, prob, chain_id=seed + c, continue_on_downdate_fail=True)
        if o.rc <= -2001:                                    # (the reference stops inside MCMC_adapt: see _check_against_oracle)
            assert e.counters(c)["status"] & 4, ckw
            continue
        np.testing.assert_array_equal(e.accepted(c), o.accepted, err_msg=str(ckw))
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta), err_msg=str(ckw))
        if cfg.usesvd:
            np.testing.assert_array_equal(_bits(e.R(c)), _bits(o.R), err_msg=str(ckw))
        else:
            np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)), err_msg=str(ckw))
        assert e.rng(c)[0] == o.rng_n, ckw
        assert bool(e.counters(c)["status"] & 1) == (o.ram_downdate_fail != 0), ckw
        if method != "ram":
            cm, mean, wsum = e.chaincov(c)
            np.testing.assert_array_equal(_bits(np.triu(cm)), _bits(np.triu(o.chaincmat)), err_msg=str(ckw))
    e.close()


@pytest.mark.parametrize("seed,waves", [(s, 1) for s in range(30)] + [(s, 2) for s in range(30, 90)])
def test_pooled_am_matrix_core_kernel_equals_lane_kernel(seed, waves, monkeypatch):
    """pooled_mfma_kernel (products as MFMA tiles) == the lane-per-chain pooled kernel, for random sizes (one to three
    passes of output blocks, ragged last block and k-block), targets, bounds, priors and the sigma2 update.  waves = 2: the
    instance that shares a SIMD between two waves (pooled_mfma_kernel<false, true>; the engine's own choice from 2048 tiles on -- the kernel
    behind bench.py's c4_pooled: sixty of the ninety draws go to it, the suite time the two-waves-per-tile variants used to take); the last
    thirty of them run pooled_mfma_ks_kernel (the LDS vector in two pieces of forty rows) where npar is 41..64."""
    from mcmcf90_amd import engine_from_problem
    monkeypatch.setenv("MCMCX_POOLED_WAVES", str(waves))
    r = np.random.default_rng(11000 + seed)
    d = int(r.choice([1, 2, 3, 5, 16, 17, 31, 48, 50, 63, 64, 65, 80, 97, 130])) if seed < 30 else int(r.choice([17, 18, 25, 31, 32, 33, 47, 48, 49, 50, 63, 64]))
    kind = str(r.choice(["gauss", "gauss", "banana"])) if d >= 2 else "gauss"
    n = int(r.choice([65, 130, 200]))
    ckw = dict(nsimu=int(r.integers(20, 60)), adaptint=int(r.choice([8, 15])), updatesigma=int(r.integers(0, 2)))
    if kind == "gauss":
        A = r.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=r.standard_normal(d) * 0.1, cmat0=np.diag(r.uniform(0.2, 1.0, d)) / d,
                   mu=r.standard_normal(d) * 0.1, lam=A @ A.T + np.eye(d))
    else:
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=(0.3 / d) * np.eye(d), b=0.05)
    if ckw["updatesigma"]:
        pkw.update(sigma2=0.8, nobs=20)
    if r.random() < 0.4:
        pkw.update(lo=np.full(d, -1.0), hi=np.full(d, 1.2))
    if r.random() < 0.4:
        pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(r.random(d) < 0.5, 0.0, 1.0))
    ks = waves == 2 and seed >= 60 and 40 < d <= 64
    monkeypatch.setenv("MCMCX_POOLED_KS", "1" if ks else "0")
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
    e.init(); e.run()
    assert e.last_kernel() == ("pooled_mfma_ks_kernel" if ks else {1: "pooled_mfma_kernel<false>", 2: "pooled_mfma_kernel<false, true>"}[waves]), e.last_kernel()
    monkeypatch.setenv("MCMCX_POOLED_SCALAR", "1")
    e2 = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
    e2.init(); e2.run()
    np.testing.assert_array_equal(e.accept_masks(), e2.accept_masks(), err_msg=str((d, kind, n, ckw)))
    np.testing.assert_array_equal(_bits(e.theta()), _bits(e2.theta()), err_msg=str((d, kind, n, ckw)))
    np.testing.assert_array_equal(_bits(e.scalars()), _bits(e2.scalars()), err_msg=str((d, kind, n, ckw)))
    np.testing.assert_array_equal(_bits(e.pooled()[3]), _bits(e2.pooled()[3]))
    e.close(); e2.close()


@pytest.mark.parametrize("seed", range(24))
def test_pooled_scam_substeps_equal_lane_kernel(seed):
    """Without adaptation the pooled SCAM (one rotation, MFMA tiles, up to 16 waves per tile) and the per-chain SCAM
    kernel (lane per chain, every chain its own copy of the same initial rotation) must produce the same chains:
    random sizes over every block / group wave split, targets, bounds, priors, sigma2 update."""
    from mcmcf90_amd import engine_from_problem
    r = np.random.default_rng(13000 + seed)
    d = int(r.choice([2, 3, 7, 15, 16, 17, 33, 48, 63, 64, 65, 81, 100, 113, 129]))
    kind = str(r.choice(["gauss", "gauss", "banana"]))
    n = int(r.choice([65, 130]))
    ckw = dict(nsimu=int(r.integers(4, 9)), method="scam", doadapt=0, updatesigma=int(r.integers(0, 2)))
    if kind == "gauss":
        A = r.standard_normal((d, d)) / np.sqrt(d)
        pkw = dict(kind="gauss", npar=d, par0=r.standard_normal(d) * 0.1, cmat0=np.diag(r.uniform(0.2, 1.0, d)) / d,
                   mu=r.standard_normal(d) * 0.1, lam=A @ A.T + np.eye(d))
    else:
        pkw = dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=np.diag(r.uniform(0.2, 0.4, d)) / d, b=0.05)
    if ckw["updatesigma"]:
        pkw.update(sigma2=0.8, nobs=20)
    if r.random() < 0.4:
        pkw.update(lo=np.full(d, -1.0), hi=np.full(d, 1.2))
    if r.random() < 0.4:
        pkw.update(pri_mu=np.zeros(d), pri_sig=np.where(r.random(d) < 0.5, 0.0, 1.0))
    e = engine_from_problem(ckw, pkw, nchains=n, pooled=1, record_accept=1)
    e.init(); e.run()
    e2 = engine_from_problem(ckw, pkw, nchains=n, pooled=0, record_accept=1)
    e2.init(); e2.run()
    np.testing.assert_array_equal(e.accept_masks(), e2.accept_masks(), err_msg=str((d, kind, n, ckw)))
    np.testing.assert_array_equal(_bits(e.theta()), _bits(e2.theta()), err_msg=str((d, kind, n, ckw)))
    np.testing.assert_array_equal(_bits(e.scalars()), _bits(e2.scalars()), err_msg=str((d, kind, n, ckw)))
    for c in (0, n - 1):
        assert e.rng(c)[0] == e2.rng(c)[0]
    e.close(); e2.close()


def _draw_cols(seed):
    """Random configuration with nycol = 2 or 3 response columns (model column j: th0 exp(-th(1+j) x), npar = 1 + nycol)."""
    r = np.random.default_rng(31000 + seed)
    ny = int(r.choice([2, 3]))
    method = str(r.choice(["dram", "dram", "er", "scam", "ram"]))
    nsimu = int(r.integers(80, 300)) if method != "scam" else int(r.integers(40, 100))
    ckw = dict(nsimu=nsimu, method=method, adaptint=int(r.choice([20, 50])), updatesigma=int(r.integers(0, 2)),
               N0=float(r.choice([1.0, 4.0])), S02=float(r.choice([0.0, 0.8])))
    if method == "dram":
        if r.random() < 0.5:
            ckw["drscale"] = float(r.choice([2.0, 3.0]))
        if r.random() < 0.4:
            ckw.update(doburnin=1, burnintime=int(r.integers(20, 120)), scalelimit=float(r.choice([0.05, 0.3])), greedy=int(r.integers(0, 2)))
        if r.random() < 0.25:
            ckw["adapthist"] = 60
    x = np.arange(11.0)
    rates = np.array([0.1, 0.25, 0.18])[:ny]
    Y = np.vstack([9.0 * np.exp(-k * x) + r.standard_normal(11) * 0.3 for k in rates])
    npar = 1 + ny
    scale = 10.0 if method == "ram" else 1.0
    pkw = dict(kind="expdata", npar=npar, par0=np.concatenate([[9.0], rates]), cmat0=scale * np.diag([0.02] + [0.0002] * ny),
               sigma2=r.uniform(0.3, 1.0, ny), nobs=r.integers(8, 20, ny), xdata=x, ydata=Y, lo=np.zeros(npar))
    if r.random() < 0.3:
        pkw.update(pri_mu=np.concatenate([[9.0], rates]), pri_sig=np.concatenate([[1.0], np.where(r.random(ny) < 0.5, 0.0, 0.1)]))
    return ckw, pkw


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_response_columns(oracle, seed):
    """nycol = 2 or 3 through the host callbacks against the oracle, bit for bit."""
    import ctypes as C
    from mcmcf90_amd import Engine, make_config, McmcError
    ckw, pkw = _draw_cols(seed)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)
    ny, npar = prob.ny, prob.npar

    def ssfun(th):
        out = np.zeros(ny)
        L.mcxo_ssfun_cols(C.byref(tgt), th.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out

    e = Engine(make_config(npar, 3, record_chain=1, chain_id0=2 * seed, **ckw))
    e.setpar0(prob.par0); e.setcmat0(prob.cmat0); e.setsigma2nobs(prob.sigma2v, prob.nobsv)
    e.set_target_host(ssfun, lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))))
    e.init(); e.run()
    for c in range(3):
        o = oracle.run_chain(cfg, prob, chain_id=2 * seed + c, continue_on_downdate_fail=True)
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain), err_msg=str(ckw))
        np.testing.assert_array_equal(_bits(ss), _bits(o.sschain), err_msg=str(ckw))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain), err_msg=str(ckw))
        assert e.rng(c)[0] == o.rng_n, ckw
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == \
               (o.stayed, o.bndstayed, o.draccepted, o.drtries, o.erstayed), ckw
    e.close()


@pytest.mark.parametrize("seed", range(40))
def test_random_configuration_response_columns_device_target(oracle, seed):
    """nycol = 2 or 3 with the device-resident response-column target (mcmcx_set_target_expdata_cols: the phase kernels
    with dev_eval_kernel between them, no host round trip) against the oracle, bit for bit."""
    from mcmcf90_amd import engine_from_problem
    ckw, pkw = _draw_cols(seed)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=67, record_chain=1, chain_id0=2 * seed)
    e.init(); e.run()
    # one launch per segment (round 5), not phase kernels; one instantiation per method class (round 6)
    assert e.last_kernel() == ("step_kernel_cols<scam>" if cfg.doscam else "step_kernel_cols<ram>" if cfg.method == 1 else "step_kernel_cols"), e.last_kernel()
    for c in (0, 1, 66):
        o = oracle.run_chain(cfg, prob, chain_id=2 * seed + c, continue_on_downdate_fail=True)
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain), err_msg=str(ckw))
        np.testing.assert_array_equal(_bits(ss), _bits(o.sschain), err_msg=str(ckw))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain), err_msg=str(ckw))
        assert e.rng(c)[0] == o.rng_n, ckw
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == \
               (o.stayed, o.bndstayed, o.draccepted, o.drtries, o.erstayed), ckw
    e.close()


@pytest.mark.parametrize("seed", range(30))
def test_random_configuration_gamma_shape_below_one(oracle, seed):
    """updatesigma with N0/2 + nobs/2 < 1: random_gamma's boost branch, gammar_mt(1+a, b) * u**(1/a) (mcmcrand.F90:102-105)."""
    ckw, pkw = _draw(40000 + seed)
    r = np.random.default_rng(41000 + seed)
    ckw.update(updatesigma=1, N0=float(r.choice([0.2, 0.5, 0.9])), S02=float(r.choice([0.0, 0.8])))
    pkw.update(sigma2=float(r.uniform(0.3, 1.5)), nobs=1)
    _check_against_oracle(oracle, ckw, pkw, seed)


@pytest.mark.parametrize("seed", range(10))
def test_response_columns_gamma_shape_below_one_in_one_column(oracle, seed):
    from mcmcf90_amd import engine_from_problem
    ckw, pkw = _draw_cols(600 + seed)
    ckw.update(updatesigma=1, N0=0.5)
    nobs = np.asarray(pkw["nobs"]).copy(); nobs[0] = 1
    pkw["nobs"] = nobs
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=5, record_chain=1, chain_id0=seed)
    e.init(); e.run()
    for c in (0, 4):
        o = oracle.run_chain(cfg, prob, chain_id=seed + c, continue_on_downdate_fail=True)
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain), err_msg=str(ckw))
        np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain), err_msg=str(ckw))
        assert e.rng(c)[0] == o.rng_n, ckw
    e.close()


@pytest.mark.parametrize("seed", range(8))
def test_ram_at_target_acceptance_in_pieces(oracle, seed):
    """RAM started at its target acceptance rate (cmat0 = the target's covariance: most iterations are Cholesky
    downdates, after which the proposal accumulates from the diagonal up), mcmcx_run cut at random iterations so that
    the per-chain order flag crosses launch boundaries; three chains against the oracle, bit for bit."""
    from mcmcf90_amd import engine_from_problem
    r = np.random.default_rng(777 + seed)
    d = int(r.integers(3, 24))
    A = r.standard_normal((d, d)) / np.sqrt(d)
    lam = A @ A.T + np.eye(d)
    ckw = dict(nsimu=int(r.integers(150, 400)), method="ram", updatesigma=0, alphatarget=0.234, nuparam=float(r.choice([0.6, 0.7])))
    pkw = dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=np.linalg.inv(lam), mu=np.zeros(d), lam=lam)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=seed, record_accept=1)
    e.init()
    cuts = sorted(set(int(v) for v in r.integers(2, ckw["nsimu"], size=7))) + [ckw["nsimu"]]
    for upto in cuts:
        e.run(upto)
    th = e.theta()
    ndown = 0
    for c in (0, 64, 69):
        o = oracle.run_chain(cfg, prob, chain_id=seed + c, continue_on_downdate_fail=True)
        np.testing.assert_array_equal(e.accepted(c), o.accepted, err_msg=str((ckw, cuts)))
        np.testing.assert_array_equal(_bits(th[c]), _bits(o.theta), err_msg=str((ckw, cuts)))
        np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)), err_msg=str((ckw, cuts)))
        assert e.rng(c)[0] == o.rng_n
        ndown += int((~o.accepted.astype(bool)).sum())
    assert ndown > 0.4 * 3 * ckw["nsimu"]                       # the regime the test is about: mostly rejections
    e.close()
