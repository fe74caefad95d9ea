"""Host-callback targets (the reference's link-time ssfunction / priorfun / checkbounds surface,
external_inc.h:4-33): the engine proposes on the GPU, evaluates the user's functions on the host in
chain order, decides on the GPU.  With callbacks that compute what the built-in target computes the
chain must be identical, bit for bit, to the device-resident run and to the oracle -- AM, DRAM with the
extra DR round trip, RAM, bounds, priors, sigma2 update."""
import ctypes as C
import numpy as np
import pytest
from golden_util import load

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _kw(z):
    ckw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_") and k[4:] not in ("dodr", "doscam", "usesvd")}
    pkw = {}
    for k in z.files:
        if k.startswith("prob_"):
            v = z[k]
            pkw[k[5:]] = v.item() if v.ndim == 0 else v
    return ckw, pkw


@pytest.mark.parametrize("name,nsimu", [("c1_shipped_nml", 1000), ("c3_banana20_dram", 700), ("c4_gauss50_ram", 400),
                                        ("c2_gauss10_am", 450)])
@pytest.mark.parametrize("plumbing", ["fused_mapped", "phase_launches_and_copies"])
def test_host_callbacks_equal_device_target(oracle, name, nsimu, plumbing, monkeypatch):
    """The user's ssfunction / priorfun / checkbounds on the host (external_inc.h:4-33): the iteration cut where the reference calls them.
    plumbing (round 5): with few chains an iteration is launch and wake-up latency, so by default the exchange vectors live in mapped host
    memory (no copies) and an iteration's last phase shares its launch with the next iteration's proposal; MCMCX_HOST_MAPPED=0 /
    MCMCX_HOST_FUSE=0 is the form of rounds 1-4 (device buffers, one launch per phase).  Both against the oracle bit for bit, the run cut in
    two calls (the proposal that rode ahead must not be lost or repeated across mcmcx_run calls)."""
    if plumbing != "fused_mapped":
        monkeypatch.setenv("MCMCX_HOST_MAPPED", "0"); monkeypatch.setenv("MCMCX_HOST_FUSE", "0")
    from mcmcf90_amd import Engine, make_config
    z, cfg, prob = load(name, oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = nsimu
    cfg = oracle.make_cfg(**ckw)
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_ssfun.restype = C.c_double; L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)
    calls = {"ss": 0, "cb": 0}

    def ssfun(th):
        calls["ss"] += 1
        return L.mcxo_ssfun(C.byref(tgt), th.ctypes.data_as(dp))

    def priorfun(th):
        return L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp))

    def checkbounds(th):
        calls["cb"] += 1
        return bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp)))

    npar, nch = int(pkw["npar"]), 3
    e = Engine(make_config(npar, nch, record_chain=1, chain_id0=5, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(float(pkw.get("sigma2", 1.0)), int(pkw.get("nobs", 1)))
    e.set_target_host(ssfun, priorfun, checkbounds)
    e.init(); e.run(137); e.run(138); e.run()
    for c in range(nch):
        o = oracle.run_chain(cfg, prob, chain_id=5 + c)
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ss), _bits(o.sschain))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain))
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"]) == \
               (o.stayed, o.bndstayed, o.draccepted, o.drtries)
        assert e.rng(c)[0] == o.rng_n
        np.testing.assert_array_equal(_bits(np.triu(e.R(c))), _bits(np.triu(o.R)))
    # the reference evaluates ss only for in-bounds candidates, once per proposal (+ the first point)
    tot = e.totals()
    assert calls["cb"] == tot["proposals"] + nch
    if not cfg.dodr:
        assert calls["ss"] == tot["proposals"] + nch - tot["bndstayed"]
    e.close()


@pytest.mark.parametrize("early", [False, True])
def test_host_callbacks_early_rejection(oracle, early):
    """method='er' through the host callbacks (MCMC_run_er.F90:54-101): checkbounds and priorfun first, the threshold
    drawn on the device, then ssfunction_er with each chain's own sscrit.  `early`: a user function that really stops
    summing once it is past sscrit -- the decisions, hence the chain, are the same."""
    from mcmcf90_amd import Engine, make_config
    z, cfg, prob = load("e6_expdata_er_priors", oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = 600
    cfg = oracle.make_cfg(**ckw)
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_ssfun.restype = C.c_double; L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)
    x, y = np.asarray(pkw["xdata"], float), np.asarray(pkw["ydata"], float)
    stopped = {"n": 0}

    def ssfun(th):
        return L.mcxo_ssfun(C.byref(tgt), th.ctypes.data_as(dp))

    def ssfun_er(th, crit):
        if not early:
            return ssfun(th)
        full = ssfun(th)
        if full < crit:
            return full                                            # accepted proposals must carry the exact ss
        part = 0.0
        for xi, yi in zip(x, y):                                    # a rejected one may stop at the first partial sum past sscrit
            part += (yi - th[0] * np.exp(-th[1] * xi)) ** 2
            if part >= crit:
                stopped["n"] += 1
                return part
        return full

    npar, nch = int(pkw["npar"]), 3
    e = Engine(make_config(npar, nch, record_chain=1, chain_id0=40, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(float(pkw.get("sigma2", 1.0)), int(pkw.get("nobs", 1)))
    e.set_target_host(ssfun, lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))), ssfun_er=ssfun_er)
    e.init(); e.run()
    for c in range(nch):
        o = oracle.run_chain(cfg, prob, chain_id=40 + c)
        ch, ss, s2 = e.chain(c)
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ss), _bits(o.sschain))
        np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain))
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["erstayed"]) == (o.stayed, o.bndstayed, o.erstayed)
        assert e.rng(c)[0] == o.rng_n
    if early:
        assert stopped["n"] > 0
    e.close()


@pytest.mark.parametrize("name", ["m1_expdata2_dram_dr_s2", "m2_expdata2_er_s2", "m3_expdata2_scam_s2", "m4_expdata2_ram",
                                  "m5_expdata2_burnin_greedy_priors"])
def test_two_response_columns(oracle, name):
    """nycol = 2: the user's ssfunction returns one ss per response column, sigma2 / nobs are vectors, MCMC_alpha and the
    DR / ER formulas sum over the columns, the sigma2 update draws one gamma per column, sschain has three columns and
    s2chain two.  Fixtures from the real reference; the engine through the host callbacks against fixture and oracle."""
    from mcmcf90_amd import Engine, make_config
    from golden_util import accepted_from_runlen
    z, cfg, prob = load(name, oracle)
    ckw, pkw = _kw(z)
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int
    dp = C.POINTER(C.c_double)

    def ssfun(th):
        out = np.zeros(2)
        L.mcxo_ssfun_cols(C.byref(tgt), th.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out

    cid = int(z["chain_id"])
    npar, nch = 3, 3
    e = Engine(make_config(npar, nch, record_chain=1, record_accept=1, chain_id0=cid - 1, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(pkw["sigma2"], pkw["nobs"])
    e.set_target_host(ssfun, lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)),
                      lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))))
    e.init(); e.run()
    # the fixture's chain is engine chain 1
    np.testing.assert_array_equal(e.accepted(1), accepted_from_runlen(z["runlen"]))
    ch, ss, s2 = e.chain(1)
    assert ss.shape[1] == 3 and (not cfg.updatesigma or s2.shape[1] == 2)
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(ch[-k:, :-1], z["rows_tail"], rtol=1e-7)
    np.testing.assert_allclose(ss[-k:, :-1], z["ss_tail"], rtol=1e-7)
    if cfg.updatesigma:
        np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-7)
    assert e.rng(1)[0] == int(z["rng_n"])
    for c in range(nch):
        o = oracle.run_chain(cfg, prob, chain_id=cid - 1 + c, continue_on_downdate_fail=True)
        chc, ssc, s2c = e.chain(c)
        np.testing.assert_array_equal(_bits(chc), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ssc), _bits(o.sschain))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2c), _bits(o.s2chain))
        assert e.rng(c)[0] == o.rng_n
    e.close()


@pytest.mark.parametrize("form", ["one_launch", "phase_kernels"])
@pytest.mark.parametrize("name", ["m1_expdata2_dram_dr_s2", "m2_expdata2_er_s2", "m3_expdata2_scam_s2", "m4_expdata2_ram",
                                  "m5_expdata2_burnin_greedy_priors"])
def test_two_response_columns_on_the_device_in_one_launch(oracle, name, form, monkeypatch):
    """nycol = 2 with the device-resident response-column target WITHOUT the phase kernels: step_kernel_cols runs a whole segment of
    iterations in one launch (the sums over the columns in MCMC_alpha / MCMC_sscrit / MCMC_DR_alpha13, one gamma draw per column:
    MCMC_DRAM.F90:100-135,162-206, MCMC_init.F90:119-132).  The reference's two-column fixtures -- run-length column, stream position,
    rows, ss and sigma2 tails -- and the oracle bit for bit, 130 chains incl. a ragged tile; the phase-kernel form (MCMCX_COLS_PHASED=1)
    is the same chain."""
    from mcmcf90_amd import engine_from_problem
    from golden_util import accepted_from_runlen
    if form == "phase_kernels":
        monkeypatch.setenv("MCMCX_COLS_PHASED", "1")
    z, cfg, prob = load(name, oracle)
    ckw, pkw = _kw(z)
    cid = int(z["chain_id"])
    e = engine_from_problem(ckw, pkw, nchains=130, chain_id0=cid - 1, record_chain=1, record_accept=1)
    e.init(); e.run(57); e.run()                                      # a launch boundary off the adaptation ticks
    if form == "one_launch":
        assert e.last_kernel() == ("step_kernel_cols<scam>" if cfg.doscam else "step_kernel_cols<ram>" if cfg.method == 1 else "step_kernel_cols"), e.last_kernel()
    else:
        assert e.last_kernel() == ""                                   # no sampling-kernel table entry: the iteration is cut into phase launches
    np.testing.assert_array_equal(e.accepted(1), accepted_from_runlen(z["runlen"]))
    ch, ss, s2 = e.chain(1)
    assert ss.shape[1] == 3 and (not cfg.updatesigma or s2.shape[1] == 2)
    k = z["rows_head"].shape[0]
    np.testing.assert_allclose(ch[-k:, :-1], z["rows_tail"], rtol=1e-7)
    np.testing.assert_allclose(ss[-k:, :-1], z["ss_tail"], rtol=1e-7)
    if cfg.updatesigma:
        np.testing.assert_allclose(s2[-k:], z["s2_tail"], rtol=1e-7)
    assert e.rng(1)[0] == int(z["rng_n"])
    for c in (0, 1, 64, 129):
        o = oracle.run_chain(cfg, prob, chain_id=cid - 1 + c, continue_on_downdate_fail=True)
        chc, ssc, s2c = e.chain(c)
        np.testing.assert_array_equal(_bits(chc), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ssc), _bits(o.sschain))
        if cfg.updatesigma:
            np.testing.assert_array_equal(_bits(s2c), _bits(o.s2chain))
        assert e.rng(c)[0] == o.rng_n
        cnt = e.counters(c)
        assert (cnt["stayed"], cnt["bndstayed"], cnt["draccepted"], cnt["drtries"], cnt["erstayed"]) == (o.stayed, o.bndstayed, o.draccepted, o.drtries, o.erstayed)
    e.close()


@pytest.mark.parametrize("target", ["device", "host"])
@pytest.mark.parametrize("method,extra", [("dram", dict(drscale=2.0, updatesigma=1)), ("er", dict()), ("ram", dict())])
def test_twelve_response_columns(oracle, target, method, extra):
    """nycol = 12 (round 5: the engine took at most eight; the reference takes whatever mcmcnycol.dat says, MCMC_init.F90:119-132): npar = 13, the
    response-column model on the device in one launch, and the same model through the host callbacks, against the oracle bit for bit."""
    import ctypes as C
    from mcmcf90_amd import Engine, make_config, engine_from_problem
    ny = 12
    r = np.random.default_rng(5)
    x = np.arange(9.0)
    rates = np.linspace(0.08, 0.3, ny)
    Y = np.vstack([9.0 * np.exp(-k * x) + r.standard_normal(x.size) * 0.3 for k in rates])
    ckw = dict(dict(nsimu=160, method=method, adaptint=50, updatesigma=0, N0=1.0, S02=0.6), **extra)
    pkw = dict(kind="expdata", npar=1 + ny, par0=np.concatenate([[9.0], rates]), cmat0=np.diag([0.01] + [0.0001] * ny) * (10.0 if method == "ram" else 1.0),
               sigma2=r.uniform(0.4, 1.0, ny), nobs=r.integers(5, 12, ny), xdata=x, ydata=Y, lo=np.zeros(1 + ny))
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    if target == "device":
        e = engine_from_problem(ckw, pkw, nchains=67, chain_id0=3, record_chain=1)
    else:
        L = oracle.lib(); tgt = prob.ctarget(); dp = C.POINTER(C.c_double)
        L.mcxo_priorfun.restype = C.c_double; L.mcxo_checkbounds.restype = C.c_int

        def ssfun(th):
            out = np.zeros(ny)
            L.mcxo_ssfun_cols(C.byref(tgt), th.ctypes.data_as(dp), out.ctypes.data_as(dp))
            return out
        e = Engine(make_config(1 + ny, 67, record_chain=1, chain_id0=3, **ckw))
        e.setpar0(prob.par0); e.setcmat0(prob.cmat0); e.setsigma2nobs(prob.sigma2v, prob.nobsv)
        e.set_target_host(ssfun, lambda th: L.mcxo_priorfun(C.byref(tgt), th.ctypes.data_as(dp)), lambda th: bool(L.mcxo_checkbounds(C.byref(tgt), th.ctypes.data_as(dp))))
    e.init(); e.run()
    if target == "device":
        assert e.last_kernel() == ("step_kernel_cols<ram>" if method == "ram" else "step_kernel_cols"), e.last_kernel()
    for c in (0, 66):
        o = oracle.run_chain(cfg, prob, chain_id=3 + c, continue_on_downdate_fail=True)
        ch, ss, s2 = e.chain(c)
        assert ss.shape[1] == ny + 1
        np.testing.assert_array_equal(_bits(ch), _bits(o.chain))
        np.testing.assert_array_equal(_bits(ss), _bits(o.sschain))
        if cfg.updatesigma:
            assert s2.shape[1] == ny
            np.testing.assert_array_equal(_bits(s2), _bits(o.s2chain))
        assert e.rng(c)[0] == o.rng_n
    e.close()


def test_two_columns_need_the_host_target():
    from mcmcf90_amd import Engine, make_config, McmcError
    e = Engine(make_config(2, 1, nsimu=10))
    e.setpar0([1.0, 1.0]); e.setsigma2nobs([1.0, 1.0], [5, 5]); e.set_target("banana", b=0.1)
    with pytest.raises(McmcError):
        e.init()
    e.close()
