"""The lane-group-per-chain step kernel (mcmcf90_amd/csrc/mcx_group.hpp: four chains per wave, R / R2 / iC in registers, sixteen
polar attempts of a chain at a time) against the lane-per-chain kernels and the oracle: the same chain bit for bit --
state, accept ballots, stream position, counters, the decoded chain, the adapted factors and covariance
(MCMC_run.F90:41-107, MCMC_DRAM.F90:20-31,100-186, mcmcrand.F90:166-190)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _problem(kind, d, seed, priors=False, bounds=False):
    rng = np.random.default_rng(seed)
    pkw = dict(kind=kind, npar=d, par0=np.full(d, 0.05), cmat0=(0.5 / d) * np.eye(d))
    if kind == "gauss":
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        pkw.update(mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
    elif kind == "banana":
        pkw.update(b=0.1, cmat0=(2.0 / d) * np.eye(d))
    else:
        x = np.linspace(0.0, 6.0, 21)
        y = 1.5 * np.exp(-0.4 * x) + 0.05 * rng.standard_normal(x.size)
        pkw.update(xdata=x, ydata=y, par0=np.concatenate([[1.0, 0.5], np.zeros(d - 2)])[:d], cmat0=0.01 * np.eye(d))
    if priors:
        sig = np.where(np.arange(d) % 3 == 1, -1.0, 2.0)            # every third parameter flat
        pkw.update(pri_mu=np.full(d, 0.1), pri_sig=sig)
    if bounds:
        pkw.update(lo=np.full(d, -1.5), hi=np.full(d, 1.2))
    return pkw


def _expected_group_kernel(d, drscale, gw):
    """The entry of mcx_api.hip's GROUP_TABLE a configuration runs: quads when asked for at npar <= 16; with delayed rejection the
    instantiation without R2 when drscale is a power of two (npar <= 24), the general one otherwise."""
    import math
    quad = gw == 4 and d <= 16
    if not drscale > 0.0:
        return "group_step_kernel<quad>" if quad else "group_step_kernel"
    pow2 = math.frexp(drscale)[0] == 0.5 and d <= 24
    if quad:
        return "group_step_kernel<quad, DR2>" if pow2 else "group_step_kernel<quad, DR>"
    return "group_step_kernel<DR2>" if pow2 else "group_step_kernel<DR>"


def _run(ckw, pkw, nchains, group, monkeypatch, chains=(0, 1, 5, 69), gw=16, **ekw):
    from mcmcf90_amd import engine_from_problem
    monkeypatch.setenv("MCMCX_GROUP", "1" if group else "0")
    monkeypatch.setenv("MCMCX_GROUP_GW", str(gw))
    e = engine_from_problem(ckw, pkw, nchains=nchains, chain_id0=3, record_accept=1, **ekw)
    e.init(); e.run()
    k = e.last_kernel()
    assert k.startswith("group_step_kernel") == bool(group), k
    if group:
        assert k == _expected_group_kernel(int(pkw["npar"]), float(ckw.get("drscale", 0.0)) if ckw.get("method", "dram") != "er" else 0.0, gw), k
    chains = [c for c in chains if c < nchains]
    out = dict(theta=e.theta().copy(), masks=e.accept_masks().copy(), scal=e.scalars().copy(),
               rng=[e.rng(c) for c in chains], ctr=[e.counters(c) for c in chains], R=[e.R(c).copy() for c in chains])
    if ckw.get("drscale", 0) > 0:
        out["dr"] = [e.dr_state(c) for c in chains]
    if ckw.get("doadapt", 1):
        out["cov"] = [e.chaincov(c) for c in chains]
    if ekw.get("record_chain"):
        out["chain"] = [e.chain(c) for c in chains]
    e.close()
    return out, chains


def _same(a, b):
    assert np.array_equal(_bits(a["theta"]), _bits(b["theta"])), "state"
    assert np.array_equal(a["masks"], b["masks"]), "accept ballots"
    assert np.array_equal(_bits(a["scal"]), _bits(b["scal"])), "ss1 / pri1 / sigma2 / alpha12"
    assert a["rng"] == b["rng"] and a["ctr"] == b["ctr"], (a["rng"], b["rng"], a["ctr"], b["ctr"])
    for x, y in zip(a["R"], b["R"]):
        np.testing.assert_array_equal(_bits(np.triu(x)), _bits(np.triu(y)))
    for key in ("dr", "cov", "chain"):
        if key in a:
            for x, y in zip(a[key], b[key]):
                for u, v in zip(x, y):
                    np.testing.assert_array_equal(_bits(u), _bits(v))


CASES = [
    # kind, npar, delayed rejection, priors, bounds
    ("banana", 20, 2.0, False, False),           # BASELINE config 3's shape
    ("banana", 20, 0.0, False, False),
    ("gauss", 20, 2.0, True, True),
    ("gauss", 13, 3.0, False, True),             # odd npar: the cached second deviate changes hands every iteration
    ("gauss", 11, 0.0, True, False),
    ("gauss", 16, 0.0, False, False),
    ("gauss", 17, 2.0, False, False),            # one parameter in the second slot
    ("banana", 5, 2.0, True, True),
    ("expdata", 2, 2.0, False, False),           # BASELINE config 1's model, device-resident
    ("expdata", 2, 0.0, True, True),
    ("gauss", 24, 2.0, False, False),
    ("gauss", 29, 0.0, False, True),
    ("gauss", 32, 2.0, True, False),
    ("banana", 31, 1.5, False, False),
    ("gauss", 50, 0.0, False, False),            # BASELINE config 4's target with method = 'dram': four slots, R in registers, one wave per SIMD
    ("gauss", 37, 0.0, True, True),
    ("banana", 64, 0.0, False, False),
]


@pytest.mark.parametrize("kind,d,drscale,priors,bounds", CASES, ids=["%s%d%s%s%s" % (k, d, "_dr" if s else "", "_pri" if p else "", "_bnd" if b else "") for k, d, s, p, b in CASES])
def test_group_kernel_equals_lane_kernels_and_oracle(oracle, monkeypatch, kind, d, drscale, priors, bounds):
    """130 iterations with two adaptations (the factors are re-read from the adaptation's output), 70 chains: a ragged last tile and
    a last wave with two chains of padding."""
    ckw = dict(nsimu=130, adaptint=50, updatesigma=0, drscale=drscale)
    pkw = _problem(kind, d, 40 + d, priors, bounds)
    g, chains = _run(ckw, pkw, 70, True, monkeypatch)
    l, _ = _run(ckw, pkw, 70, False, monkeypatch)
    _same(g, l)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate(chains[:3]):
        o = oracle.run_chain(cfg, prob, chain_id=3 + c)
        np.testing.assert_array_equal(_bits(g["theta"][c]), _bits(o.theta))
        assert g["rng"][i][0] == o.rng_n and g["ctr"][i]["stayed"] == o.stayed
        np.testing.assert_array_equal(_bits(np.triu(g["R"][i])), _bits(np.triu(o.R)))


@pytest.mark.parametrize("kind,d,drscale", [("expdata", 2, 2.0), ("gauss", 7, 0.0), ("banana", 20, 2.0), ("gauss", 13, 3.0)])
def test_group_kernel_with_sigma2_update(oracle, monkeypatch, kind, d, drscale):
    """updatesigma = 1 (MCMC_updatesigma2, MCMC_DRAM.F90:192-206: the gamma sampler's rejection loops on the chain's own stream, between
    the iteration's accept decision and the next iteration's normals; shape below and above 1): sigma2 chain, state and stream
    position equal to the lane kernels' and the oracle's."""
    from mcmcf90_amd import engine_from_problem
    for N0, nobs in ((1.0, 11), (0.2, 1)):
        ckw = dict(nsimu=120, adaptint=50, updatesigma=1, drscale=drscale, N0=N0, S02=0.3)
        pkw = dict(_problem(kind, d, 90 + d), sigma2=0.5, nobs=nobs)
        res = []
        for group in (True, False):
            monkeypatch.setenv("MCMCX_GROUP", "1" if group else "0")
            e = engine_from_problem(ckw, pkw, nchains=66, chain_id0=9, record_chain=1, record_accept=1)
            e.init(); e.run()
            assert e.last_kernel().startswith("group_step_kernel") == group
            res.append((e.theta().copy(), e.scalars().copy(), e.accept_masks().copy(), [e.rng(c) for c in (0, 65)], [e.chain(c) for c in (0, 65)]))
            e.close()
        a, b = res
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1])) and np.array_equal(a[2], b[2]) and a[3] == b[3]
        for x, y in zip(a[4], b[4]):
            for u, v in zip(x, y):
                np.testing.assert_array_equal(_bits(u), _bits(v))
        o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=9)
        np.testing.assert_array_equal(_bits(a[4][0][2]), _bits(o.s2chain))
        np.testing.assert_array_equal(_bits(a[0][0]), _bits(o.theta))
        assert a[3][0][0] == o.rng_n


QUAD_CASES = [c for c in CASES if c[1] <= 16] + [("gauss", 10, 0.0, False, False), ("gauss", 1, 0.0, False, False), ("gauss", 15, 4.0, True, True), ("banana", 16, 3.0, False, False)]


@pytest.mark.parametrize("kind,d,drscale,priors,bounds", QUAD_CASES, ids=["%s%d%s%s%s" % (k, d, "_dr" if s else "", "_pri" if p else "", "_bnd" if b else "") for k, d, s, p, b in QUAD_CASES])
def test_quad_group_kernel_equals_lane_kernels_and_oracle(oracle, monkeypatch, kind, d, drscale, priors, bounds):
    """The same kernel with FOUR lanes per chain (sixteen chains per wave: small npar with the chip full): npar <= 16, every target,
    both delayed-rejection instantiations, with the sigma2 update and with early rejection."""
    for extra in (dict(), dict(updatesigma=1, N0=1.0, S02=0.3), dict(method="er") if not drscale else dict(adaptint=30)):
        ckw = dict(dict(nsimu=130, adaptint=50, updatesigma=0, drscale=drscale), **extra)
        pkw = _problem(kind, d, 540 + d, priors, bounds)
        g, chains = _run(ckw, pkw, 70, True, monkeypatch, gw=4)
        l, _ = _run(ckw, pkw, 70, False, monkeypatch)
        _same(g, l)
        o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=3)
        np.testing.assert_array_equal(_bits(g["theta"][0]), _bits(o.theta))
        assert g["rng"][0][0] == o.rng_n and g["ctr"][0]["stayed"] == o.stayed


@pytest.mark.parametrize("kind,d,priors,bounds", [("gauss", 9, True, True), ("banana", 20, False, False), ("expdata", 2, True, True)])
def test_group_kernel_early_rejection(oracle, monkeypatch, kind, d, priors, bounds):
    """method = 'er' (MCMC_run_er.F90:46-104: the threshold u is drawn before ss is looked at; a candidate the prior alone rejects counts in
    erstayed): the lane kernels' chain and the oracle's."""
    ckw = dict(nsimu=130, adaptint=50, updatesigma=0, method="er")
    pkw = _problem(kind, d, 140 + d, priors, bounds)
    g, chains = _run(ckw, pkw, 70, True, monkeypatch)
    l, _ = _run(ckw, pkw, 70, False, monkeypatch)
    _same(g, l)
    o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=3)
    np.testing.assert_array_equal(_bits(g["theta"][0]), _bits(o.theta))
    assert g["rng"][0][0] == o.rng_n and g["ctr"][0]["erstayed"] == o.erstayed and g["ctr"][0]["stayed"] == o.stayed


def test_group_kernel_power_of_two_drscale_and_its_range_check(oracle, monkeypatch):
    """drscale = 2**k takes the instantiation that forms the second-stage proposal as (R'z) / drscale (exact while every nonzero
    element of R lies in [2**-500, 2**500]); a factor outside that range raises the device flag and the general instantiation
    (R2 as stored) runs instead -- both the lane kernels' chain bit for bit.  drscale = 3 always takes the general one."""
    from mcmcf90_amd import engine_from_problem
    d = 20
    for drscale, tiny, kernel in ((4.0, False, "group_step_kernel<DR2>"), (0.5, True, "group_step_kernel<DR2>"), (3.0, False, "group_step_kernel<DR>")):
        pkw = _problem("gauss", d, 77)
        if tiny:
            cm = np.array(pkw["cmat0"]); cm[3, 3] = 1e-304; pkw["cmat0"] = cm          # R(3,3) ~ 5e-153 < 2**-500
        ckw = dict(nsimu=60, adaptint=50, updatesigma=0, drscale=drscale, doadapt=0 if tiny else 1)
        res = []
        for group in (True, False):
            monkeypatch.setenv("MCMCX_GROUP", "1" if group else "0")
            e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=5, record_accept=1)
            e.init(); e.run()
            if group:
                assert e.last_kernel() == kernel, e.last_kernel()
            res.append((e.theta().copy(), e.accept_masks().copy(), [e.rng(c) for c in (0, 69)], [e.counters(c) for c in (0, 69)]))
            e.close()
        a, b = res
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3], (drscale, tiny)
        assert sum(c["drtries"] for c in a[3]) > 0


@pytest.mark.parametrize("kind,d,drscale,nch", [("gauss", 50, 0.0, 70), ("gauss", 64, 0.0, 70), ("gauss", 37, 2.0, 70), ("gauss", 33, 0.0, 130), ("gauss", 16, 2.0, 200),
                                                 ("gauss", 1, 0.0, 70), ("gauss", 48, 3.0, 70), ("gauss", 64, 2.0, 66), ("banana", 20, 2.0, 1030), ("gauss", 10, 0.0, 1030),
                                                 ("gauss", 49, 0.0, 130),
                                                 ("gauss", 20, 2.0, 70), ("banana", 31, 3.0, 70), ("gauss", 17, 0.0, 70), ("gauss", 5, 2.0, 70)])
def test_tile_factorisation_kernel(oracle, monkeypatch, kind, d, drscale, nch):
    """tile_factor_kernel (round 5: dpotf2, and dtrti2 / dlauu2 with delayed rejection, on the packed matrices of 4 NW neighbouring chains in LDS -- one to
    four columns / rows per lane, workgroups of 4, 2 or 1 waves, a tile's workgroups on one XCD) against adapt_post_kernel's own factorisation
    (MCMCX_TILE_FACTOR=0): R, R2, iC, the states and ballots after three adaptations bit for bit, ragged tiles and ragged workgroups, chain counts past one
    XCD round; and against the oracle."""
    from mcmcf90_amd import engine_from_problem
    ckw = dict(nsimu=170, adaptint=50, updatesigma=0, drscale=drscale)
    pkw = _problem(kind, d, 700 + d)
    picks = (0, 63, nch - 1)
    res = []
    for tf in ("1", "0"):
        monkeypatch.delenv("MCMCX_GROUP", raising=False)
        monkeypatch.setenv("MCMCX_TILE_FACTOR", tf)
        e = engine_from_problem(ckw, pkw, nchains=nch, chain_id0=4, record_accept=1)
        e.init(); e.run()
        res.append((e.theta().copy(), e.accept_masks().copy(), [e.R(c).copy() for c in picks], [e.dr_state(c) for c in picks] if drscale else [],
                    [e.counters(c) for c in picks], e.totals()))
        e.close()
    a, b = res
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[4] == b[4] and a[5] == b[5]
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(_bits(np.triu(x)), _bits(np.triu(y)))
    for x, y in zip(a[3], b[3]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(_bits(np.triu(u)), _bits(np.triu(v)))
    for i, c in enumerate(picks[::2]):
        o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=4 + c)
        np.testing.assert_array_equal(_bits(np.triu(a[2][2 * i])), _bits(np.triu(o.R)))
        np.testing.assert_array_equal(_bits(a[0][c]), _bits(o.theta))
        if drscale:
            np.testing.assert_array_equal(_bits(np.triu(a[3][2 * i][1])), _bits(np.triu(o.iC)))


RAM_CASES = [
    # kind, npar, start ("default": cmat0 small, RAM adapts by updates; "target": cmat0 = the target's covariance, about half of the iterations downdate), extras
    ("gauss", 50, "default", {}),                 # BASELINE config 4's shape: four slots, the instantiation for npar <= 56
    ("gauss", 50, "target", {}),
    ("gauss", 64, "target", {}),
    ("gauss", 57, "default", {}),
    ("gauss", 33, "target", dict(updatesigma=1)),
    ("gauss", 17, "target", dict(bounds=True)),
    ("gauss", 16, "default", dict(priors=True)),
    ("gauss", 7, "target", dict(doburnin=1, burnintime=40)),
    ("gauss", 1, "target", {}),
    ("banana", 20, "target", {}),
    ("expdata", 2, "default", dict(bounds=True, updatesigma=1)),
    ("gauss", 50, "target", dict(record_chain=1)),
    ("gauss", 24, "fail", {}),                    # alphatarget near 1 and a large step: choldowndate fails (INFO = -1) on some chains, which go on flagged
]


@pytest.mark.parametrize("kind,d,start,extra", RAM_CASES, ids=["%s%d_%s%s" % (k, d, st, "_" + "_".join(sorted(x)) if x else "") for k, d, st, x in RAM_CASES])
def test_group_ram_kernel_equals_lane_kernels_and_oracle(oracle, monkeypatch, kind, d, start, extra):
    """group_ram_kernel (round 5: method = 'ram' with sixteen lanes per chain, the factor in registers for the launch, DCHUD and DCHDD performed on it
    there -- MCMC_run_ram.F90:45-179, dchud.f:122-139, dchdd.f:141-179) against the lane-per-chain RAM kernels and the oracle: state, factor, ballots,
    counters incl. the downdate count and the status bit, stream position, bit for bit -- update and downdate chains mixed in one wave, the proposal
    order after a successful downdate, a failed downdate, bounds (alpha12 left stale), priors, the sigma2 update, burn-in, launches cut at odd places,
    a recorded chain, a ragged tile and a wave with padding chains."""
    from mcmcf90_amd import engine_from_problem
    extra = dict(extra)
    pkw = _problem(kind, d, 900 + d, extra.pop("priors", False), extra.pop("bounds", False))
    ekw = dict(record_chain=1) if extra.pop("record_chain", 0) else {}
    ckw = dict(dict(nsimu=130, method="ram", adaptint=100, updatesigma=0), **extra)
    if start == "target" and kind == "gauss":
        pkw["cmat0"] = np.linalg.inv(np.asarray(pkw["lam"]))
    if start == "fail":
        ckw.update(alphatarget=0.99, nuparam=0.05)
        pkw["cmat0"] = 4.0 * np.linalg.inv(np.asarray(pkw["lam"]))
    if ckw.get("updatesigma"):
        pkw.update(sigma2=0.8, nobs=15)
    picks = (0, 1, 63, 69)
    res = []
    for group in ("1", "0"):
        monkeypatch.setenv("MCMCX_RAM_GROUP", group)
        e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=6, record_accept=1, **ekw)
        e.init(); e.run(60); e.run(61); e.run()
        assert (e.last_kernel() == "group_ram_kernel") == (group == "1"), e.last_kernel()
        res.append(dict(theta=e.theta().copy(), masks=e.accept_masks().copy(), scal=e.scalars().copy(), rng=[e.rng(c) for c in picks], ctr=[e.counters(c) for c in picks],
                        R=[e.R(c).copy() for c in picks], tot=e.totals(), chain=[e.chain(c) for c in picks[:2]] if ekw else []))
        e.close()
    a, b = res
    assert np.array_equal(_bits(a["theta"]), _bits(b["theta"])) and np.array_equal(a["masks"], b["masks"]) and np.array_equal(_bits(a["scal"]), _bits(b["scal"]))
    assert a["rng"] == b["rng"] and a["ctr"] == b["ctr"] and a["tot"] == b["tot"], (a["ctr"], b["ctr"], a["tot"], b["tot"])
    for x, y in zip(a["R"], b["R"]):
        np.testing.assert_array_equal(_bits(np.triu(x)), _bits(np.triu(y)))
    for x, y in zip(a["chain"], b["chain"]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(_bits(u), _bits(v))
    if start == "target" and d > 1:
        assert a["tot"]["downdates"] > 0
    if start == "fail":
        assert a["tot"]["status"] & 1, a["tot"]
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    for i, c in enumerate(picks):
        o = oracle.run_chain(cfg, prob, chain_id=6 + c, continue_on_downdate_fail=True)
        np.testing.assert_array_equal(_bits(a["theta"][c]), _bits(o.theta))
        np.testing.assert_array_equal(_bits(np.triu(a["R"][i])), _bits(np.triu(o.R)))
        assert a["rng"][i][0] == o.rng_n and a["ctr"][i]["stayed"] == o.stayed and a["ctr"][i]["bndstayed"] == o.bndstayed
        assert bool(a["ctr"][i]["status"] & 1) == (o.ram_downdate_fail != 0)


def test_group_kernel_full_chain_and_burnin(oracle, monkeypatch):
    """record_chain (every accepted row through the ring, ballots from the accept bytes), burn-in scaling + greedy restart, launches
    cut at 256 iterations and by mcmcx_run calls of odd lengths."""
    from mcmcf90_amd import engine_from_problem
    d = 20
    ckw = dict(nsimu=700, adaptint=60, updatesigma=0, drscale=2.0, doburnin=1, burnintime=150, badaptint=30, greedy=1, scalelimit=0.1, scalefactor=2.0)
    pkw = _problem("banana", d, 7)
    res = []
    for group in (True, False):
        monkeypatch.setenv("MCMCX_GROUP", "1" if group else "0")
        e = engine_from_problem(ckw, pkw, nchains=66, chain_id0=11, record_chain=1, record_accept=1)
        e.init()
        for upto in (17, 18, 301, 700):
            e.run(upto)
        assert e.last_kernel().startswith("group_step_kernel") == group
        res.append((e.theta().copy(), e.accept_masks().copy(), [e.chain(c) for c in (0, 65)], [e.rng(c) for c in (0, 65)], [e.chaincov(c) for c in (0, 65)], [e.counters(c) for c in (0, 65)]))
        e.close()
    a, b = res
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[5] == b[5]
    for x, y in zip(a[2] + a[4], b[2] + b[4]):
        for u, v in zip(x, y):
            np.testing.assert_array_equal(_bits(u), _bits(v))
    o = oracle.run_chain(oracle.make_cfg(**ckw), oracle.Problem(**pkw), chain_id=11)
    np.testing.assert_array_equal(_bits(a[0][0]), _bits(o.theta))
    assert a[3][0][0] == o.rng_n
    ch = a[2][0][0]
    assert ch.shape[0] == o.chain.shape[0] and np.array_equal(_bits(ch), _bits(o.chain))


def test_group_kernel_config3_fixture(oracle, monkeypatch):
    """BASELINE config 3's fixture from the real reference (banana d = 20, DRAM): run-length column and stream position."""
    from golden_util import load
    from mcmcf90_amd import engine_from_problem
    z, cfg, prob = load("c3_banana20_dram", oracle)
    monkeypatch.setenv("MCMCX_GROUP", "1")
    ckw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_") and k[4:] not in ("dodr", "doscam", "usesvd")}
    pkw = {k[5:]: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files if k.startswith("prob_")}
    e = engine_from_problem(ckw, pkw, nchains=5, chain_id0=int(z["chain_id"]), record_chain=1)
    e.init(); e.run()
    assert e.last_kernel().startswith("group_step_kernel")
    ch, ss, s2 = e.chain(0)
    assert np.array_equal(ch[:, -1].astype(np.int64), np.asarray(z["runlen"], dtype=np.int64))
    assert e.rng(0)[0] == int(z["rng_n"])
    e.close()


def test_engine_picks_quads_for_config2_and_groups_for_config3(monkeypatch):
    """The engine's own choice at BASELINE sizes (no MCMCX_GROUP / MCMCX_GROUP_GW): config 2 (Gaussian d = 10, AM, 65536 chains) on the
    quad-group kernel, config 3 (banana d = 20, DRAM, here 32768 of its 262144 chains) on the sixteen-lane one -- and every chain's state,
    stream position and accept ballots equal to the lane-per-chain kernels' after 230 iterations with two adaptations."""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    for wl, n, want in (("c2", 65536, "group_step_kernel<quad>"), ("c3", 32768, "group_step_kernel<DR2>")):
        ckw, pkw, _ = problem(wl, 231, adaptint=100)
        res = []
        for lane in (False, True):
            monkeypatch.delenv("MCMCX_GROUP_GW", raising=False)
            if lane:
                monkeypatch.setenv("MCMCX_GROUP", "0")
            else:
                monkeypatch.delenv("MCMCX_GROUP", raising=False)
            e = engine_from_problem(ckw, pkw, nchains=n, record_accept=1)
            e.init(); e.run()
            if not lane:
                assert e.last_kernel() == want, e.last_kernel()
            res.append((e.theta().copy(), e.accept_masks().copy(), e.scalars().copy(), [e.rng(c) for c in (0, n // 2 + 3, n - 1)], e.totals()))
            e.close()
        a, b = res
        assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and np.array_equal(_bits(a[2]), _bits(b[2])) and a[3] == b[3] and a[4] == b[4], wl
