"""The oracle against the REAL reference on randomised configurations (dev container only: needs oracle/_ref/mcxref,
built from /root/reference; skipped on the GPU box).  Same generator as tests/test_gpu_fuzz.py, so the option space the
device is checked on against the oracle is the one the oracle is checked on against mcmcf90 itself: identical
run-length column (accept sequence), identical number of uniforms drawn, states to BLAS/libm rounding."""
import importlib.util
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("gpu_fuzz_gen", os.path.join(HERE, "test_gpu_fuzz.py"))
_gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gen)


def _scam_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed):
    """A SCAM configuration whose covariance is ill posed for ANY dgesvd (clustered or vanishing singular values: the
    rotation is rounding noise) is not set aside any more: the reference runs with MKL's dgesvd, every call logged
    (oracle/_ref/mcxref_mkllog), and the oracle takes the logged rotation and singular values at the same adaptations in
    place of its own SVD (like tests/test_oracle_svd.py and the config-5 fixture).  Everything else -- MCMC_run_scam.F90:38-138,
    the covariance MCMC_adapt builds -- must reproduce the MKL-linked chain: run-length column, stream position, states.
    Returns False when the case cannot be checked this way (a dgesvd call reported info != 0, or the tick count differs)."""
    ref = rr.run_reference(cfg, prob, chain_id=seed, svd_log=True)
    ticks = [it for it in range(cfg.adaptint, cfg.nsimu + 1, cfg.adaptint)
             if cfg.doadapt and it >= cfg.burnintime + cfg.adaptint + cfg.adapthist and not (cfg.adaptend > 0 and it > cfg.adaptend)]
    if len(ref.svd_calls) != 1 + len(ticks) or any(info != 0 or not np.isfinite(sv).all() or not np.isfinite(U).all() for info, sv, U in ref.svd_calls):
        return False

    def factors(call):
        info, sv, U = call
        sv = sv.copy()
        tol = sv[0] / cfg.condmax
        if sv[-1] <= tol:
            sv[sv < tol] = tol
        return U, np.sqrt(sv)

    lc = oracle.LiveChain(cfg, prob, chain_id=seed)
    U, sd = factors(ref.svd_calls[0])
    lc.set_R(U); lc.set_qcovstd(sd)
    for k, it in enumerate(ticks):
        lc.run(it)
        U, sd = factors(ref.svd_calls[1 + k])
        lc.set_R(U); lc.set_qcovstd(sd)
    lc.run(cfg.nsimu)
    c = lc.ch.contents
    n = prob.npar
    ch = np.ctypeslib.as_array(c.chain, shape=(cfg.nsimu, n + 1))[:c.chainind].copy()
    rng_n = c.rng.n
    lc.close()
    np.testing.assert_array_equal(ref.chain[:, -1].astype(np.int64), ch[:, -1].astype(np.int64))
    assert ref.rng_n == rng_n
    scale = np.maximum(np.abs(ch[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(ref.chain[:, :-1] - ch[:, :-1]) / scale) < 1e-7
    return True


def _svd_dram_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed):
    """The same for method = 'dram' with condmax > 0 (covtor_svd, matutils.F90:378-453): at every MCMC_calculate_R of the
    oracle (MCMC_init, the AM ticks, the burn-in ticks that fall through to it -- the oracle counts its calls) the factor is
    rebuilt from what MKL's dgesvd returned to the reference there: singular values floored at s_1 / condmax,
    R = U sqrt(s) 2.4/sqrt(npar), and with delayed rejection R2 = R / drscale, iC = dpotri('U', R) (MCMC_adapt.F90:204-225).
    Where the floor bites the covariance itself becomes U s U' -- a function of the covariance that does not depend on the
    basis chosen inside a cluster -- so the oracle's own replacement stands as long as both agree that it bit.
    Returns False when the case cannot be checked this way."""
    import ctypes as C
    ref = rr.run_reference(cfg, prob, chain_id=seed, svd_log=True)
    calls = ref.svd_calls
    if not calls or any(info != 0 or not np.isfinite(sv).all() or not np.isfinite(U).all() or not sv[0] > 0.0 for info, sv, U in calls):
        return False
    n = prob.npar
    DP = C.POINTER(C.c_double)
    lc = oracle.LiveChain(cfg, prob, chain_id=seed)

    def put(call):
        info, sv, U = call
        sv = sv.copy()
        tol = sv[0] / cfg.condmax
        floored = bool(sv[-1] <= tol)
        if floored:
            sv[sv < tol] = tol
        if floored != bool(lc.ch.contents.svd_floored):
            return False
        R = (U * np.sqrt(sv)) * 2.4 / np.sqrt(float(n))           # R(i,j) = U(i,j) sqrt(s_j) 2.4/sqrt(n)
        lc.set_R(R)
        if cfg.dodr:
            B = np.asfortranarray(R.copy())
            if oracle.lib().mcxo_potri_u(n, B.ctypes.data_as(DP)) != 0:
                return False
            lc.set_dr(R / cfg.drscale, np.array(B))
        return True

    k = 0
    ok = lc.ch.contents.n_calcR == 1 and put(calls[0])
    step = max(1, min(cfg.adaptint, cfg.badaptint if cfg.badaptint > 0 else cfg.adaptint))
    it = step
    while ok and it <= cfg.nsimu:
        before = lc.ch.contents.n_calcR
        lc.run(it)
        if lc.ch.contents.n_calcR != before:
            k += 1
            ok = lc.ch.contents.n_calcR == before + 1 and k < len(calls) and put(calls[k])
        it += 1 if step == 1 else (step - it % step if it % step else step)
    if ok:
        lc.run(cfg.nsimu)
        ok = (k + 1 == len(calls))
    if not ok:
        lc.close()
        return False
    c = lc.ch.contents
    ch = np.ctypeslib.as_array(c.chain, shape=(cfg.nsimu, n + 1))[:c.chainind].copy()
    rng_n = c.rng.n
    lc.close()
    np.testing.assert_array_equal(ref.chain[:, -1].astype(np.int64), ch[:, -1].astype(np.int64))
    assert ref.rng_n == rng_n
    scale = np.maximum(np.abs(ch[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(ref.chain[:, :-1] - ch[:, :-1]) / scale) < 1e-7
    return True


def _first_ill_posed_tick(oracle, cfg, prob, seed):
    """The first iteration at which MCMC_adapt may hand LAPACK a covariance it cannot be expected to agree on with itself
    (None: there is none) -- the condition number of chaincmat at every iteration where MCMC_adapt may factor it."""
    n = prob.npar
    lc = oracle.LiveChain(cfg, prob, chain_id=seed)
    bad = None
    step = max(1, min(cfg.adaptint, cfg.badaptint if cfg.badaptint > 0 else cfg.adaptint))
    for it in range(step, cfg.nsimu + 1, step):
        lc.run(it)
        cm = np.ctypeslib.as_array(lc.ch.contents.chaincmat, shape=(n, n)).copy()
        cm = np.triu(cm) + np.triu(cm, 1).T
        if not np.isfinite(cm).all():                          # e.g. a one-row window: covmat divides by wsum - 1 = 0
            bad = it
            break
        ev = np.linalg.eigvalsh(cm)
        # Cholesky paths: only what an FP64 dpotrf cannot be expected to agree on with itself (cond > 1e10) is set
        # aside -- cond 1e6 (BASELINE config 5's regime) is well inside what must reproduce.  The SVD paths keep the
        # wider margin: there the singular VECTORS matter, and they turn by O(eps * cond).
        if ev[-1] <= 0 or ev[0] < (1e-6 if cfg.usesvd else 1e-10) * ev[-1]:
            bad = it
            break
        # the SVD paths also need distinct singular values: inside a cluster the basis is arbitrary
        if cfg.usesvd and n > 1 and np.min(np.diff(ev)) < 1e-6 * ev[-1]:
            bad = it
            break
    lc.close()
    return bad


def _well_posed(oracle, cfg, prob, seed):
    return _first_ill_posed_tick(oracle, cfg, prob, seed) is None


def _agree(r, o, cfg, ckw):
    """run-length column, stream position, states, sigma2 chain of a reference run and an oracle run"""
    np.testing.assert_array_equal(r.chain[:, -1].astype(np.int64), o.chain[:, -1].astype(np.int64), err_msg=str(ckw))
    assert r.rng_n == o.rng_n, ckw
    scale = np.maximum(np.abs(o.chain[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(r.chain[:, :-1] - o.chain[:, :-1]) / scale) < 1e-7, ckw
    assert r.sschain.shape == o.sschain.shape
    np.testing.assert_allclose(r.sschain[:, :-1], o.sschain[:, :-1], rtol=1e-7, atol=1e-300)
    if cfg.updatesigma:
        assert r.s2chain.shape == o.s2chain.shape
        np.testing.assert_allclose(r.s2chain, o.s2chain, rtol=1e-7)


def _up_to_the_ill_posed_tick(oracle, rr, ckw, cfg, prob, seed, o, bad):
    """A configuration whose covariance becomes numerically singular (or, on the SVD paths, clustered) at the adaptation of iteration
    `bad` is not set aside.  What LAPACK makes of that matrix is rounding noise -- but everything UP TO the
    call is not, and neither is the set of things the reference can do with the answer (MCMC_adapt.F90:167-170, 204-225):
      (1) the reference's own reaction on the full run is asserted: it completes (dpotrf / dgesvd reported failure -> 'not adapting',
          old factor kept; or reported success on noise) or it stops in dpotri ('cannot invert cmat') -- nothing else;
      (2) if it completes and its chain IS the oracle's (both kept the old factor, or the noise did not reach an accept decision), the
          whole run is compared like a well-posed one;
      (3) otherwise both sides are cut at nsimu = bad -- the tick is then the run's last act, its factor proposes nothing -- and the
          run-length column, the stream position, the states AND the covariance handed to the factorisation (mcmccovf.dat against the
          oracle's chaincmat: same rows, same Welford folds) must agree.
    Returns 'full' or 'prefix'."""
    pinned = bool(cfg.usesvd)
    completed = None
    try:
        completed = rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=pinned)
    except RuntimeError as ex:                                # the program stopped: only dpotri's stop is a legitimate one here
        assert "cannot invert cmat" in str(ex), "the reference stopped for another reason:\n" + str(ex)[-1500:]
        assert cfg.dodr, "dpotri is only reached with delayed rejection"
    if completed is not None:
        try:
            _agree(completed, o, cfg, ckw)
            return "full"
        except AssertionError:
            pass
    ckw2 = dict(ckw, nsimu=int(bad))
    cfg2 = oracle.make_cfg(**ckw2)
    o2 = oracle.run_chain(cfg2, prob, chain_id=seed)
    assert not o2.ram_downdate_fail
    try:
        r2 = rr.run_reference(cfg2, prob, chain_id=seed, pinned_svd=pinned)
    except RuntimeError as ex:
        # the stop in dpotri comes before the chain files are written: the tick's own iteration cannot be looked at -- one iteration less can
        assert "cannot invert cmat" in str(ex) and cfg.dodr, str(ex)[-1500:]
        cfg2 = oracle.make_cfg(**dict(ckw, nsimu=int(bad) - 1))
        o2 = oracle.run_chain(cfg2, prob, chain_id=seed)
        r2 = rr.run_reference(cfg2, prob, chain_id=seed, pinned_svd=pinned)
        _agree(r2, o2, cfg2, ckw)
        return "prefix"
    _agree(r2, o2, cfg2, ckw2)
    # the matrix handed to LAPACK at the tick (upper triangle: covmat fills both, the oracle's record keeps the upper one)
    cm_o = np.triu(o2.chaincmat) + np.triu(o2.chaincmat, 1).T
    cm_r = np.triu(r2.chaincmat) + np.triu(r2.chaincmat, 1).T
    if np.isfinite(cm_o).all():
        sc = max(np.abs(cm_o).max(), 1e-300)
        # (with condmax > 0 covtor_svd may have replaced the covariance by U s U' -- the floor bit -- on either side: noise again, not compared)
        if not cfg.usesvd:
            assert np.max(np.abs(cm_r - cm_o)) <= 1e-6 * sc, ckw2
    else:
        assert not np.isfinite(cm_r).all(), "the oracle's covariance is not finite, the reference's is"
    return "prefix"


@pytest.mark.parametrize("seed", range(400))
def test_oracle_equals_reference_on_random_configuration(oracle, seed):
    from oracle import refrun as rr
    if not rr.available():
        pytest.skip("oracle/_ref/mcxref not built (needs /root/reference)")
    ckw, pkw = _gen._draw(seed)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    try:
        o = oracle.run_chain(cfg, prob, chain_id=seed)
    except RuntimeError:
        with pytest.raises(Exception):                        # the reference stops in MCMC_init too
            rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=bool(cfg.usesvd))
        return
    if o.ram_downdate_fail:
        cfg, o = _prefix_before_failed_downdate(oracle, rr, ckw, cfg, prob, seed, o)
        if cfg is None:
            return
    if cfg.method != 1 and not _well_posed(oracle, cfg, prob, seed):
        if cfg.doscam and os.path.exists(rr.EXE_MKLLOG) and _scam_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed):
            return
        if cfg.usesvd and not cfg.doscam and cfg.method == 0 and os.path.exists(rr.EXE_MKLLOG) and \
                _svd_dram_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed):
            return
        _up_to_the_ill_posed_tick(oracle, rr, ckw, cfg, prob, seed, o, _first_ill_posed_tick(oracle, cfg, prob, seed))
        return
    r = rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=bool(cfg.usesvd))
    np.testing.assert_array_equal(r.chain[:, -1].astype(np.int64), o.chain[:, -1].astype(np.int64), err_msg=str(ckw))
    assert r.rng_n == o.rng_n, ckw
    scale = np.maximum(np.abs(o.chain[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(r.chain[:, :-1] - o.chain[:, :-1]) / scale) < 1e-7, ckw
    if cfg.updatesigma:
        np.testing.assert_allclose(r.s2chain, o.s2chain, rtol=1e-7)
    # the SVD paths a second time, independently of the pinned routine: against the MKL-linked chain with its logged factors
    if cfg.usesvd and cfg.method in (0, 2) and os.path.exists(rr.EXE_MKLLOG):
        if cfg.doscam:
            _scam_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed)
        else:
            _svd_dram_against_the_mkl_chain_with_its_logged_factors(oracle, rr, cfg, prob, seed)


def _prefix_before_failed_downdate(oracle, rr, ckw, cfg, prob, seed, o, pinned_svd=False):
    """A Cholesky downdate with INFO = -1 stops the reference (matutils.F90:719-722) at iteration k = what the oracle
    reports.  Check that it does stop there, then hand back the run cut at k - 1 so that the caller compares the
    whole prefix -- the iterations that decide whether the failing downdate is reached at all."""
    k = int(o.ram_downdate_fail)
    with pytest.raises(RuntimeError):                         # no chain file: the program stopped inside choldowndate
        rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=pinned_svd)
    if k - 1 < 2:
        return None, None
    ckw2 = dict(ckw, nsimu=k - 1)
    cfg2 = oracle.make_cfg(**ckw2)
    o2 = oracle.run_chain(cfg2, prob, chain_id=seed)
    assert not o2.ram_downdate_fail
    try:                                                      # the reference with nsimu = k must fail too: the failure is AT k
        rr.run_reference(oracle.make_cfg(**dict(ckw, nsimu=k)), prob, chain_id=seed, pinned_svd=pinned_svd)
        raise AssertionError("the reference survives iteration %d where the oracle's downdate fails: %s" % (k, ckw))
    except RuntimeError:
        pass
    return cfg2, o2


@pytest.mark.parametrize("seed", range(120))
def test_oracle_equals_reference_with_response_columns(oracle, seed):
    """nycol = 2 or 3 (vector-valued ssfunction, one sigma2 per column): the oracle against the real reference."""
    from oracle import refrun as rr
    if not rr.available():
        pytest.skip("oracle/_ref/mcxref not built (needs /root/reference)")
    ckw, pkw = _gen._draw_cols(seed)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    o = oracle.run_chain(cfg, prob, chain_id=seed)
    if o.ram_downdate_fail:                                   # the reference stops there (matutils.F90:717-722): the run up to the stop is compared
        cfg, o = _prefix_before_failed_downdate(oracle, rr, ckw, cfg, prob, seed, o)
        if cfg is None:
            return
    if cfg.method != 1:
        bad = _first_ill_posed_tick(oracle, cfg, prob, seed)
        if bad is not None:
            _up_to_the_ill_posed_tick(oracle, rr, ckw, cfg, prob, seed, o, bad)
            return
    r = rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=bool(cfg.usesvd))
    np.testing.assert_array_equal(r.chain[:, -1].astype(np.int64), o.chain[:, -1].astype(np.int64), err_msg=str(ckw))
    assert r.rng_n == o.rng_n, ckw
    assert r.sschain.shape == o.sschain.shape
    np.testing.assert_allclose(r.sschain[:, :-1], o.sschain[:, :-1], rtol=1e-7)
    if cfg.updatesigma:
        assert r.s2chain.shape == o.s2chain.shape
        np.testing.assert_allclose(r.s2chain, o.s2chain, rtol=1e-7)


@pytest.mark.parametrize("seed", range(6))
def test_oracle_equals_reference_with_gamma_shape_below_one(oracle, seed):
    """N0/2 + nobs/2 < 1: random_gamma's u**(1/a) branch (mcmcrand.F90:102-105), which the reference warns about and the
    device engine refuses; the oracle still has to consume the stream the way the reference does."""
    from oracle import refrun as rr
    if not rr.available():
        pytest.skip("oracle/_ref/mcxref not built (needs /root/reference)")
    x = np.linspace(0, 5, 9)
    y = 2 * np.exp(-0.5 * x)
    cfg = oracle.make_cfg(nsimu=300, method="dram" if seed % 2 else "scam", adaptint=50, updatesigma=1, N0=0.5, S02=0.8)
    prob = oracle.Problem("expdata", 2, par0=np.array([2.0, 0.5]), cmat0=np.diag([0.01, 0.01]), sigma2=0.5, nobs=1,
                          xdata=x, ydata=y)
    o = oracle.run_chain(cfg, prob, chain_id=seed)
    r = rr.run_reference(cfg, prob, chain_id=seed)
    np.testing.assert_array_equal(r.chain[:, -1].astype(np.int64), o.chain[:, -1].astype(np.int64))
    assert r.rng_n == o.rng_n
    np.testing.assert_allclose(r.s2chain, o.s2chain, rtol=1e-12)


@pytest.mark.parametrize("seed", range(120))
def test_oracle_equals_reference_ram_with_svd_factor(oracle, seed):
    """method='ram' with condmax > 0 (MCMC_run_ram.F90:96-97, 168-172: the full SVD factor goes through matmulx and
    dchud / dchdd as it is).  Most draws end in a failed downdate, where the reference stops; the rest must agree."""
    from oracle import refrun as rr
    if not rr.available():
        pytest.skip("oracle/_ref/mcxref not built (needs /root/reference)")
    ckw, pkw = _gen._draw_ram_svd(seed)
    cfg = oracle.make_cfg(**ckw)
    prob = oracle.Problem(**pkw)
    try:
        o = oracle.run_chain(cfg, prob, chain_id=seed)
    except RuntimeError:
        with pytest.raises(Exception):
            rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=True)
        return
    if o.ram_downdate_fail:
        cfg, o = _prefix_before_failed_downdate(oracle, rr, ckw, cfg, prob, seed, o, pinned_svd=True)
        if cfg is None:
            return
    r = rr.run_reference(cfg, prob, chain_id=seed, pinned_svd=True)
    np.testing.assert_array_equal(r.chain[:, -1].astype(np.int64), o.chain[:, -1].astype(np.int64), err_msg=str(ckw))
    assert r.rng_n == o.rng_n, ckw
    scale = np.maximum(np.abs(o.chain[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(r.chain[:, :-1] - o.chain[:, :-1]) / scale) < 1e-7, ckw
    if cfg.updatesigma:
        np.testing.assert_allclose(r.s2chain, o.s2chain, rtol=1e-7)
