"""mcmcx_config::scam_fast (opt-in, method = 'scam'): the componentwise proposal formed as oldpar + delta U(:,j) instead
of the reference's U (U'oldpar + delta e_j) (MCMC_run_scam.F90:106-115).  Not the reference's operations, so not bit for
bit its chain: what north_star asks of floating point -- the accept-index sequence identical under the fixed stream, states /
moments within 1e-6 relative -- is what is checked here, against the reference fixtures and against the oracle
(reference-order arithmetic).  The default (scam_fast = 0) stays the bit-exact form covered everywhere else."""
import ctypes as C
import numpy as np
import pytest
from golden_util import load, accepted_from_runlen

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


@pytest.mark.parametrize("name", ["s1_gauss6_scam", "s4_expdata_scam_s2", "s5_banana20_scam"])
@pytest.mark.parametrize("waves", [1, 8, 0])          # 0: the matrix-core form (Gaussian target only; the others run the lane kernels again)
def test_fast_scam_reproduces_the_reference_fixtures(oracle, name, waves, monkeypatch):
    """Per-chain rotations, the reference's SCAM fixtures (real reference, several adaptations each): run-length column and
    stream position identical, the chain's rows and the final covariance / mean within 1e-6 relative."""
    from mcmcf90_amd import engine_from_problem
    if waves:
        monkeypatch.setenv("MCMCX_SCAM_WAVES", str(waves))
        monkeypatch.setenv("MCMCX_SCAM_FAST_LANES", "1")
    z, cfg, prob = load(name, oracle)
    ckw, pkw = _kw(z)
    cid = int(z["chain_id"])
    e = engine_from_problem(ckw, pkw, nchains=70, chain_id0=cid - 1, record_accept=1, record_chain=1, scam_fast=1)
    e.init(); e.run()
    assert e.last_kernel() == ("scam_kernel" if waves == 1 else "scam_mw_kernel<8>" if (waves == 8 or str(pkw["kind"]) != "gauss") else "scam_pooled_kernel<per-chain>"), e.last_kernel()
    np.testing.assert_array_equal(e.accepted(1), accepted_from_runlen(z["runlen"]))
    ch, ss, s2 = e.chain(1)
    np.testing.assert_array_equal(ch[:, -1].astype(np.int32), z["runlen"])
    assert e.rng(1)[0] == int(z["rng_n"])
    k = z["rows_head"].shape[0]
    scale = np.maximum(np.abs(z["rows_tail"]).max(axis=0), 1e-3)
    assert np.max(np.abs(ch[:k, :-1] - z["rows_head"]) / scale) < 1e-6
    assert np.max(np.abs(ch[-k:, :-1] - z["rows_tail"]) / scale) < 1e-6
    cm, mean, _ = e.chaincov(1)
    assert np.max(np.abs(np.triu(cm) - np.triu(z["chaincmat"]))) / np.max(np.abs(z["chaincmat"])) < 1e-6
    assert np.max(np.abs(mean - z["chainmean"])) / max(np.max(np.abs(z["chainmean"])), 1e-300) < 1e-6
    # the other chains against the oracle (reference-order arithmetic): same decisions, same stream position
    for c in (0, 64, 69):
        o = oracle.run_chain(cfg, prob, chain_id=cid - 1 + c)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        assert e.rng(c)[0] == o.rng_n
        chc, _, _ = e.chain(c)
        assert np.max(np.abs(chc[:, :-1] - o.chain[:, :-1]) / np.maximum(np.abs(o.chain[:, :-1]).max(axis=0), 1e-3)) < 1e-6
    e.close()


@pytest.mark.parametrize("lanes", [0, 1], ids=["matrix_cores", "lane_kernels"])
def test_fast_scam_config5_up_to_the_first_adaptation(oracle, lanes, monkeypatch):
    """BASELINE config 5's target (d = 200, cond 1e6), per-chain: 9 iterations x 200 componentwise proposals with the
    initial rotation, against the oracle -- identical decisions and stream position, states to 1e-9 of their scale.  (Past an
    adaptation the d = 200 trajectory is not a function of the inputs alone, DESIGN.md section 8: no form of the
    arithmetic, this one included, can be compared there.)"""
    from mcmcf90_amd import engine_from_problem
    from mcmcf90_amd.workloads import problem
    if lanes:
        monkeypatch.setenv("MCMCX_SCAM_FAST_LANES", "1")
    ckw, pkw, _ = problem("c5", 10, adaptint=10)
    e = engine_from_problem(ckw, pkw, nchains=70, record_accept=1, scam_fast=1)
    e.init(); e.run(9)
    cfg = oracle.make_cfg(**ckw); prob = oracle.Problem(**pkw)
    th = e.theta()
    for c in (0, 63, 69):
        o = oracle.run_chain(cfg, prob, chain_id=c, upto=9)
        np.testing.assert_array_equal(e.accepted(c), o.accepted)
        assert e.rng(c)[0] == o.rng_n
        assert np.max(np.abs(th[c] - o.theta)) < 1e-9 * np.max(np.abs(o.theta))
    e.close()


@pytest.mark.parametrize("d,N", [(30, 200), (130, 150)])
def test_fast_pooled_scam_against_the_reference_order_form(oracle, d, N):
    """Pooled SCAM (one rotation from the pooled covariance, scam_pooled_kernel): the fast form skips the two rotation
    products.  Same accept ballots and stream positions as the reference-order form (itself bit-equal to the restatement,
    tests/test_gpu_pooled.py) through two pooled adaptations, states to 1e-9."""
    from mcmcf90_amd import engine_from_problem
    rng = np.random.default_rng(d)
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    # (d = 130 with 150 chains: the pooled covariance is nearly rank deficient, its rotation not a stable function of the states --
    #  compared up to the first tick only, like config 5 above; d = 30 with 200 chains goes through two ticks)
    ckw = dict(nsimu=26 if d <= 30 else 10, method="scam", adaptint=10, updatesigma=0)
    pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
    res = []
    for fast in (0, 1):
        e = engine_from_problem(ckw, pkw, nchains=N, pooled=1, record_accept=1, scam_fast=fast)
        e.init(); e.run()
        res.append((e.accept_masks(), e.theta(), [e.rng(c)[0] for c in (0, 64, N - 1)], e.pooled()[3]))
        e.close()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    assert res[0][2] == res[1][2]
    assert np.max(np.abs(res[0][1] - res[1][1])) < 1e-9 * np.max(np.abs(res[0][1]))
    if d <= 30:
        assert np.max(np.abs(np.abs(res[0][3]) - np.abs(res[1][3]))) < 1e-6      # the pooled rotation after two ticks (up to column signs)
        assert not np.array_equal(_bits(res[0][1]), _bits(res[1][1]))            # (it IS another arithmetic once the rotation is not the identity)


def test_fast_scam_with_host_callbacks_equals_the_device_target(oracle):
    """The host-callback form of a SCAM sub-step (host_phase_kernel<5>) takes the same fast proposal: bit-equal to the
    device-resident fast run."""
    from mcmcf90_amd import Engine, make_config, engine_from_problem
    z, cfg, prob = load("s1_gauss6_scam", oracle)
    ckw, pkw = _kw(z)
    ckw["nsimu"] = 230
    L = oracle.lib()
    tgt = prob.ctarget()
    L.mcxo_ssfun.restype = C.c_double
    dp = C.POINTER(C.c_double)
    npar = int(pkw["npar"])
    e = Engine(make_config(npar, 3, chain_id0=5, scam_fast=1, **ckw))
    e.setpar0(pkw["par0"]); e.setcmat0(np.asarray(pkw["cmat0"], dtype=float).reshape(npar, npar))
    e.setsigma2nobs(1.0, 1)
    e.set_target_host(lambda th: L.mcxo_ssfun(C.byref(tgt), th.ctypes.data_as(dp)))
    e.init(); e.run()
    r = engine_from_problem(ckw, pkw, nchains=3, chain_id0=5, scam_fast=1)
    r.init(); r.run()
    np.testing.assert_array_equal(_bits(e.theta()), _bits(r.theta()))
    assert [e.rng(c)[0] for c in range(3)] == [r.rng(c)[0] for c in range(3)]
    e.close(); r.close()
