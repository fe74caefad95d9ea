"""The oracle (oracle/mcx_oracle.c) against the fixtures the REAL Fortran reference produced
(tests/golden/, oracle/gen_golden.py): the accept-index sequence (run-length column of
chain.mat, MCMC_aux.F90:167-175) and the number of uniforms drawn must be identical; the
floating-point chain agrees to the BLAS/libm rounding difference (reference = MKL + glibc)."""
import numpy as np
import pytest
from golden_util import names, load, accepted_from_runlen

RTOL = 1e-7      # relative to the column scale; observed <= 5e-9


@pytest.mark.parametrize("name", names())
def test_oracle_matches_reference_fixture(oracle, name):
    z, cfg, prob = load(name, oracle)
    if "mkl_logged" in z.files:
        pytest.skip("made with MKL's dgesvd: test_c5_fixture_through_two_adaptations_with_the_logged_factors")
    o = oracle.run_chain(cfg, prob, chain_id=int(z["chain_id"]))
    assert o.rc == 0
    assert o.rng_n == int(z["rng_n"])                       # same stream consumption
    assert o.chainind == int(z["chainind"])
    np.testing.assert_array_equal(o.chain[:, -1].astype(np.int32), z["runlen"])      # bit-exact index work
    np.testing.assert_array_equal(o.accepted, accepted_from_runlen(z["runlen"]))
    k = z["rows_head"].shape[0]
    scale = np.maximum(np.abs(z["rows_tail"]).max(axis=0), 1e-3)
    assert np.max(np.abs(o.chain[:k, :-1] - z["rows_head"]) / scale) < RTOL
    assert np.max(np.abs(o.chain[-k:, :-1] - z["rows_tail"]) / scale) < RTOL
    ssh, sst = (o.sschain[:k, :-1], o.sschain[-k:, :-1]) if z["ss_head"].ndim == 2 else (o.sschain[:k, 0], o.sschain[-k:, 0])
    np.testing.assert_allclose(ssh, z["ss_head"], rtol=1e-7, atol=1e-9)        # nycol > 1: every ss column
    np.testing.assert_allclose(sst, z["ss_tail"], rtol=1e-7, atol=1e-9)
    if "s2_head" in z.files:
        ks = z["s2_head"].shape[0]
        np.testing.assert_allclose(o.s2chain[:ks], z["s2_head"], rtol=1e-7)
        np.testing.assert_allclose(o.s2chain[-ks:], z["s2_tail"], rtol=1e-7)
    cs = max(np.max(np.abs(z["chaincmat"])), 1e-300)             # (a chain that never moves -- fixture e8 -- has covariance 0)
    assert np.max(np.abs(o.chaincmat - z["chaincmat"])) / cs < 1e-9
    np.testing.assert_allclose(o.chainmean, z["chainmean"], rtol=1e-9, atol=1e-9 * np.abs(z["chainmean"]).max() + 1e-12)


def test_c5_fixture_through_two_adaptations_with_the_logged_factors(oracle):
    """BASELINE config 5 (d=200 SCAM) from the MKL-linked reference, through the adaptations at iterations 100 and 200.
    The oracle takes the rotation and singular values MKL's dgesvd returned at MCMC_init and at both adaptations in place
    of its own SVD; everything else -- MCMC_run_scam.F90:38-138 (250 x 200 componentwise proposals: rotations, alpha, the
    uniforms drawn) and the covariance MCMC_adapt builds for the next scam_svd -- must reproduce the reference: identical
    run-length column and stream position, every stored row to rounding level."""
    from golden_util import logged_factors
    z, cfg, prob = load("c5_illcond200_scam", oracle)
    assert cfg.adaptint == 100 and len(z["svd_ticks"]) == 2
    lc = oracle.LiveChain(cfg, prob, chain_id=int(z["chain_id"]))
    scale = np.maximum(np.abs(z["rows_tail"]).max(axis=0), 1e-3)
    for k, (it, U, sd) in enumerate(logged_factors(z, cfg)):
        if it > 0:
            lc.run(it)                                     # the tick at `it` ran the oracle's own SVD: replace its result
            assert np.max(np.abs(lc.theta - z["rows_at_ticks"][k - 1]) / scale) < RTOL
        lc.set_R(U); lc.set_qcovstd(sd)
    lc.run(cfg.nsimu)
    c = lc.ch.contents
    assert c.rng.n == int(z["rng_n"])
    assert c.chainind == int(z["chainind"])
    n = prob.npar
    ch = np.ctypeslib.as_array(c.chain, shape=(cfg.nsimu, n + 1))[:c.chainind]
    np.testing.assert_array_equal(ch[:, -1].astype(np.int32), z["runlen"])
    k = z["rows_head"].shape[0]
    assert np.max(np.abs(ch[:k, :-1] - z["rows_head"]) / scale) < RTOL
    assert np.max(np.abs(ch[-k:, :-1] - z["rows_tail"]) / scale) < RTOL
    cm = np.ctypeslib.as_array(c.chaincmat, shape=(n, n)).T
    assert np.max(np.abs(cm - z["chaincmat"])) / np.max(np.abs(z["chaincmat"])) < 1e-9
    mean = np.ctypeslib.as_array(c.chainmean, shape=(n,))
    np.testing.assert_allclose(mean, z["chainmean"], rtol=1e-9, atol=1e-9 * np.abs(z["chainmean"]).max() + 1e-12)
    lc.close()


def test_incremental_run_equals_one_shot(oracle):
    z, cfg, prob = load("c3_banana20_dram", oracle)
    full = oracle.run_chain(cfg, prob, chain_id=3)
    part = oracle.run_chain(cfg, prob, chain_id=3, upto=777)
    np.testing.assert_array_equal(full.accepted[:777], part.accepted)
    assert part.simuind == 777
