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
    cs = np.max(np.abs(z["chaincmat"]))
    assert np.max(np.abs(o.chaincmat - z["chaincmat"])) / cs < 1e-9
    np.testing.assert_allclose(o.chainmean, z["chainmean"], rtol=1e-9, atol=1e-9 * np.abs(z["chainmean"]).max() + 1e-12)


def test_incremental_run_equals_one_shot(oracle):
    z, cfg, prob = load("c3_banana20_dram", oracle)
    full = oracle.run_chain(cfg, prob, chain_id=3)
    part = oracle.run_chain(cfg, prob, chain_id=3, upto=777)
    np.testing.assert_array_equal(full.accepted[:777], part.accepted)
    assert part.simuind == 777
