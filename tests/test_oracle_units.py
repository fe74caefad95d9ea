"""Known-answer and property tests of the oracle's numerical pieces."""
import ctypes as C
import numpy as np
import pytest


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    import subprocess, tempfile, os
    src = r'''
#include <stdio.h>
#include "mcx_rng.h"
int main(void){
  uint32_t o[4];
  uint32_t c0[4]={0,0,0,0},k0[2]={0,0}; mcxo_philox4x32_10(c0,k0,o); printf("%08x %08x %08x %08x\n",o[0],o[1],o[2],o[3]);
  uint32_t c1[4]={0xffffffffu,0xffffffffu,0xffffffffu,0xffffffffu},k1[2]={0xffffffffu,0xffffffffu}; mcxo_philox4x32_10(c1,k1,o); printf("%08x %08x %08x %08x\n",o[0],o[1],o[2],o[3]);
  uint32_t c2[4]={0x243f6a88u,0x85a308d3u,0x13198a2eu,0x03707344u},k2[2]={0xa4093822u,0x299f31d0u}; mcxo_philox4x32_10(c2,k2,o); printf("%08x %08x %08x %08x\n",o[0],o[1],o[2],o[3]);
  return 0; }'''
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "k.c"), "w").write(src)
        subprocess.check_call(["gcc", "-O1", "-I", here, os.path.join(d, "k.c"), "-o", os.path.join(d, "k")])
        out = subprocess.check_output([os.path.join(d, "k")]).decode().split("\n")
    assert out[0] == "6627e8d5 e169c58d bc57ac4c 9b00dbd8"
    assert out[1] == "408f276d 41c83b0e a20bc7c6 6d5451fd"
    assert out[2] == "d16cfe09 94fdcceb 5001e420 24126ea1"


def _ulps(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_log_exp_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(20000), 10.0 ** rng.uniform(-300, 300, 5000), 1 + rng.uniform(-1e-6, 1e-6, 2000),
                         [1.0, 0.5, 2.0, 5e-324, 2.2250738585072014e-308]])
    got = np.array([L.mcxo_log(float(x)) for x in xs])
    ref = np.log(xs)
    nz = ref != 0
    assert np.max(_ulps(got[nz], ref[nz])) <= 1.0
    assert np.all(got[~nz] == 0)
    assert L.mcxo_log(0.0) == -np.inf and np.isnan(L.mcxo_log(-1.0)) and L.mcxo_log(np.inf) == np.inf
    ts = np.concatenate([-rng.random(20000) * 708.0, rng.uniform(-745, 709, 5000), rng.uniform(-1e-3, 1e-3, 2000), [0.0]])
    got = np.array([L.mcxo_exp(float(t)) for t in ts])
    ref = np.exp(ts)
    normal = ref > 1e-300
    assert np.max(_ulps(got[normal], ref[normal])) <= 1.0
    assert L.mcxo_exp(-800.0) == 0.0 and L.mcxo_exp(800.0) == np.inf and L.mcxo_exp(0.0) == 1.0


def test_alpha_clamps(oracle):
    """MCMC_alpha, MCMC_DRAM.F90:108-116."""
    L = oracle.lib()
    assert L.mcxo_alpha(10.0, 0.0, 9.0, 0.0, 1.0) == 1.0
    assert L.mcxo_alpha(10.0, 0.0, 10.0, 0.0, 1.0) == 1.0
    assert L.mcxo_alpha(0.0, 0.0, 1500.0, 0.0, 1.0) == 0.0            # tst < log_realmin
    a = L.mcxo_alpha(0.0, 0.0, 2.0, 1.0, 2.0)
    assert abs(a - np.exp(-0.5 * (1.0 + 1.0))) < 1e-15


def test_normal_pair_caching(oracle):
    """normal_bm caches the second deviate (mcmcrand.F90:172-189): an odd number of normals leaves one saved."""
    L = oracle.lib()
    g = oracle.Rng(); g.key[0] = 1; g.key[1] = 2
    z = [L.mcxo_normal(C.byref(g)) for _ in range(3)]
    assert g.saved == 1 and g.n % 2 == 0
    n_before = g.n
    z4 = L.mcxo_normal(C.byref(g))
    assert g.n == n_before and g.saved == 0 and z4 == g.saved_y
    zs = np.array([L.mcxo_normal(C.byref(g)) for _ in range(20000)])
    assert abs(zs.mean()) < 0.03 and abs(zs.std() - 1) < 0.03


def test_gamma_moments(oracle):
    L = oracle.lib()
    g = oracle.Rng(); g.key[0] = 3; g.key[1] = 4
    x = np.array([L.mcxo_gamma(C.byref(g), 6.0, 0.5) for _ in range(20000)])
    assert abs(x.mean() - 3.0) < 0.05 and abs(x.var() - 1.5) < 0.08


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [1, 2, 5, 20, 50])
def test_cholesky_and_inverse(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n)
    U = np.asfortranarray(S.copy())
    assert L.mcxo_potrf_u(n, _dp(U)) == 0
    Ut = np.triu(U)
    np.testing.assert_allclose(Ut.T @ Ut, S, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(np.tril(U, -1), np.tril(S, -1))       # lower triangle untouched, like dpotrf
    assert L.mcxo_potri_u(n, _dp(U)) == 0
    inv = np.triu(U) + np.triu(U, 1).T
    np.testing.assert_allclose(inv @ S, np.eye(n), atol=1e-10)
    bad = np.asfortranarray(np.diag(np.r_[np.ones(n - 1), -1.0]) if n > 1 else -np.ones((1, 1)))
    assert L.mcxo_potrf_u(n, _dp(bad)) == n


@pytest.mark.parametrize("n", [1, 3, 10, 50])
def test_trmv_chud_chdd(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n)
    R = np.asfortranarray(np.linalg.cholesky(S).T.copy())
    z = rng.standard_normal(n); x = z.copy()
    L.mcxo_trmv_ut(n, _dp(R), _dp(x))
    np.testing.assert_allclose(x, R.T @ z, rtol=1e-13, atol=1e-13)
    xd = z.copy()                                                    # the post-downdate order: diagonal first, rows descending
    L.mcxo_trmv_ut_desc(n, _dp(R), _dp(xd))
    np.testing.assert_allclose(xd, R.T @ z, rtol=1e-13, atol=1e-13)
    ref = np.array([np.sum((R[:j + 1, j] * z[:j + 1])[::-1].cumsum()[-1:]) for j in range(n)])
    np.testing.assert_allclose(xd, ref, rtol=1e-13, atol=1e-13)
    v = rng.standard_normal(n) * 0.3
    c = np.zeros(n); s = np.zeros(n)
    R1 = R.copy(order="F")
    L.mcxo_chud(n, _dp(R1), _dp(v), _dp(c), _dp(s))
    R1u = np.triu(R1)
    np.testing.assert_allclose(R1u.T @ R1u, S + np.outer(v, v), rtol=1e-12, atol=1e-12)
    R2 = R1.copy(order="F")
    assert L.mcxo_chdd(n, _dp(R2), _dp(v), _dp(c), _dp(s)) == 0
    R2u = np.triu(R2)
    np.testing.assert_allclose(R2u.T @ R2u, S, rtol=1e-11, atol=1e-11)
    big = (R.T @ np.ones(n)) * 1.5                                   # ||R^-T x|| >= 1 -> INFO = -1, R untouched
    R3 = R.copy(order="F")
    assert L.mcxo_chdd(n, _dp(R3), _dp(big), _dp(c), _dp(s)) == -1
    np.testing.assert_array_equal(R3, R)


def test_covmat_update_matches_batch(oracle):
    """Weighted Welford update (matutils.F90:283-310) == two-pass batch (:311-338) on the expanded sample."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    n, p = 40, 6
    x = np.ascontiguousarray(rng.standard_normal((n, p)))
    w = rng.integers(1, 5, n).astype(np.float64)
    cm = np.zeros((p, p), order="F"); mean = np.zeros(p); ws = C.c_double(0.0)
    L.mcxo_covmat(10, p, _dp(x), p, _dp(w), 10, _dp(cm), _dp(mean), C.byref(ws), 1)      # wsum == 0 -> batch branch
    assert ws.value == w[:10].sum()
    L.mcxo_covmat(n - 10, p, _dp(x[10:]), p, _dp(w[10:]), n - 10, _dp(cm), _dp(mean), C.byref(ws), 1)
    xe = np.repeat(x, w.astype(int), axis=0)
    np.testing.assert_allclose(cm, np.cov(xe.T), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(mean, xe.mean(axis=0), rtol=1e-12, atol=1e-13)
    assert ws.value == w.sum()
    np.testing.assert_array_equal(cm, cm.T)


def test_cfg_check_derivations(oracle):
    """mcmcinit.F90:235-368."""
    c = oracle.make_cfg(nsimu=10, method="ram", drscale=3.0)
    assert c.drscale == 0.0 and c.dodr == 0
    c = oracle.make_cfg(nsimu=10, method="scam", doburnin=1)
    assert c.doscam == 1 and c.doburnin == 0 and c.condmax == 1e15 and c.usesvd == 1
    c = oracle.make_cfg(nsimu=10, badaptint=-1, adaptint=77, drscale=2.0)
    assert c.badaptint == 77 and c.dodr == 1 and c.usesvd == 0
    with pytest.raises(ValueError):
        oracle.make_cfg(nsimu=10, scalelimit=0.7)
