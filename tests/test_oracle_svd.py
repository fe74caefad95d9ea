"""The pinned SVD (oracle/mcx_svd.h, what stands in for the unpinned LAPACK dgesvd of matutils.F90:409,615) and
the SCAM path's independence of the LAPACK one links, statistically."""
import ctypes as C
import os
import numpy as np
import pytest


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("n", [1, 2, 5, 20, 50])
def test_symsvd_known_answers(oracle, n):
    L = oracle.lib()
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    sv = np.sort(10.0 ** rng.uniform(-6, 1, n))[::-1]
    A = (Q * sv) @ Q.T
    A = (A + A.T) / 2
    G = np.asfortranarray(A.copy()); V = np.zeros((n, n), order="F"); s = np.zeros(n)
    sweeps = L.mcxo_symsvd(n, _dp(G), _dp(V), _dp(s))
    assert sweeps < 60
    assert np.all(np.diff(s) <= 0)
    np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), rtol=1e-9, atol=1e-13 * sv[0])
    np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-13)
    np.testing.assert_allclose((V * s) @ V.T, A, atol=1e-12 * sv[0])


def test_symsvd_diagonal_and_repeated(oracle):
    L = oracle.lib()
    A = np.diag([3.0, 3.0, 1.0, 0.0])
    G = np.asfortranarray(A.copy()); V = np.zeros((4, 4), order="F"); s = np.zeros(4)
    L.mcxo_symsvd(4, _dp(G), _dp(V), _dp(s))
    np.testing.assert_array_equal(s, [3.0, 3.0, 1.0, 0.0])
    np.testing.assert_array_equal(V, np.eye(4))


def test_gemv_orders(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    n = 7
    A = np.asfortranarray(rng.standard_normal((n, n))); x = rng.standard_normal(n); y = np.zeros(n)
    L.mcxo_gemv(0, n, _dp(A), _dp(x), _dp(y))
    np.testing.assert_allclose(y, A @ x, rtol=1e-13)
    L.mcxo_gemv(1, n, _dp(A), _dp(x), _dp(y))
    np.testing.assert_allclose(y, A.T @ x, rtol=1e-13)


def test_scam_posterior_does_not_depend_on_the_lapack_linked(oracle):
    """The real reference linked with MKL's dgesvd and with the pinned routine: possibly different singular-vector
    signs, hence different realised chains, but the same posterior.  (Dev container only: needs oracle/_ref.)"""
    from oracle import refrun as rr
    if not (rr.available() and os.path.exists(rr.EXE_SVD)):
        pytest.skip("oracle/_ref not built here")
    d = 4
    S = np.array([[4.0, 1.2, 0.0, 0.3], [1.2, 1.0, 0.2, 0.0], [0.0, 0.2, 0.25, 0.05], [0.3, 0.0, 0.05, 2.0]])
    prob = oracle.Problem("gauss", d, np.zeros(d), 0.1 * np.eye(d), mu=np.array([1.0, -1.0, 0.5, 0.0]), lam=np.linalg.inv(S))
    cfg = oracle.make_cfg(nsimu=40000, method="scam", adaptint=200, updatesigma=0)
    out = []
    for pinned in (False, True):
        r = rr.run_reference(cfg, prob, chain_id=77, pinned_svd=pinned)
        w = r.chain[:, -1]; x = r.chain[:, :-1]
        keep = np.cumsum(w) > 4000
        m = np.average(x[keep], axis=0, weights=w[keep])
        c = np.cov(x[keep].T, fweights=w[keep].astype(int))
        out.append((m, c, x[600].copy()))
    # (with this matrix MKL and the Jacobi routine disagree on singular-vector signs: the two chains part ways at the
    #  first adaptation; they need not on another LAPACK)
    for m, c, _ in out:
        np.testing.assert_allclose(m, prob.mu, atol=0.12)
        np.testing.assert_allclose(c, S, atol=0.45)
