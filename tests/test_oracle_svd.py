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


@pytest.mark.parametrize("seed", range(24))
def test_scam_against_the_mkl_linked_reference_with_its_own_factors(oracle, seed):
    """MCMC_run_scam (MCMC_run_scam.F90:38-138) pinned independently of the Jacobi routine: the reference runs with MKL's
    dgesvd, every call logged (oracle/_ref/mcxref_mkllog); the oracle takes the rotation U and sqrt(s) of each call
    (scam_svd's floor applied, matutils.F90:633-645) in place of its own SVD at the same adaptation.  Everything else
    -- the rotations U'theta / U rot, the componentwise proposals, alpha, the accept decisions, the covariance the next
    adaptation sees -- must then reproduce the MKL-linked chain: identical run-length column and stream position."""
    from oracle import refrun as rr
    if not (rr.available() and os.path.exists(rr.EXE_MKLLOG)):
        pytest.skip("oracle/_ref/mcxref_mkllog not built here")
    r0 = np.random.default_rng(500 + seed)
    d = int(r0.integers(2, 9))
    A = r0.standard_normal((d, d)) / np.sqrt(d)
    S = A @ A.T + np.diag(r0.uniform(0.05, 2.0, d))
    adaptint = int(r0.choice([20, 50]))
    nsimu = int(r0.integers(150, 400))
    updatesigma = int(r0.integers(0, 2))
    kw = dict(nsimu=nsimu, method="scam", adaptint=adaptint, updatesigma=updatesigma)
    pkw = dict(sigma2=0.7, nobs=12) if updatesigma else {}
    prob = oracle.Problem("gauss", d, r0.standard_normal(d) * 0.2, np.diag(r0.uniform(0.05, 0.4, d)), mu=r0.standard_normal(d) * 0.3,
                          lam=np.linalg.inv(S), **pkw)
    cfg = oracle.make_cfg(**kw)
    ref = rr.run_reference(cfg, prob, chain_id=seed, svd_log=True)
    ticks = [it for it in range(adaptint, nsimu + 1, adaptint) if it >= cfg.burnintime + adaptint + cfg.adapthist]
    assert len(ref.svd_calls) == 1 + len(ticks), (len(ref.svd_calls), ticks)

    def factors(call):
        info, sv, U = call
        sv = sv.copy()
        tol = sv[0] / cfg.condmax
        if sv[-1] <= tol:
            sv[sv < tol] = tol
        return U, np.sqrt(sv)

    lc = oracle.LiveChain(cfg, prob, chain_id=seed)
    U, sd = factors(ref.svd_calls[0])
    lc.set_R(U); lc.set_qcovstd(sd)                        # MCMC_init's first MCMC_calculate_R
    for k, it in enumerate(ticks):
        lc.run(it)                                         # the tick at `it` ran the oracle's own SVD: replace its result
        U, sd = factors(ref.svd_calls[1 + k])
        lc.set_R(U); lc.set_qcovstd(sd)
    lc.run(nsimu)
    acc = lc.accepted
    c = lc.ch.contents
    assert c.rng.n == ref.rng_n
    np.testing.assert_array_equal(acc, rr.accepted_from_chain(ref.chain, nsimu))
    ch = np.ctypeslib.as_array(c.chain, shape=(nsimu, d + 1))[:c.chainind]
    scale = np.maximum(np.abs(ref.chain[:, :-1]).max(axis=0), 1e-3)
    assert np.max(np.abs(ch[:, :-1] - ref.chain[:, :-1]) / scale) < 1e-7
    lc.close()
