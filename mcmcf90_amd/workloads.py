"""Synthetic problems of BASELINE.json's configurations: one definition for bench.py and the tests."""
import numpy as np


def corr_gauss_precision(d, rho=0.5):
    """C4: Sigma_ij = rho^|i-j|, Lambda = inv(Sigma) (tridiagonal, stored dense)."""
    S = rho ** np.abs(np.subtract.outer(np.arange(d), np.arange(d)))
    return np.linalg.inv(S)


def illcond_gauss_precision(d=200, decades=6.0, seed=5, nreflect=8):
    """C5: Lambda = Q diag(10^linspace(0, decades, d)) Q', Q a fixed-seed product of Householder reflections."""
    rng = np.random.default_rng(seed)
    Q = np.eye(d)
    for _ in range(nreflect):
        v = rng.standard_normal(d)
        v /= np.linalg.norm(v)
        Q = Q - 2.0 * np.outer(Q @ v, v)
    L = (Q * (10.0 ** np.linspace(0.0, decades, d))) @ Q.T
    return 0.5 * (L + L.T)


def problem(name, nsimu, adaptint=100):
    """(cfg kwargs, problem kwargs, proposals per iteration and chain excluding DR retries) of configuration c2..c5."""
    if name == "c2":
        d = 10
        return (dict(nsimu=nsimu, adaptint=adaptint, updatesigma=0),
                dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=np.eye(d)), 1)
    if name == "c3":
        d = 20
        return (dict(nsimu=nsimu, adaptint=adaptint, updatesigma=0, drscale=2.0),
                dict(kind="banana", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), b=0.1), 1)
    if name == "c4":
        d = 50
        return (dict(nsimu=nsimu, method="ram", updatesigma=0, adaptint=adaptint),
                dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=0.01 * np.eye(d), mu=np.zeros(d), lam=corr_gauss_precision(d)), 1)
    if name == "c5":
        d = 200
        return (dict(nsimu=nsimu, method="scam", updatesigma=0, adaptint=adaptint, condmax=1e15),
                dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=1e-6 * np.eye(d), mu=np.zeros(d), lam=illcond_gauss_precision(d)), d)
    raise ValueError(name)
