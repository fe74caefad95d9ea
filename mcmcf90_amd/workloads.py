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


# the eleven observations of BASELINE config 1's two-parameter decay model y = theta1 exp(-theta2 x) (the data of fixture c1_expdata_dram)
# and a second response column with its own rate (fixtures m1..m5: nycol = 2, one error variance per column)
C1_X = np.arange(11.0)
C1_Y = np.array([[9.33, 9.4, 8.99, 7.06, 7.13, 6.69, 4.69, 4.24, 4.77, 3.86, 4.02],
                 [9.2, 6.92597768, 5.32804721, 4.44333303, 3.28181496, 2.41072887, 2.17694223, 1.59131293, 1.02648565, 1.08065636, 0.8203814]])


def problem(name, nsimu, adaptint=100):
    """(cfg kwargs, problem kwargs, proposals per iteration and chain excluding DR retries) of configuration c2..c5, and of config 1's
    model at scale: c1 (one response column) and c1x (two columns, one sigma2 each -- the bundled testcase's semantics, MCMC_DRAM.F90:52-63)."""
    if name == "c1":
        return (dict(nsimu=nsimu, adaptint=adaptint, updatesigma=1, drscale=2.0),
                dict(kind="expdata", npar=2, par0=np.array([10.0, 0.1]), cmat0=np.diag([0.2, 0.001]), sigma2=0.5, nobs=11,
                     xdata=C1_X, ydata=C1_Y[0], lo=np.zeros(2)), 1)
    if name == "c1x":
        return (dict(nsimu=nsimu, adaptint=adaptint, updatesigma=1, drscale=2.0),
                dict(kind="expdata", npar=3, par0=np.array([9.0, 0.1, 0.2]), cmat0=np.diag([0.02, 0.0001, 0.0002]), sigma2=np.array([0.5, 0.3]),
                     nobs=np.array([11, 13]), xdata=C1_X, ydata=C1_Y, lo=np.zeros(3)), 1)
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
