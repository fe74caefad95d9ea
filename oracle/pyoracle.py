"""oracle/pyoracle.py -- TEST INFRASTRUCTURE. Not part of the product path.

ctypes binding of the C restatement (oracle/mcx_oracle.c -> oracle/_build/libmcxoracle.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("MCX_ORACLE_LIB") or os.path.join(_HERE, "_build", "libmcxoracle.so")   # override: the sanitizer build

METHODS = {"dram": 0, "ram": 1, "scam": 2, "er": 3}
TARGETS = {"gauss": 0, "banana": 1, "expdata": 2}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("nsimu", "doadapt", "doburnin", "adaptint", "adapthist", "badaptint", "adaptend", "initcmatn",
                 "burnintime", "greedy", "updatesigma", "method")] + \
               [(n, C.c_double) for n in
                ("scalelimit", "scalefactor", "drscale", "N0", "S02", "condmax", "alphatarget", "nuparam")] + \
               [(n, C.c_int) for n in ("dodr", "doscam", "usesvd")]


_DP = C.POINTER(C.c_double)


class Target(C.Structure):
    _fields_ = [("kind", C.c_int), ("npar", C.c_int), ("mu", _DP), ("lam", _DP), ("banana_b", C.c_double),
                ("ndata", C.c_int), ("xdata", _DP), ("ydata", _DP), ("lo", _DP), ("hi", _DP),
                ("pri_mu", _DP), ("pri_sig", _DP), ("ny", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 2), ("n", C.c_uint64), ("saved", C.c_int), ("saved_y", C.c_double)]


class Chain(C.Structure):
    _fields_ = [("cfg", Cfg), ("tgt", Target), ("rng", Rng), ("npar", C.c_int),
                ("par0", _DP), ("cmat0", _DP), ("sigma2", C.c_double), ("S02", C.c_double), ("nobs", C.c_int),
                ("R", _DP), ("R2", _DP), ("iC", _DP), ("chaincmat", _DP), ("chainmean", _DP),
                ("chainwsum", C.c_double), ("chain", _DP), ("sschain", _DP), ("s2chain", _DP),
                ("accepted", C.POINTER(C.c_uint8)), ("alpha_trace", _DP),
                ("simuind", C.c_int), ("chainind", C.c_int),
                ("stayed", C.c_int), ("bndstayed", C.c_int), ("draccepted", C.c_int), ("drtries", C.c_int),
                ("nprop", C.c_uint64),
                ("ad_istart", C.c_int), ("ad_istartind", C.c_int), ("ad_lastind", C.c_int), ("ad_lastfreq", C.c_int),
                ("info_last", C.c_int), ("ram_downdate_fail", C.c_int),
                ("oldpar", _DP), ("ss1", C.c_double), ("sspri1", C.c_double), ("alpha12", C.c_double),
                ("continue_on_downdate_fail", C.c_int), ("qcovstd", _DP), ("erstayed", C.c_int),
                ("ny", C.c_int), ("ss1v", C.c_double * 32), ("sigma2v", C.c_double * 32), ("nobsv", C.c_int * 32),       # MCXO_NYMAX
                ("trmv_desc", C.c_int), ("last_u", _DP), ("n_calcR", C.c_int), ("svd_floored", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.mcxo_cfg_defaults.argtypes = [C.POINTER(Cfg)]
        L.mcxo_cfg_check.argtypes = [C.POINTER(Cfg)]
        L.mcxo_cfg_check.restype = C.c_int
        L.mcxo_chain_create.restype = C.POINTER(Chain)
        L.mcxo_chain_create.argtypes = [C.POINTER(Cfg), C.POINTER(Target), _DP, _DP, C.c_double, C.c_int,
                                        C.c_uint32, C.c_uint32]
        L.mcxo_chain_create_ny.restype = C.POINTER(Chain)
        L.mcxo_chain_create_ny.argtypes = [C.POINTER(Cfg), C.POINTER(Target), _DP, _DP, _DP, C.POINTER(C.c_int), C.c_int,
                                           C.c_uint32, C.c_uint32]
        L.mcxo_ssfun_cols.argtypes = [C.POINTER(Target), _DP, _DP]
        L.mcxo_chain_free.argtypes = [C.POINTER(Chain)]
        L.mcxo_chain_run.argtypes = [C.POINTER(Chain), C.c_int]
        L.mcxo_chain_run.restype = C.c_int
        for f in ("mcxo_log", "mcxo_exp"):
            getattr(L, f).argtypes = [C.c_double]
            getattr(L, f).restype = C.c_double
        L.mcxo_normal.argtypes = [C.POINTER(Rng)]
        L.mcxo_normal.restype = C.c_double
        L.mcxo_gamma.argtypes = [C.POINTER(Rng), C.c_double, C.c_double]
        L.mcxo_gamma.restype = C.c_double
        L.mcxo_trmv_ut.argtypes = [C.c_int, _DP, _DP]
        L.mcxo_trmv_ut_desc.argtypes = [C.c_int, _DP, _DP]
        L.mcxo_potrf_u.argtypes = [C.c_int, _DP]
        L.mcxo_potrf_u.restype = C.c_int
        L.mcxo_potri_u.argtypes = [C.c_int, _DP]
        L.mcxo_potri_u.restype = C.c_int
        L.mcxo_chud.argtypes = [C.c_int, _DP, _DP, _DP, _DP]
        L.mcxo_chdd.argtypes = [C.c_int, _DP, _DP, _DP, _DP]
        L.mcxo_chdd.restype = C.c_int
        L.mcxo_symsvd.argtypes = [C.c_int, _DP, _DP, _DP]
        L.mcxo_symsvd.restype = C.c_int
        L.mcxo_gemv.argtypes = [C.c_int, C.c_int, _DP, _DP, _DP]
        L.mcxo_covmat.argtypes = [C.c_int, C.c_int, _DP, C.c_int, _DP, C.c_int, _DP, _DP, _DP, C.c_int]
        L.mcxo_alpha.argtypes = [C.c_double] * 5
        L.mcxo_alpha.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(_DP) if a is not None else None


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def make_cfg(**kw):
    """namelist /mcmc/ defaults (mcmcinit.F90:184-230) overridden by kw, then the checks (:235-368)."""
    c = Cfg()
    lib().mcxo_cfg_defaults(C.byref(c))
    for k, v in kw.items():
        if k == "method":
            v = METHODS[v] if isinstance(v, str) else v
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    rc = lib().mcxo_cfg_check(C.byref(c))
    if rc != 0:
        raise ValueError("mcmcinit check failed (reference would stop)")
    return c


class Problem:
    """Target + initial values; keeps numpy buffers alive."""

    def __init__(self, kind, npar, par0, cmat0, sigma2=1.0, nobs=1, mu=None, lam=None, b=0.1, xdata=None,
                 ydata=None, lo=None, hi=None, pri_mu=None, pri_sig=None):
        self.kind, self.npar = kind, int(npar)
        self.par0 = f64(par0).reshape(npar)
        self.cmat0 = f64(cmat0).reshape(npar, npar)
        # nycol > 1 (expdata with ydata [ny][ndata], npar = 1 + ny): sigma2 and nobs are vectors of length ny
        self.sigma2v = f64(np.atleast_1d(sigma2))
        self.nobsv = np.ascontiguousarray(np.atleast_1d(nobs), dtype=np.int32)
        self.ny = len(self.sigma2v)
        assert len(self.nobsv) == self.ny
        self.sigma2, self.nobs = float(self.sigma2v[0]), int(self.nobsv[0])
        self.mu = f64(mu) if mu is not None else None
        self.lam = f64(lam) if lam is not None else None      # row-major lam[i, j]
        self.b = float(b)
        self.xdata = f64(xdata) if xdata is not None else None
        self.ydata = f64(ydata) if ydata is not None else None
        self.lo = f64(lo) if lo is not None else None
        self.hi = f64(hi) if hi is not None else None
        self.pri_mu = f64(pri_mu) if pri_mu is not None else None
        self.pri_sig = f64(pri_sig) if pri_sig is not None else None

    def ctarget(self):
        t = Target()
        t.kind, t.npar = TARGETS[self.kind], self.npar
        t.mu, t.lam, t.banana_b = _dp(self.mu), _dp(self.lam), self.b
        t.ndata = 0 if self.xdata is None else len(self.xdata)
        t.xdata, t.ydata = _dp(self.xdata), _dp(self.ydata)
        t.lo, t.hi, t.pri_mu, t.pri_sig = _dp(self.lo), _dp(self.hi), _dp(self.pri_mu), _dp(self.pri_sig)
        t.ny = self.ny
        if self.ny > 1:
            assert self.kind == "expdata" and self.ydata.size == self.ny * t.ndata and self.npar == 1 + self.ny
        return t


class Result:
    pass


def run_chain(cfg, prob, seed=0x6D636D63, chain_id=0, upto=None, continue_on_downdate_fail=False):
    """Run one chain through the oracle; returns the reference-visible outputs."""
    L = lib()
    tgt = prob.ctarget()
    cm = np.asfortranarray(prob.cmat0)            # column-major like the reference
    ch = L.mcxo_chain_create_ny(C.byref(cfg), C.byref(tgt), _dp(prob.par0), cm.ctypes.data_as(_DP),
                                _dp(prob.sigma2v), prob.nobsv.ctypes.data_as(C.POINTER(C.c_int)), prob.ny, seed, chain_id)
    if not ch:
        raise RuntimeError("could not factor the initial covariance")
    try:
        ch.contents.continue_on_downdate_fail = 1 if continue_on_downdate_fail else 0
        rc = L.mcxo_chain_run(ch, cfg.nsimu if upto is None else upto)
        c = ch.contents
        n, ns = prob.npar, c.simuind
        r = Result()
        r.rc = rc
        r.simuind, r.chainind = c.simuind, c.chainind
        r.chain = np.ctypeslib.as_array(c.chain, shape=(cfg.nsimu, n + 1))[:c.chainind].copy()
        ny = prob.ny
        r.sschain = np.ctypeslib.as_array(c.sschain, shape=(cfg.nsimu, ny + 1))[:c.chainind].copy()
        r.s2chain = np.ctypeslib.as_array(c.s2chain, shape=(cfg.nsimu, ny))[:ns].copy()
        if ny == 1:
            r.s2chain = r.s2chain[:, 0]
        r.accepted = np.ctypeslib.as_array(c.accepted, shape=(cfg.nsimu,))[:ns].copy()
        r.alpha = np.ctypeslib.as_array(c.alpha_trace, shape=(cfg.nsimu,))[:ns].copy()
        r.R = np.ctypeslib.as_array(c.R, shape=(n, n)).T.copy()            # -> R[i, j]
        r.R2 = np.ctypeslib.as_array(c.R2, shape=(n, n)).T.copy()
        r.iC = np.ctypeslib.as_array(c.iC, shape=(n, n)).T.copy()
        r.chaincmat = np.ctypeslib.as_array(c.chaincmat, shape=(n, n)).T.copy()
        r.chainmean = np.ctypeslib.as_array(c.chainmean, shape=(n,)).copy()
        r.chainwsum = c.chainwsum
        r.qcovstd = np.ctypeslib.as_array(c.qcovstd, shape=(n,)).copy()
        r.sigma2 = c.sigma2
        r.sigma2v = np.array(c.sigma2v[:ny]); r.ss1v = np.array(c.ss1v[:ny])
        r.theta = np.ctypeslib.as_array(c.oldpar, shape=(n,)).copy()
        r.ss1, r.sspri1 = c.ss1, c.sspri1
        r.stayed, r.bndstayed, r.draccepted, r.drtries = c.stayed, c.bndstayed, c.draccepted, c.drtries
        r.nprop = c.nprop
        r.erstayed = c.erstayed
        r.rng_n = c.rng.n
        r.rng_saved, r.rng_saved_y = c.rng.saved, c.rng.saved_y
        r.ram_downdate_fail = c.ram_downdate_fail
        return r
    finally:
        L.mcxo_chain_free(ch)


class LiveChain:
    """An oracle chain kept alive between segments, whose proposal factor can be replaced from outside
    (used to restate the engine's pooled mode, where adaptation happens across chains)."""

    def __init__(self, cfg, prob, seed=0x6D636D63, chain_id=0):
        L = lib()
        self.cfg, self.prob, self.n = cfg, prob, prob.npar
        self._tgt = prob.ctarget()
        cm = np.asfortranarray(prob.cmat0)
        self.ch = L.mcxo_chain_create_ny(C.byref(cfg), C.byref(self._tgt), _dp(prob.par0), cm.ctypes.data_as(_DP),
                                         _dp(prob.sigma2v), prob.nobsv.ctypes.data_as(C.POINTER(C.c_int)), prob.ny, seed, chain_id)
        if not self.ch:
            raise RuntimeError("could not factor the initial covariance")

    def run(self, upto):
        rc = lib().mcxo_chain_run(self.ch, upto)
        assert rc == 0, rc

    def set_R(self, R):
        """R[i, j] upper triangular -> the chain's column-major factor."""
        view = np.ctypeslib.as_array(self.ch.contents.R, shape=(self.n, self.n))      # view[j, i] = R(i, j)
        view[:, :] = np.asarray(R, dtype=np.float64).T

    def set_dr(self, R2, iC):
        """delayed rejection: the second-stage factor R2[i, j] and the inverse covariance iC[i, j] (upper triangles)."""
        v2 = np.ctypeslib.as_array(self.ch.contents.R2, shape=(self.n, self.n))
        v2[:, :] = np.asarray(R2, dtype=np.float64).T
        vi = np.ctypeslib.as_array(self.ch.contents.iC, shape=(self.n, self.n))
        vi[:, :] = np.asarray(iC, dtype=np.float64).T

    def set_qcovstd(self, std):
        """SCAM: the proposal standard deviations along the rotated axes."""
        np.ctypeslib.as_array(self.ch.contents.qcovstd, shape=(self.n,))[:] = np.asarray(std, dtype=np.float64)

    @property
    def theta(self):
        return np.ctypeslib.as_array(self.ch.contents.oldpar, shape=(self.n,)).copy()

    @property
    def last_u(self):
        """normal deviates of the latest first-stage proposal"""
        return np.ctypeslib.as_array(self.ch.contents.last_u, shape=(self.n,)).copy()

    @property
    def alpha12(self):
        return float(self.ch.contents.alpha12)

    @property
    def stayed(self):
        return int(self.ch.contents.stayed)

    @property
    def drtries(self):
        return int(self.ch.contents.drtries)

    @property
    def draccepted(self):
        return int(self.ch.contents.draccepted)

    @property
    def erstayed(self):
        return int(self.ch.contents.erstayed)

    @property
    def accepted(self):
        c = self.ch.contents
        return np.ctypeslib.as_array(c.accepted, shape=(self.cfg.nsimu,))[:c.simuind].copy()

    def close(self):
        if self.ch:
            lib().mcxo_chain_free(self.ch)
            self.ch = None
