"""Host-side mirror of mcmcf90's user surface on top of the C ABI.

`McmcConfig` carries the numeric members of namelist /mcmc/ (mcmcinit.F90:74-82) with the reference's
defaults (:184-230); `Engine` follows the call sequence of a reference driver program:
MCMC_setpar0 / MCMC_setcmat0 / MCMC_setsigma2nobs (MCMC_init.F90:168-347, testcases/mcmcrun3.F90:18-23),
then mcmc_main() = init + run (mcmc_main.F90:12-44), then the chain/sschain/s2chain arrays.
Every call goes through libmcmcx.so; nothing is computed in Python.
"""
import ctypes as C
import numpy as np
from . import _lib

METHODS = {"dram": 0, "ram": 1, "scam": 2, "er": 3}


class McmcError(RuntimeError):
    pass


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def make_config(npar, nchains=1, **kw):
    c = _lib.Config()
    _lib.load().mcmcx_config_defaults(C.byref(c))
    c.npar, c.nchains = int(npar), int(nchains)
    for k, v in kw.items():
        if k == "method" and isinstance(v, str):
            v = METHODS[v]
        if not hasattr(c, k):
            raise KeyError(k)
        setattr(c, k, v)
    return c


class Comm:
    """The node's communicator (include/mcmcx.h, "several GPUs of one node"): one process per GPU, RCCL underneath
    (backend "rccl"), or the host-staged transport for ranks that share one GPU (backend "host")."""
    BACKENDS = {"rccl": 0, "host": 1}

    def __init__(self, key, rank, nranks, device, backend="rccl"):
        self.L = _lib.load()
        self.h = C.c_void_p()
        rc = self.L.mcmcx_comm_create(str(key).encode(), int(rank), int(nranks), int(device), self.BACKENDS[backend], C.byref(self.h))
        if rc < 0:
            raise McmcError(self.L.mcmcx_last_error().decode())
        self.rank, self.size = int(rank), int(nranks)

    def _chk(self, rc):
        if rc < 0:
            raise McmcError(self.L.mcmcx_last_error().decode())

    def barrier(self):
        self._chk(self.L.mcmcx_comm_barrier(self.h))

    def allreduce(self, values, op="sum"):
        """Sum / maximum of a few host doubles over the ranks."""
        a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        self._chk(self.L.mcmcx_comm_allreduce_host(self.h, _dp(a), a.size, 1 if op == "max" else 0))
        return a

    def close(self):
        if self.h:
            self.L.mcmcx_comm_destroy(self.h)
            self.h = C.c_void_p()


class Engine:
    def __init__(self, cfg):
        self.L = _lib.load()
        self.cfg = cfg
        self.h = C.c_void_p()
        self._chk(self.L.mcmcx_create(C.byref(cfg), C.byref(self.h)))
        self.npar, self.nchains, self.nsimu = cfg.npar, cfg.nchains, cfg.nsimu

    def _chk(self, rc):
        if rc < 0:
            raise McmcError(self.L.mcmcx_last_error().decode())
        return rc

    def close(self):
        if self.h:
            self.L.mcmcx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- MCMC_setpar0 / MCMC_setcmat0 / MCMC_setsigma2nobs
    def setpar0(self, par0):
        a = _f64(par0)
        self._chk(self.L.mcmcx_set_par0(self.h, _dp(a), a.size))

    def setcmat0(self, cmat0):
        a = np.asfortranarray(np.asarray(cmat0, dtype=np.float64))
        self._chk(self.L.mcmcx_set_cmat0(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0]))

    def setsigma2nobs(self, sigma2, nobs):
        """Scalars, or vectors of length nycol (one error variance per response column of ssfunction)."""
        s = _f64(np.atleast_1d(sigma2)); n = np.ascontiguousarray(np.atleast_1d(nobs), dtype=np.int32)
        assert len(s) == len(n)
        self.nycol = len(s)
        self._chk(self.L.mcmcx_set_sigma2nobs(self.h, _dp(s), n.ctypes.data_as(C.POINTER(C.c_int32)), len(s)))

    # --- the device-resident user callbacks
    def set_target(self, kind, mu=None, lam=None, b=0.1, xdata=None, ydata=None):
        if kind == "gauss":
            self._chk(self.L.mcmcx_set_target_gauss(self.h, _dp(_f64(mu)), _dp(_f64(lam))))
        elif kind == "banana":
            self._chk(self.L.mcmcx_set_target_banana(self.h, float(b)))
        elif kind == "expdata":
            x, y = _f64(xdata), np.ascontiguousarray(np.asarray(ydata, dtype=np.float64))
            if y.ndim == 2:                                   # response columns: y[nycol][ndata]
                self._chk(self.L.mcmcx_set_target_expdata_cols(self.h, x.size, y.shape[0], _dp(x), _dp(y.ravel())))
            else:
                self._chk(self.L.mcmcx_set_target_expdata(self.h, x.size, _dp(x), _dp(y)))
        else:
            raise KeyError(kind)

    def set_target_host(self, ssfun, priorfun=None, checkbounds=None, ssfun_er=None):
        """User callbacks on the host, like the reference's link-time ssfunction / priorfun / checkbounds:
        ssfun(theta) -> float, priorfun(theta) -> float, checkbounds(theta) -> bool, theta a numpy vector;
        ssfun_er(theta, sscrit) -> float is the early-rejection form used by method='er' (default: ssfun)."""
        n = self.npar

        def _ss(th, npar, ny, out, user):
            v = np.atleast_1d(ssfun(np.ctypeslib.as_array(th, shape=(n,)).copy()))
            for j in range(ny):
                out[j] = float(v[j])

        def _pri(th, npar, user):
            return float(priorfun(np.ctypeslib.as_array(th, shape=(n,)).copy()))

        def _cb(th, npar, user):
            return 1 if checkbounds(np.ctypeslib.as_array(th, shape=(n,)).copy()) else 0

        self._cb_keep = (_lib.SSFUN_T(_ss), _lib.PRIORFUN_T(_pri) if priorfun else _lib.PRIORFUN_T(),
                         _lib.CHECKBOUNDS_T(_cb) if checkbounds else _lib.CHECKBOUNDS_T())
        self._chk(self.L.mcmcx_set_target_host(self.h, self._cb_keep[0], self._cb_keep[1], self._cb_keep[2], None))
        if ssfun_er is not None:
            def _ss_er(th, npar, ny, crit, out, user):
                v = np.atleast_1d(ssfun_er(np.ctypeslib.as_array(th, shape=(n,)).copy(), crit))
                for j in range(ny):
                    out[j] = float(v[j])
            self._er_keep = _lib.SSFUN_ER_T(_ss_er)
            self._chk(self.L.mcmcx_set_target_host_er(self.h, self._er_keep))

    def set_target_host_batch(self, ssfun_batch, priorfun=None, checkbounds=None, nthreads=1):
        """Batched user callback: ssfun_batch(theta[n, npar]) -> ss[n] or ss[n, ny], called once per stage (or from
        `nthreads` threads on disjoint slices)."""
        n = self.npar

        def _ssb(th, npar, nb, ny, out, user):
            a = np.ctypeslib.as_array(th, shape=(nb, n)).copy()
            v = np.asarray(ssfun_batch(a), dtype=np.float64).reshape(nb, ny)
            np.ctypeslib.as_array(out, shape=(nb, ny))[:, :] = v

        def _pri(th, npar, user):
            return float(priorfun(np.ctypeslib.as_array(th, shape=(n,)).copy()))

        def _cb(th, npar, user):
            return 1 if checkbounds(np.ctypeslib.as_array(th, shape=(n,)).copy()) else 0

        self._cbb_keep = (_lib.SSFUN_BATCH_T(_ssb), _lib.PRIORFUN_T(_pri) if priorfun else _lib.PRIORFUN_T(),
                          _lib.CHECKBOUNDS_T(_cb) if checkbounds else _lib.CHECKBOUNDS_T())
        self._chk(self.L.mcmcx_set_target_host_batch(self.h, self._cbb_keep[0], self._cbb_keep[1], self._cbb_keep[2], None, int(nthreads)))

    def set_target_module(self, code_object, kernel_name, userdata=None):
        """The user's ssfunction / priorfun / checkbounds as device code (include/mcmcx_target.h): `code_object` is the
        hipcc --genco output, userdata a bytes-like object or a float64 array copied to the device."""
        buf = None if userdata is None else np.ascontiguousarray(userdata)
        self._chk(self.L.mcmcx_set_target_module(self.h, str(code_object).encode(), str(kernel_name).encode(),
                                                 None if buf is None else buf.ctypes.data_as(C.c_void_p), 0 if buf is None else buf.nbytes))

    def set_bounds(self, lo=None, hi=None):
        lo = _f64(lo) if lo is not None else None
        hi = _f64(hi) if hi is not None else None
        self._chk(self.L.mcmcx_set_bounds(self.h, _dp(lo), _dp(hi)))

    def set_priors(self, mu, sig):
        self._chk(self.L.mcmcx_set_priors(self.h, _dp(_f64(mu)), _dp(_f64(sig))))

    # --- mcmc_main_one: MCMC_run1 / MCMC_run1_er, the arithmetic of one invocation (MCMC_run1.F90:131-199); all chains at once
    def set_target_external(self):
        """Every evaluation of ssfunction / priorfun / checkbounds is the caller's: init evaluates nothing, run refuses."""
        self._chk(self.L.mcmcx_set_target_external(self.h))

    def _rows(self, a, k):
        return None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64).reshape(-1, k), (self.nchains, k)))

    def run1_decide(self, drstage, oldpar1, ssprev1, sspri1, newpar, ss, sspri, oldpar2=None, ssprev2=None, sspri2=None, alpha12=None):
        """(alpha[nchains], reject[nchains]): MCMC_alpha(oldpar1 -> newpar) in stage 1, MCMC_DR_alpha13(oldpar2, oldpar1, newpar)
        in stage 2 with drscale > 0, then MCMC_reject.  Vectors [nchains][npar] / [nchains][nycol] / [nchains] (or one row for all)."""
        n, ny = self.npar, getattr(self, "nycol", 1)
        A = [self._rows(oldpar2, n), self._rows(ssprev2, ny), self._rows(sspri2, 1), self._rows(oldpar1, n), self._rows(ssprev1, ny),
             self._rows(sspri1, 1), self._rows(alpha12, 1), self._rows(newpar, n), self._rows(ss, ny), self._rows(sspri, 1)]
        alpha = np.zeros(self.nchains); rej = np.zeros(self.nchains, dtype=np.int32)
        self._chk(self.L.mcmcx_run1_decide(self.h, int(drstage), *[_dp(a) for a in A], _dp(alpha), rej.ctypes.data_as(C.POINTER(C.c_int32))))
        return alpha, rej.astype(bool)

    def run1_propose(self, stage, from_):
        out = np.zeros((self.nchains, self.npar))
        self._chk(self.L.mcmcx_run1_propose(self.h, int(stage), _dp(self._rows(from_, self.npar)), _dp(out)))
        return out

    def run1_sscrit(self, ssprev1, sspri1):
        out = np.zeros(self.nchains)
        self._chk(self.L.mcmcx_run1_sscrit(self.h, _dp(self._rows(ssprev1, getattr(self, "nycol", 1))), _dp(self._rows(sspri1, 1)), _dp(out)))
        return out

    # --- mcmc_main
    def set_stream(self, hip_stream):
        """Run on the caller's HIP stream (e.g. torch.cuda.Stream().cuda_stream) instead of the engine's own; before init."""
        self._chk(self.L.mcmcx_set_stream(self.h, C.c_void_p(int(hip_stream))))

    def init(self):
        self._chk(self.L.mcmcx_init(self.h))

    def run(self, upto=None):
        """Returns 0, or 2 (MCMCX_INTERRUPTED) when a caught signal ended the run early (see mcmcx_run)."""
        return self._chk(self.L.mcmcx_run(self.h, self.nsimu if upto is None else int(upto)))

    def sync(self):
        self._chk(self.L.mcmcx_sync(self.h))

    @property
    def simuind(self):
        return self.L.mcmcx_simuind(self.h)

    # --- results
    def counters(self, chain=0):
        a = np.zeros(8, dtype=np.int32)
        self._chk(self.L.mcmcx_get_counters(self.h, chain, a.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(stayed=int(a[0]), bndstayed=int(a[1]), draccepted=int(a[2]), drtries=int(a[3]),
                    chainind=int(a[4]), status=int(a[5]), erstayed=int(a[6]), curcount=int(a[7]))

    def totals(self):
        a = np.zeros(7, dtype=np.int64)
        self._chk(self.L.mcmcx_get_totals_n(self.h, a.ctypes.data_as(C.POINTER(C.c_int64)), a.size))
        return dict(stayed=int(a[0]), bndstayed=int(a[1]), draccepted=int(a[2]), drtries=int(a[3]), proposals=int(a[4]),
                    downdates=int(a[5]), status=int(a[6]))

    def theta(self):
        a = np.zeros((self.nchains, self.npar))
        self._chk(self.L.mcmcx_get_theta(self.h, _dp(a)))
        return a

    def scalars(self):
        a = np.zeros((self.nchains, 4))
        self._chk(self.L.mcmcx_get_scalars(self.h, _dp(a)))
        return a          # ss1, sspri1, sigma2, alpha12

    def rng(self, chain=0):
        n, s, y = C.c_uint64(), C.c_int32(), C.c_double()
        self._chk(self.L.mcmcx_get_rng(self.h, chain, C.byref(n), C.byref(s), C.byref(y)))
        return n.value, s.value, y.value

    def R(self, chain=0):
        a = np.zeros((self.npar, self.npar), order="F")
        self._chk(self.L.mcmcx_get_R(self.h, chain, a.ctypes.data_as(C.POINTER(C.c_double))))
        return np.array(a)

    def qcovstd(self, chain=0):
        a = np.zeros(self.npar)
        self._chk(self.L.mcmcx_get_qcovstd(self.h, chain, _dp(a)))
        return a

    def dr_state(self, chain=0):
        r2 = np.zeros((self.npar, self.npar), order="F"); ic = np.zeros((self.npar, self.npar), order="F")
        dp = C.POINTER(C.c_double)
        self._chk(self.L.mcmcx_get_dr(self.h, chain, r2.ctypes.data_as(dp), ic.ctypes.data_as(dp)))
        return np.array(r2), np.array(ic)

    def chaincov(self, chain=0):
        a = np.zeros((self.npar, self.npar), order="F"); m = np.zeros(self.npar); w = C.c_double()
        self._chk(self.L.mcmcx_get_chaincov(self.h, chain, a.ctypes.data_as(C.POINTER(C.c_double)), _dp(m), C.byref(w)))
        return np.array(a), m, w.value

    def accepted(self, chain=0):
        a = np.zeros(self.simuind, dtype=np.uint8)
        self._chk(self.L.mcmcx_get_accepted(self.h, chain, a.ctypes.data_as(C.POINTER(C.c_uint8))))
        return a

    def accept_masks(self):
        nt = C.c_int32()
        self._chk(self.L.mcmcx_get_accept_masks(self.h, None, C.byref(nt)))
        a = np.zeros((self.simuind, nt.value), dtype=np.uint64)
        self._chk(self.L.mcmcx_get_accept_masks(self.h, a.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(nt)))
        return a

    def chain(self, chain=0):
        n, ny = self.simuind, getattr(self, "nycol", 1)
        ch = np.zeros((n, self.npar + 1)); ss = np.zeros((n, ny + 1)); s2 = np.zeros((n, ny)); nr = C.c_int32()
        self._chk(self.L.mcmcx_get_chain(self.h, chain, _dp(ch), _dp(ss), _dp(s2), C.byref(nr)))
        return ch[:nr.value], ss[:nr.value], (s2[:, 0] if ny == 1 else s2)

    def pooled_moments(self):
        n = self.L.mcmcx_pooled_moments_len(self.h)
        a = np.zeros(n)
        self._chk(self.L.mcmcx_pooled_moments(self.h, _dp(a)))
        return a

    def pooled_moments_dev(self, dev_ptr):
        """Pooled moments straight into device memory (e.g. a torch tensor's data_ptr()); async on the engine stream."""
        self._chk(self.L.mcmcx_pooled_moments_dev(self.h, C.c_void_p(int(dev_ptr))))

    def set_comm(self, comm):
        """Attach the node's communicator (before init): pooled-mode ticks and allreduce_moments span all ranks."""
        self._comm = comm
        self._chk(self.L.mcmcx_set_comm(self.h, comm.h if comm is not None else None))

    def allreduce_moments(self, fetch=True):
        """Pooled moments of the chains of ALL ranks (collective).  fetch=False leaves them on the device (async)."""
        if not fetch:
            self._chk(self.L.mcmcx_allreduce_moments(self.h, None))
            return None
        a = np.zeros(self.L.mcmcx_pooled_moments_len(self.h))
        self._chk(self.L.mcmcx_allreduce_moments(self.h, _dp(a)))
        return a

    def set_exchange(self, fn, dev_ptr):
        """Pooled mode over several GPUs: fn() must all-reduce (sum) the device buffer at dev_ptr in place."""
        self._xkeep = _lib.EXCHANGE_T(lambda user: fn())
        self._chk(self.L.mcmcx_set_exchange(self.h, self._xkeep, None, C.c_void_p(int(dev_ptr))))

    def pooled(self):
        n = self.npar
        cm = np.zeros((n, n), order="F"); R = np.zeros((n, n), order="F"); m = np.zeros(n); w = C.c_double()
        dp = C.POINTER(C.c_double)
        self._chk(self.L.mcmcx_get_pooled(self.h, cm.ctypes.data_as(dp), _dp(m), C.byref(w), R.ctypes.data_as(dp)))
        return np.array(cm), m, w.value, np.array(R)

    def debug_set_factor(self, R, qcovstd=None):
        """Test probe: every chain's SVD proposal factor replaced by R[i, j] (scam: the rotation U) and qcovstd."""
        a = np.asfortranarray(np.asarray(R, dtype=np.float64))
        q = None if qcovstd is None else _f64(qcovstd)
        self._chk(self.L.mcmcx_debug_set_factor(self.h, a.ctypes.data_as(C.POINTER(C.c_double)), _dp(q)))

    def last_kernel(self):
        """Name of the sampling kernel the last run() launched (mcmcx_last_kernel)."""
        return (self.L.mcmcx_last_kernel(self.h) or b"").decode()

    def kernel_time(self, reset=False):
        ms, nl, ns = C.c_double(), C.c_int64(), C.c_int64()
        self._chk(self.L.mcmcx_kernel_time(self.h, C.byref(ms), C.byref(nl), C.byref(ns), int(reset)))
        return ms.value, nl.value, ns.value


def kernel_table():
    """[(family, name), ...]: every entry of the engine's kernel-selection tables (mcmcx_debug_kernel_table); no device needed."""
    L = _lib.load()
    n = L.mcmcx_debug_kernel_table(-1, None, 0)
    out = []
    for i in range(n):
        buf = C.create_string_buffer(128)
        L.mcmcx_debug_kernel_table(i, buf, 128)
        fam, name = buf.value.decode().split(":", 1)
        out.append((fam, name))
    return out


def engine_from_problem(cfg_kw, prob_kw, nchains=1, **extra):
    """Build an Engine from the same (cfg, problem) dictionaries the oracle / golden fixtures use."""
    pk = dict(prob_kw)
    npar = int(pk["npar"])
    extra = dict(extra)
    comm = extra.pop("comm", None)
    external = extra.pop("external", False)          # MCMC_run1: the caller evaluates the target
    cfg = make_config(npar, nchains, **cfg_kw, **extra)
    e = Engine(cfg)
    if comm is not None:
        e.set_comm(comm)
    e.setpar0(pk["par0"])
    e.setcmat0(np.asarray(pk["cmat0"], dtype=np.float64).reshape(npar, npar))
    e.setsigma2nobs(pk.get("sigma2", 1.0), pk.get("nobs", 1))
    if external:
        e.set_target_external()
        return e
    e.set_target(str(pk["kind"]), mu=pk.get("mu"), lam=pk.get("lam"), b=float(pk.get("b", 0.1)),
                 xdata=pk.get("xdata"), ydata=pk.get("ydata"))
    if pk.get("lo") is not None or pk.get("hi") is not None:
        e.set_bounds(pk.get("lo"), pk.get("hi"))
    if pk.get("pri_mu") is not None:
        e.set_priors(pk["pri_mu"], pk["pri_sig"])
    return e
