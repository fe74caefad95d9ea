"""oracle/run1.py -- TEST INFRASTRUCTURE. Not part of the product path.

The one-evaluation-per-invocation file protocol of the reference (mcmc_main_one, mcmc_main.F90:49-70 -> MCMC_run1,
MCMC_run1.F90:31-256, or MCMC_run1_er, MCMC_run1_er.F90:28-234), restated with the files held in a dict:

    invoke(files, cfg, prob, seed)   = one program run: MCMC_init, MCMC_run1[_er], MCMC_writechains

The arithmetic (MCMC_alpha / MCMC_DR_alpha13 / MCMC_reject / MCMC_propose / MCMC_sscrit) is the C restatement's
(mcx_oracle.c: mcxo_run1_*), on a chain object created per invocation like MCMC_init does (R, R2, iC from cmat0, the
stream keyed by this invocation's seed).  `files` keys: the namelist /mcmcrun/ (drstage, isimu, ieval, nrej, alpha12,
sscrit: mcmcrun1.F90:20-23) and the data files the reference reads and writes -- par (mcmcpar.dat, the caller's),
mean (meanfile), parf (parffile), oldpar1, oldpar2, ssprev1, ssprev2, parnew, sscritfile, accepted, done.

Pinned against the real reference: tests/test_oracle_run1.py drives oracle/_ref/mcxref_one invocation by invocation.
"""
import ctypes as C
import copy
import numpy as np
from . import pyoracle as po

_DP = C.POINTER(C.c_double)
HUGE = np.finfo(np.float64).max


def new_files(par0):
    """init_mcmcrun_namelist (mcmcrun1.F90:27-37) + the caller's starting point in mcmcpar.dat; nothing else exists yet"""
    return dict(drstage=1, isimu=1, ieval=0, nrej=0, alpha12=0.0, sscrit=-1.0,
                par=np.array(par0, dtype=np.float64), mean=None, parf=None, oldpar1=None, oldpar2=None,
                ssprev1=None, ssprev2=None, parnew=None, sscritfile=None, accepted=None, done=False)


def _bind():
    L = po.lib()
    if not getattr(L, "_run1_bound", False):
        L.mcxo_run1_decide.restype = C.c_int
        L.mcxo_run1_decide.argtypes = [C.POINTER(po.Chain), C.c_int, _DP, _DP, C.c_double, _DP, _DP, C.c_double, C.c_double,
                                       _DP, _DP, C.c_double, _DP]
        L.mcxo_run1_propose.restype = None
        L.mcxo_run1_propose.argtypes = [C.POINTER(po.Chain), C.c_int, _DP, _DP]
        L.mcxo_run1_sscrit.restype = C.c_double
        L.mcxo_run1_sscrit.argtypes = [C.POINTER(po.Chain), _DP, C.c_double]
        L.mcxo_ssfun_cols.restype = None
        L.mcxo_priorfun.restype = C.c_double
        L.mcxo_priorfun.argtypes = [C.POINTER(po.Target), _DP]
        L.mcxo_checkbounds.restype = C.c_int
        L.mcxo_checkbounds.argtypes = [C.POINTER(po.Target), _DP]
        L._run1_bound = True
    return L


def _p(a):
    return a.ctypes.data_as(_DP)


class _Inv:
    """one invocation's MCMC_init: the chain object + the user's functions"""

    def __init__(self, cfg, prob, par0, seed):
        self.L = _bind()
        self.prob = copy.copy(prob)
        self.prob.par0 = np.array(par0, dtype=np.float64)
        self.lc = po.LiveChain(cfg, self.prob, seed=seed, chain_id=0)
        self.tgt = self.lc._tgt
        self.n, self.ny = prob.npar, prob.ny
        self.dodr = bool(self.lc.ch.contents.cfg.dodr)

    def ss(self, th):
        out = np.zeros(max(self.ny, 1))
        self.L.mcxo_ssfun_cols(C.byref(self.tgt), _p(th), _p(out))
        return out[:self.ny]

    def pri(self, th):
        return float(self.L.mcxo_priorfun(C.byref(self.tgt), _p(np.ascontiguousarray(th, dtype=np.float64))))

    def pri_of_ss(self, ssv):
        """MCMC_priorfun(ssprev1) (MCMC_run1.F90:126,133): the reference hands the ss vector to priorfun; the first nycol
        entries of an npar-vector, the rest zero (what the engine's shim does as well)"""
        pad = np.zeros(self.n)
        k = min(self.n, self.ny)
        pad[:k] = ssv[:k]
        return self.pri(pad)

    def inb(self, th):
        return bool(self.L.mcxo_checkbounds(C.byref(self.tgt), _p(th)))

    def decide(self, drstage, oldpar2, ssprev2, sspri2, oldpar1, ssprev1, sspri1, alpha12, newpar, ss, sspri):
        a = C.c_double(0.0)
        rej = self.L.mcxo_run1_decide(self.lc.ch, drstage, _p(oldpar2), _p(ssprev2), sspri2, _p(oldpar1), _p(ssprev1), sspri1,
                                      alpha12, _p(newpar), _p(ss), sspri, C.byref(a))
        return a.value, bool(rej)

    def propose(self, stage, frm):
        out = np.zeros(self.n)
        self.L.mcxo_run1_propose(self.lc.ch, stage, _p(np.ascontiguousarray(frm)), _p(out))
        return out

    def sscrit(self, ssprev1, sspri1):
        return float(self.L.mcxo_run1_sscrit(self.lc.ch, _p(ssprev1), sspri1))

    def close(self):
        self.lc.close()


def invoke(f, cfg, prob, seed):
    """one run of the program: returns the new files dict (the caller then copies parnew -> par, as the driver script of
    the reference's protocol does)"""
    f = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in f.items()}
    er = (cfg.method == po.METHODS["er"])
    iv = _Inv(cfg, prob, f["par"], seed)
    try:
        par0 = f["par"].copy()                       # MCMC_init: par0 from parfile; chainmean = par0 (MCMC_init.F90:99-101)
        chainmean = par0.copy()
        dodr = iv.dodr and not er                    # MCMC_run1_er.F90:55
        reject = False
        alpha = 0.0
        if er:
            f["drstage"] = 1
        if f["isimu"] == 1:                          # MCMC_run1.F90:62-93 / MCMC_run1_er.F90:72-100
            newpar = par0.copy(); oldpar1 = par0.copy(); oldpar2 = par0.copy()
            sspri = iv.pri(newpar)
            ss = iv.ss(newpar)
            f["ieval"] += 1
            ssprev1 = ss.copy(); ssprev2 = ss.copy()
            sspri1 = sspri                           # ER: the reference leaves sspri1 unset in its first invocation (MCMC_run1_er.F90:43);
                                                     # as compiled here it holds sspri, the prior of the point just evaluated -- which
                                                     # is what it stands for -- so that is the value stated (and what the shim does)
            f["isimu"] += 1
            f["nrej"] = 1
            if f["mean"] is not None:
                par0 = f["mean"].copy()
        else:
            if not er and f["drstage"] > 1 and dodr:  # MCMC_run1.F90:97-105
                oldpar2 = f["oldpar2"].copy(); oldpar1 = f["oldpar1"].copy()
                ssprev2 = f["ssprev2"].copy(); ssprev1 = f["ssprev1"].copy()
            else:
                oldpar1 = f["parf"].copy(); ssprev1 = f["ssprev1"].copy()
                # Without DR the reference never loads oldpar2 / ssprev2 (locals, MCMC_run1.F90:44-45): after an accept they
                # are set before use, after a REJECT the next proposal starts from whatever the stack holds (observed: zeros,
                # 2.6e-260).  Stated here, and in the shim, as what the protocol means: the last accepted point of mcmcparf.dat.
                oldpar2 = oldpar1.copy(); ssprev2 = ssprev1.copy()
            par0 = f["mean"].copy()                  # :108: the point evaluated is the one in meanfile
            newpar = par0.copy()
            sspri = iv.pri(newpar)
            ss = iv.ss(newpar)
            f["ieval"] += 1
            sspri1 = iv.pri_of_ss(ssprev1)
            if er:                                   # MCMC_run1_er.F90:131-152
                s = 0.0
                for j in range(iv.ny):
                    s = s + ss[j] / prob.sigma2v[j]
                reject = bool(s >= f["sscrit"])
                f["isimu"] += 1
                if not reject:
                    f["nrej"] = 1
                    oldpar1 = newpar.copy(); ssprev1 = ss.copy()
                    f["alpha12"] = alpha
                    sspri1 = sspri
            else:
                sspri2 = iv.pri_of_ss(ssprev2) if (f["drstage"] > 1 and dodr) else 0.0
                alpha, reject = iv.decide(f["drstage"], oldpar2, ssprev2, sspri2, oldpar1, ssprev1, sspri1, f["alpha12"],
                                          newpar, ss, sspri)
                if reject:                           # MCMC_run1.F90:145-172
                    if dodr:
                        if f["drstage"] == 1:
                            f["drstage"] = 2
                            ssprev2 = ssprev1.copy(); ssprev1 = ss.copy()
                            oldpar2 = oldpar1.copy(); oldpar1 = newpar.copy()
                            f["alpha12"] = alpha
                        else:
                            f["isimu"] += 1
                            f["drstage"] = 1
                            ssprev1 = ssprev2.copy(); oldpar1 = oldpar2.copy()
                    else:
                        f["isimu"] += 1
                        f["drstage"] = 1
                else:
                    f["drstage"] = 1
                    f["isimu"] += 1
                    f["nrej"] = 1
                    oldpar1 = newpar.copy(); ssprev1 = ss.copy()
                    f["alpha12"] = alpha
                    ssprev2 = ssprev1.copy(); oldpar2 = oldpar1.copy()
        # the next value, until one is inside the bounds (MCMC_run1.F90:180-212 / MCMC_run1_er.F90:162-190)
        inbounds = False
        f["nrej"] = 1
        cur = oldpar1 if er else oldpar2
        while not inbounds:
            newpar = iv.propose(2 if (not er and f["drstage"] > 1 and dodr) else 1, cur)
            inbounds = iv.inb(newpar)
            if not inbounds:
                if not er and f["drstage"] == 1 and dodr:
                    f["drstage"] = 2
                    oldpar1 = newpar.copy()
                    ssprev1 = np.full(iv.ny, HUGE)
                    f["alpha12"] = 0.0
                else:
                    f["drstage"] = 1
                    f["isimu"] += 1
                    f["nrej"] += 1
                    if not er:
                        oldpar1 = oldpar2.copy(); ssprev1 = ssprev2.copy()
            elif er:
                sspri = iv.pri(newpar)
                crit = iv.sscrit(ssprev1, sspri1)
                if sspri >= crit:
                    inbounds = False
                    f["isimu"] += 1
                    f["nrej"] += 1
                    f["sscrit"] = crit
                else:
                    f["sscrit"] = crit - sspri
        # files written by MCMC_run1 (:226-248) and MCMC_writechains (MCMC_aux.F90:53-56)
        f["parnew"] = newpar.copy()
        f["ssprev1"] = ssprev1.copy()
        if er:
            f["sscritfile"] = f["sscrit"]
            f["parf"] = oldpar1.copy()
        else:
            if dodr:
                f["oldpar2"] = oldpar2.copy(); f["oldpar1"] = oldpar1.copy(); f["ssprev2"] = ssprev2.copy()
            f["parf"] = oldpar2.copy()
        f["mean"] = chainmean
        f["accepted"] = not reject
        f["done"] = f["done"] or (f["ieval"] >= cfg.nsimu)
        f["alpha"] = alpha
        f["chainrow"] = np.concatenate([f["parf"], [float(f["nrej"])]])
        return f
    finally:
        iv.close()
