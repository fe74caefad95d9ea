"""Loads a tests/golden/run1/*.npz fixture (oracle/gen_golden_run1.py: the real reference's mcmc_main_one, invocation by
invocation) and compares one invocation's files dict (oracle/run1.py layout) with it."""
import os
import numpy as np

RUN1 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "run1")


def names():
    return sorted(f[:-4] for f in os.listdir(RUN1) if f.endswith(".npz"))


def load(name, po):
    z = np.load(os.path.join(RUN1, name + ".npz"))
    ckw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    for k in ("dodr", "doscam", "usesvd"):
        ckw.pop(k)
    cfg = po.make_cfg(**ckw)
    pkw = {}
    for k in z.files:
        if k.startswith("prob_"):
            v = z[k]
            pkw[k[5:]] = v.item() if v.ndim == 0 else v
    pkw["kind"] = str(pkw["kind"])
    return z, cfg, po.Problem(**pkw)


def check_invocation(z, k, f, rtol, what):
    """integers and the accept flag exactly; floating point to rtol (the reference's BLAS / libm against the pinned ones)"""
    got = [f["drstage"], f["isimu"], f["ieval"], f["nrej"]]
    assert got == list(z["nml"][k]), "%s, invocation %d: drstage/isimu/ieval/nrej %s, reference %s" % (what, k, got, list(z["nml"][k]))
    assert bool(f["accepted"]) == bool(z["accepted"][k]), "%s, invocation %d: accept flag" % (what, k)
    np.testing.assert_allclose(f["parnew"], z["parnew"][k], rtol=rtol, atol=1e-13, err_msg="%s, invocation %d: mcmcparnew.dat" % (what, k))
    np.testing.assert_allclose(f["parf"], z["parf"][k], rtol=rtol, atol=1e-13, err_msg="%s, invocation %d: mcmcparf.dat" % (what, k))
    np.testing.assert_allclose(f["alpha12"], z["alpha12"][k], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(f["sscrit"], z["sscrit"][k], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(f["chainrow"], z["chainrow"][k], rtol=rtol, atol=1e-13)
    if np.all(np.asarray(z["ssprev1"][k]) < 1e300):
        np.testing.assert_allclose(f["ssprev1"], z["ssprev1"][k], rtol=1e-7)
