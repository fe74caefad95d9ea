"""Load a tests/golden/*.npz fixture (made by oracle/gen_golden.py from the real reference)."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load(name, po):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ckw = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    derived = {k: ckw.pop(k) for k in ("dodr", "doscam", "usesvd")}
    cfg = po.make_cfg(**ckw)
    for k, v in derived.items():
        assert getattr(cfg, k) == v
    pkw = {}
    for k in z.files:
        if k.startswith("prob_"):
            v = z[k]
            pkw[k[5:]] = v.item() if v.ndim == 0 else v
    pkw["kind"] = str(pkw["kind"])
    prob = po.Problem(**pkw)
    return z, cfg, prob


def accepted_from_runlen(runlen):
    cnt = np.asarray(runlen, dtype=np.int64)
    acc = np.zeros(int(cnt.sum()), dtype=np.uint8)
    acc[np.concatenate([[0], np.cumsum(cnt)[:-1]])] = 1
    return acc


def logged_factors(z, cfg):
    """A fixture made with MKL's dgesvd and its call log (oracle/gen_golden.py, MKL_LOGGED): [(iteration, U, qcovstd)] with
    iteration 0 = MCMC_init's MCMC_calculate_R, then one entry per adaptation; scam_svd's floor applied to the logged
    singular values (matutils.F90:633-645)."""
    n = z["svd_s"].shape[1]
    out = []
    for k, sv in enumerate(z["svd_s"]):
        sv = sv.copy()
        tol = sv[0] / cfg.condmax
        if sv[-1] <= tol:
            sv[sv < tol] = tol
        U = np.eye(n) if k == 0 else z["svd_U"][k - 1]
        out.append((0 if k == 0 else int(z["svd_ticks"][k - 1]), U, np.sqrt(sv)))
    return out
