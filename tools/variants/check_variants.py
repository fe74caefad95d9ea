"""tools/variants/check_variants.py -- on the GPU box: the measured negatives of tools/variants against the library's own kernels, bit for bit.

    tools/build_variant.sh neg -DMCX_VARIANTS
    MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_neg.so python tools/variants/check_variants.py

Each form is forced by its environment switch (read by tools/variants/variants.inc in a variant build only) and compared with the run of the same
engine build without the switch: states, accept ballots, stream positions, factors.  These comparisons were suite tests up to round 5."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def run(ckw, pkw, n, env, want, **ekw):
    from mcmcf90_amd import engine_from_problem
    for k in ("MCMCX_POOLED_WAVES", "MCMCX_DR_GENERAL", "MCMCX_SVD_SHARED_ROT", "MCMCX_GROUP", "MCMCX_DR_BIG", "MCMCX_COV_FIFO"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine_from_problem(ckw, pkw, nchains=n, record_accept=1, **ekw)
    e.init(); e.run()
    k = e.last_kernel()
    assert want is None or k == want, (k, want)
    out = (e.theta().copy(), e.accept_masks().copy(), [e.rng(c) for c in (0, n - 1)], [e.R(c).copy() for c in (0, n - 1)])
    e.close()
    return out


def same(a, b, what):
    ok = np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(a[1], b[1]) and a[2] == b[2] and all(np.array_equal(_bits(x), _bits(y)) for x, y in zip(a[3], b[3]))
    print("%-60s %s" % (what, "bit-equal" if ok else "DIFFERS"))
    return ok


def main():
    if "MCMCX_LIBRARY" not in os.environ:
        raise SystemExit("set MCMCX_LIBRARY to a library built with tools/build_variant.sh NAME -DMCX_VARIANTS")
    ok = True
    rng = np.random.default_rng(3)
    for d in (17, 33, 50, 64):                                  # pooled AM on the matrix cores: two waves per tile / half a tile per wave
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        ckw = dict(nsimu=230, adaptint=100, updatesigma=0)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.3), cmat0=(0.3 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        base = run(ckw, pkw, 140, {"MCMCX_POOLED_WAVES": "1"}, "pooled_mfma_kernel<false>", pooled=1)
        ok &= same(run(ckw, pkw, 140, {"MCMCX_POOLED_WAVES": "3"}, "pooled_mfma2_kernel", pooled=1), base, "pooled_mfma2_kernel npar %d" % d)
        ok &= same(run(ckw, pkw, 140, {"MCMCX_POOLED_WAVES": "4"}, "pooled_mfma3_kernel", pooled=1), base, "pooled_mfma3_kernel npar %d" % d)
    for d in (7, 23):                                           # delayed rejection through the general step_body<DR>
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        ckw = dict(nsimu=130, adaptint=50, updatesigma=0, drscale=2.0)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.5 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        base = run(ckw, pkw, 70, {"MCMCX_GROUP": "0", "MCMCX_DR_BIG": "0"}, "step_kernel_dr")
        ok &= same(run(ckw, pkw, 70, {"MCMCX_GROUP": "0", "MCMCX_DR_BIG": "0", "MCMCX_DR_GENERAL": "1"}, "step_kernel<false, true, false>"), base, "step_kernel<false, true, false> npar %d" % d)
    for d in (49, 100):                                         # the blocked SVD's rotations once per pair
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        ckw = dict(nsimu=45, method="scam", adaptint=14, updatesigma=0)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.3 / d) * np.eye(d), mu=np.zeros(d), lam=A @ A.T + np.diag(10.0 ** np.linspace(-1, 2, d)))
        base = run(ckw, pkw, 70, {}, None)
        ok &= same(run(ckw, pkw, 70, {"MCMCX_SVD_SHARED_ROT": "1"}, None), base, "svd_sweep_stream32s_kernel npar %d" % d)
    for d, dr in ((10, 0.0), (23, 2.0), (50, 0.0)):             # the covariance update's folds behind a per-lane FIFO (round 6)
        A = rng.standard_normal((d, d)) / np.sqrt(d)
        ckw = dict(nsimu=330, adaptint=100, updatesigma=0, drscale=dr)
        pkw = dict(kind="gauss", npar=d, par0=np.full(d, 0.05), cmat0=(0.5 / d) * np.eye(d), mu=np.linspace(-1, 1, d), lam=A @ A.T + np.eye(d))
        base = run(ckw, pkw, 200, {}, None)
        ok &= same(run(ckw, pkw, 200, {"MCMCX_COV_FIFO": "1"}, None), base, "adapt_cov_{diag,off}_fifo_kernel npar %d" % d)
    print("all bit-equal" if ok else "DIFFERENCES FOUND")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
