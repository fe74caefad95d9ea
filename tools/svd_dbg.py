"""dev aid: fuzz seed -> where the engine's SVD factor first parts from the oracle's (python tools/svd_dbg.py SEED)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from oracle import pyoracle as po
from mcmcf90_amd import engine_from_problem
import test_gpu_fuzz as tf
seed = int(sys.argv[1])
ckw, pkw = tf._draw(seed)
print(ckw); print({k: (v if np.size(v) < 8 else "...") for k, v in pkw.items()})
cfg = po.make_cfg(**ckw); prob = po.Problem(**pkw)
e = engine_from_problem(ckw, pkw, nchains=67, chain_id0=3 * seed, record_accept=1, record_chain=1)
e.init()
b = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
for upto in list(range(1, ckw["nsimu"] + 1)):
    e.run(upto); e.sync()
    bad = False
    for c in (0, 1, 66):
        o = po.run_chain(cfg, prob, chain_id=3 * seed + c, upto=upto, continue_on_downdate_fail=True)
        R = e.R(c)
        th = e.theta()[c]
        if not np.array_equal(b(R), b(o.R)) or not np.array_equal(b(th), b(o.theta)):
            print("upto", upto, "chain", c, "R equal", np.array_equal(b(R), b(o.R)), "theta equal", np.array_equal(b(th), b(o.theta)))
            print(" engine R", R.ravel(), "\n oracle R", o.R.ravel(), "\n diff ulps", (b(R).astype(np.int64) - b(o.R).astype(np.int64)).ravel())
            cm, mean, w = e.chaincov(c)
            print(" chaincmat equal", np.array_equal(b(cm), b(o.chaincmat)), cm.ravel(), o.chaincmat.ravel())
            bad = True
    if bad:
        break
else:
    print("no difference in chains 0, 1, 66")
e.close()
