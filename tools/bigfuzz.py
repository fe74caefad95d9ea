"""One-off extended fuzz (GPU box): the generators of tests/test_gpu_fuzz.py over seeds outside the test suite's range --
the base option space, method='ram' with condmax > 0, updatesigma with gamma shape < 1, the device-resident response
columns and runs cut into pieces -- device vs oracle, bit for bit.  python tools/bigfuzz.py [first_seed=300] [last_seed=2300]   (round 1: 3167 configurations, 0 failures).
The engine picks its kernels (round 4: the lane-group kernels wherever they cover a configuration); MCMCX_GROUP=1 MCMCX_GROUP_GW=4 in the
environment forces the quad form where it applies, MCMCX_GROUP=0 the lane-per-chain kernels."""
import importlib.util, sys, os, time, traceback
import numpy as np
sys.path.insert(0, os.getcwd())
spec = importlib.util.spec_from_file_location("g", "tests/test_gpu_fuzz.py"); g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
from oracle import pyoracle as po; po.build()
import torch; torch.cuda.init()
bad = []
t0 = time.time()
n = 0
A = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
for seed in range(A, B):
    for name, draw in (("base", g._draw), ("ramsvd", g._draw_ram_svd)):
        if name == "ramsvd" and seed % 4: continue
        try:
            ckw, pkw = draw(seed)
            g._check_against_oracle(po, ckw, pkw, seed); n += 1
        except Exception as e:
            bad.append((name, seed, repr(e)[:200]))
    if seed % 3 == 0:
        try:
            ckw, pkw = g._draw(40000 + seed)
            r = np.random.default_rng(41000 + seed)
            ckw.update(updatesigma=1, N0=float(r.choice([0.2, 0.5, 0.9])), S02=float(r.choice([0.0, 0.8])))
            pkw.update(sigma2=float(r.uniform(0.3, 1.5)), nobs=1)
            g._check_against_oracle(po, ckw, pkw, seed); n += 1
        except Exception as e:
            bad.append(("gamma", seed, repr(e)[:200]))
    if seed % 5 == 0:
        for fn in (g.test_random_configuration_response_columns_device_target, g.test_random_configuration_in_pieces):
            try:
                fn(po, 100000 + seed, "auto") if fn is g.test_random_configuration_in_pieces else fn(po, 100000 + seed); n += 1
            except Exception as e:
                bad.append((fn.__name__, seed, repr(e)[:200]))
    if seed % 2 == 0:                                     # npar 13..64: RAM in both regimes (group_ram_kernel / the wide panels), AM / DRAM / ER with the tile factorisation
        try:
            g.test_random_configuration_larger_npar(po, 200000 + seed, "auto"); n += 1
        except Exception as e:
            bad.append(("larger_npar", seed, repr(e)[:200]))
    if seed % 50 == 0:
        print("seed", seed, "checked", n, "failures", len(bad), "%.0f s" % (time.time() - t0), flush=True)
    if time.time() - t0 > float(os.environ.get("BIGFUZZ_SECONDS", "1500")): 
        print("time limit at seed", seed); break
print("configs checked", n, "failures", len(bad))
for b in bad[:20]: print(b)
