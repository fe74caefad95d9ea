"""tools/pooled_restate_fuzz.py [first] [last] [ram|scam] -- random POOLED configurations (tests/test_gpu_pooled.py::_draw_pooled: AM / delayed rejection /
early rejection; burn-in scaling, greedy, AP window, adaptend, initcmatn; sigma2 update, bounds, priors; Gaussian and banana targets; npar 2..64;
ragged tiles; the run cut in two calls) against the tick-by-tick restatement built from single-chain oracles: states, accept sequences, the shared
factor, mean and weight, bit for bit.  The engine picks its kernels.  GPU box; POOLED_FUZZ_SECONDS bounds it."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("tp", os.path.join(ROOT, "tests", "test_gpu_pooled.py")); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from oracle import pyoracle as po; po.build()

A = int(sys.argv[1]) if len(sys.argv) > 1 else 48
B = int(sys.argv[2]) if len(sys.argv) > 2 else 448
RAM = len(sys.argv) > 3 and sys.argv[3] == "ram"       # third argument "ram": pooled RAM draws (_draw_pooled_ram) instead of AM / DR / ER
SCAM = len(sys.argv) > 3 and sys.argv[3] == "scam"     # ... "scam": pooled SCAM draws
check = tp._check_pooled_ram_against_restatement if RAM else tp._check_pooled_scam_against_restatement if SCAM else tp._check_pooled_against_restatement
bad = []; t0 = time.time(); n = 0; kernels = {}
for seed in range(A, B):
    try:
        k = check(po, seed); kernels[k] = kernels.get(k, 0) + 1; n += 1
    except Exception as ex:
        bad.append((seed, repr(ex)[:400]))
    if seed % 50 == 0: print("seed", seed, "checked", n, "failures", len(bad), "%.0f s" % (time.time() - t0), flush=True)
    if time.time() - t0 > float(os.environ.get("POOLED_FUZZ_SECONDS", "600")): print("time limit at seed", seed); break
print("pooled configurations checked", n, "failures", len(bad), "kernels", kernels)
for b in bad[:12]: print(b)
