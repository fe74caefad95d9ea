"""tools/bignpar_fuzz.py [first] [last] [scam] -- tests/test_gpu_fuzz.py::_check_larger_npar at npar 65..300: RAM (update and downdate sweeps over up to
45 150 elements), AM / DRAM / ER with the lane kernels' global-scratch forms, SVD factors through the blocked Jacobi up to npar 130; 66 chains
(a ragged tile), two of them against the oracle bit for bit.  GPU box; BIGNPAR_SECONDS bounds it."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
from oracle import pyoracle as po; po.build()
A = int(sys.argv[1]) if len(sys.argv) > 1 else 500016
B = int(sys.argv[2]) if len(sys.argv) > 2 else 500416
SCAM = len(sys.argv) > 3 and sys.argv[3] == "scam"     # third argument "scam": _check_scam_npar (npar 13..120) instead
bad = []; t0 = time.time(); n = 0
for seed in range(A, B):
    try:
        (g._check_scam_npar(po, seed) if SCAM else g._check_larger_npar(po, seed, 65, 301)); n += 1
    except Exception as ex:
        bad.append((seed, repr(ex)[:500]))
    if seed % 10 == 0: print("seed", seed, "checked", n, "failures", len(bad), "%.0f s" % (time.time() - t0), flush=True)
    if time.time() - t0 > float(os.environ.get("BIGNPAR_SECONDS", "300")): print("time limit at seed", seed); break
print("scam npar 13..120" if SCAM else "npar 65..300", "configurations checked", n, "failures", len(bad))
for b in bad[:12]: print(b)
