"""tools/host_fuzz.py [first] [last] -- random configurations (tests/test_gpu_fuzz.py::_draw) with the target on the HOST (the oracle's C target
functions behind the engine's callback interface), three chains each, against the oracle bit for bit; MCMCX_HOST_MAPPED / MCMCX_HOST_FUSE in the
environment select the plumbing.  GPU box; HOST_FUZZ_SECONDS bounds it."""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
from oracle import pyoracle as po; po.build()
A = int(sys.argv[1]) if len(sys.argv) > 1 else 24
B = int(sys.argv[2]) if len(sys.argv) > 2 else 424
bad = []; t0 = time.time(); n = 0; kinds = {}
for seed in range(A, B):
    try:
        k = g._check_host_callbacks_against_oracle(po, seed); kinds[k] = kinds.get(k, 0) + 1; n += 1
    except Exception as ex:
        bad.append((seed, repr(ex)[:400]))
    if seed % 20 == 0: print("seed", seed, "checked", n, "failures", len(bad), "%.0f s" % (time.time() - t0), flush=True)
    if time.time() - t0 > float(os.environ.get("HOST_FUZZ_SECONDS", "300")): print("time limit at seed", seed); break
print("host-callback configurations checked", n, "failures", len(bad), kinds)
for b in bad[:12]: print(b)
