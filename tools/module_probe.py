"""tools/module_probe.py -- a user's own __device__ target (mcmcx_set_target_module: three launches per iteration, the user's kernel
between the two phase kernels) against the built-in banana target (one launch per hundred iterations) over chain counts.  GPU box."""
import ctypes as C, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_user_module as T
from mcmcf90_amd import Engine, make_config

d = tempfile.mkdtemp()
src = os.path.join(d, "user_target.hip"); open(src, "w").write(T.USER_SRC)
hsaco = os.path.join(d, "user_target.hsaco")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), src, "-o", hsaco])
npar = 20
data = np.concatenate([np.linspace(0.5, 2.0, npar), [0.3]])
for nch in (64, 1024, 16384, 131072):
    for what in ("module", "builtin banana"):
        e = Engine(make_config(npar, nch, nsimu=401, updatesigma=0, adaptint=100, drscale=0.0))
        e.setpar0(np.full(npar, 0.1)); e.setcmat0(0.05 * np.eye(npar)); e.setsigma2nobs(np.full(1, 0.8), np.full(1, 15))
        if what == "module": e.set_target_module(hsaco, "user_target", data)
        else: e.set_target("banana", b=0.1)
        e.init(); e.run(201); e.sync()
        t0 = time.perf_counter(); e.run(401); e.sync(); dt = time.perf_counter() - t0
        print("%7d chains  %-15s %9.3g proposals/s  %7.1f us per iteration" % (nch, what, nch * 200 / dt, dt / 200 * 1e6), flush=True)
        e.close()
