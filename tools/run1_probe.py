"""tools/run1_probe.py -- dev-container aid: drives the REAL reference's mcmc_main_one (oracle/_ref/mcxref_one) a few
invocations in a scratch directory and prints what each leaves behind (MCMC_run1.F90 / MCMC_run1_er.F90)."""
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import pyoracle as po, refrun

method = sys.argv[1] if len(sys.argv) > 1 else "dram"
drscale = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ninv = int(sys.argv[3]) if len(sys.argv) > 3 else 6
d = 3
rng = np.random.default_rng(1)
A = rng.standard_normal((d, d)); lam = A @ A.T + d * np.eye(d)
cfg = po.make_cfg(nsimu=12, drscale=drscale, adaptint=100, updatesigma=0, method=method)
prob = po.Problem(kind="gauss", npar=d, par0=np.array([0.1, -0.2, 0.3]), cmat0=0.05 * np.eye(d), mu=np.zeros(d), lam=lam,
                  lo=np.array([-0.3, -np.inf, -np.inf]))
w = tempfile.mkdtemp(prefix="run1_")
refrun.write_inputs(w, cfg, prob, extra_nml=" verbosity = 1\n")
open(os.path.join(w, "mcmcrun.nml"), "w").write("&mcmcrun\n drstage=1, isimu=1, ieval=0, nrej=0, alpha12=0.0, sscrit=-1.0\n/\n")
exe = os.path.join(os.path.dirname(refrun.EXE), "mcxref_one")
for k in range(ninv):
    env = dict(os.environ, MCX_SEED=str(1000 + k), MCX_CHAIN="0", MKL_NUM_THREADS="1")
    p = subprocess.run([exe], env=env, cwd=w, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    print("=== invocation", k, "rc", p.returncode)
    print(p.stdout.decode()[-1800:])
    for f in sorted(os.listdir(w)):
        if f.endswith(".dat") or f == "mcmcrun.nml":
            print("  ", f, ":", open(os.path.join(w, f)).read().strip().replace("\n", " | ")[:300])
    print("  files:", sorted(os.listdir(w)))
    shutil.copy(os.path.join(w, "mcmcparnew.dat"), os.path.join(w, "mcmcpar.dat"))
shutil.rmtree(w)
