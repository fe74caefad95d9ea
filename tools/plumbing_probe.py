"""tools/plumbing_probe.py -- BASELINE configuration 0 through the shim: the reference's own testcases/mcmcrun.F90 (unmodified, linked
against libmcmcxf.a + libmcmcx.so: the user's Fortran ssfunction runs on the host, one chain), nsimu = 10000, wall time of the
program.  GPU box."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_fortran_shim as T
from golden_util import load
from oracle import pyoracle as oracle

z, cfg, prob = load("c1_shipped_nml", oracle)
exe = os.path.join(ROOT, "oracle", "_ref", "tc_mcmcrun")
for label, nml in (("host callbacks, 1 chain", T.NML.split("&mcmcx")[0]),):
    for nsimu in (1000, 10000):
        with tempfile.TemporaryDirectory() as d:
            T._write_inputs(d, z, nml.replace("nsimu       = 1000", "nsimu       = %d" % nsimu).replace("verbosity   = 1", "verbosity   = 0"))
            t0 = time.perf_counter()
            p = subprocess.run([exe], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            dt = time.perf_counter() - t0
            print("%-36s nsimu %6d: %.2f s (rc %d)" % (label, nsimu, dt, p.returncode), flush=True)
