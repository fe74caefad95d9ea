#!/bin/bash
# tools/c1_profile.sh -- BASELINE configuration 0 through the shim (the reference's testcases/mcmcrun.F90, one chain, host callbacks, nsimu 10000):
# kernel trace of the program, where its 45 us per iteration go.  GPU box.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
mkdir -p gpurun_out/c1run gpurun_out/c1prof
python - <<'PY'
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gpu_fortran_shim as T
from golden_util import load
from oracle import pyoracle as oracle
z, cfg, prob = load("c1_shipped_nml", oracle)
nml = T.NML.split("&mcmcx")[0]
T._write_inputs(os.path.join(ROOT, "gpurun_out", "c1run"), z, nml.replace("nsimu       = 1000", "nsimu       = 10000").replace("verbosity   = 1", "verbosity   = 0"))
PY
cd gpurun_out/c1run && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/c1prof -o c1 -- $ROOT/oracle/_ref/tc_mcmcrun > $ROOT/gpurun_out/c1prof/run.log 2>&1
cd $ROOT && python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c1prof/**/*kernel_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
for r in rows[:14]:
    print("%-90s calls %7s avg %9.1f ns total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6)); tot += float(r["TotalDurationNs"]) / 1e6
print("sum of listed kernels %.1f ms" % tot)
PY
