#!/bin/bash
# round 6, GPU call d: the covariance update's FIFO in its flat-loop form -- adaptation parity, then the tick's kernels; the variants check
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/parity_subset.txt 2>&1; echo "parity subset rc $?"; tail -n 2 $O/parity_subset.txt
tools/tick_ab.sh r06d > $O/tick_ab.txt 2>&1; cat $O/tick_ab.txt
python - > $O/dr_choice.txt 2>&1 <<'PY'
import os, numpy as np
os.environ["MCMCX_GROUP"] = "0"
from mcmcf90_amd import engine_from_problem
for d in (2, 7, 20, 23):
    A = np.random.default_rng(d).standard_normal((d, d)) / np.sqrt(d)
    e = engine_from_problem(dict(nsimu=30, adaptint=10, updatesigma=0, drscale=2.0), dict(kind="gauss", npar=d, par0=np.zeros(d), cmat0=np.eye(d) / d, mu=np.zeros(d), lam=A @ A.T + np.eye(d)), nchains=70)
    e.init(); e.run(); print(d, e.last_kernel()); e.close()
PY
cat $O/dr_choice.txt
MCMCX_LIBRARY=$PWD/variants_build/libmcmcx_neg.so python tools/variants/check_variants.py > $O/check_variants.txt 2>&1; echo "variants rc $?"; cat $O/check_variants.txt | tail -n 20
