#!/bin/bash
# tools/final_check.sh -- on the GPU box: the whole GPU suite, the default bench.py (driver style), smoke()
mkdir -p gpurun_out
MCMCX_SUITE_BUDGET_STRICT=1 timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; rc=$?; echo suite rc $rc; tail -3 gpurun_out/gpu_suite.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; rc=$?; echo bench rc $rc
[ $rc -eq 0 ] || { tail -20 gpurun_out/bench_default.err; exit $rc; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_default.json"))
print(j["value"], j["roofline"]["frac"], j["roofline"]["traffic"], j["roofline"]["kernel"])
print({k: (round(v["value"] / 1e6, 1), v["kernel"]) for k, v in j["other_configs"].items()})
print(json.dumps(j["cpu_baseline"])[:700])
PY
python -c "import __graft_entry__ as g; g.smoke()"
# the `extended` set (non-production duplicates and the slow lane-SVD cases above npar 256: ADVICE round 5 -- a release always has them)
MCMCX_EXTENDED=1 timeout -k 10 700 python -m pytest tests -m "gpu and extended" -x -q > gpurun_out/gpu_extended.log 2>&1; echo extended rc $?; tail -3 gpurun_out/gpu_extended.log
