#!/bin/bash
# Profiles of the headline workload (run on the GPU box from the repo root; writes gpurun_out/prof_*):
#   1. rocprofv3 --kernel-trace --stats                      -> per-kernel time
#   2. rocprofv3 --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE        -> HBM bytes per launch (separate passes, no tracing)
# Summaries: python tools/profile_summary.py gpurun_out
REPO=$PWD
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_kt_bench.json 2> $OUT/prof_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch_bench.json 2> $OUT/prof_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_write_bench.json 2> $OUT/prof_write.err
cd $REPO
python3 tools/profile_summary.py $OUT > $OUT/prof_summary.json 2> $OUT/prof_summary.err
find $OUT/prof_kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
