#!/bin/bash
# tools/profile_round.sh TAG [bench.py arguments...]   -- run on the GPU box from the repo root.
# Profiles one bench.py configuration into gpurun_out/prof/TAG/:
#   kt/     rocprofv3 --kernel-trace --stats                          -> per-kernel time
#   fetch/  rocprofv3 --pmc FETCH_SIZE        (own pass, no tracing)  -> HBM read bytes per launch (x2 on gfx950, see the guide)
#   write/  rocprofv3 --pmc WRITE_SIZE        (own pass)              -> HBM write bytes per launch
#   sq/     rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES
#   grbm/   rocprofv3 --pmc GRBM_GUI_ACTIVE                           -> effective clock
# Summaries: python tools/profile_summary.py gpurun_out/prof/TAG
TAG=$1; shift
REPO=$PWD
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_kt.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_grbm.json 2> $OUT/grbm.err
cd $REPO
python3 tools/profile_summary.py $OUT > $OUT/summary.json 2> $OUT/summary.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.csv" -size +1M -delete
find $OUT -name "*.db" -delete
cat $OUT/summary.json | head -60
