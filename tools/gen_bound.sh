#!/bin/bash
# tools/gen_bound.sh -- on the GPU box: the round count no pooling of polar attempts can beat.  variants_build/libmcmcx_allok.so is the engine
# built with -DMCX_PROBE_ALLOK (every polar attempt accepted: NOT the reference's stream) -- c2's quads then need exactly two rounds of attempts
# for their five pairs, c3's rows of sixteen exactly one, a lane of the pooled kernel 32 attempts for its 25 pairs: the generator's cost with
# PERFECT sharing of attempts between the lanes of a wave, everything else unchanged.  In-tree library and variant alternate on the same box.
cd "$(dirname "$0")/.." || exit 1
tools/lib_ab.sh "--workload c2 --steps 10 --warmup 3" "--workload c3 --steps 6 --warmup 2" "--pooled --steps 6 --warmup 2"
