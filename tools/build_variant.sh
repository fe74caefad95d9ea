#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: an experimental build of the engine next to the real one
# (variants_build/libmcmcx_NAME.so, which travels to the GPU box; -DMCX_VARIANTS adds tools/variants; select it with MCMCX_LIBRARY=... for bench.py / tests)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/variants_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value -Wno-cuda-compat "$@" \
    -o $ROOT/variants_build/libmcmcx_$NAME.so $ROOT/mcmcf90_amd/csrc/mcx_api.hip -L/opt/rocm/lib -lrccl -lrt -lpthread
