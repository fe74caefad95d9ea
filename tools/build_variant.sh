#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: an experimental build of the engine next to the real one
# (tools/_build/libmcmcx_NAME.so; select it with MCMCX_LIBRARY=... for bench.py / tests)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wno-unused-value "$@" \
    -o $ROOT/tools/_build/libmcmcx_$NAME.so $ROOT/mcmcf90_amd/csrc/mcx_api.hip -L/opt/rocm/lib -lrccl -lrt -lpthread
