#!/bin/bash
# tools/r06_fuzz.sh PART -- on the GPU box: the one-off fuzz harnesses on the round's final build, seeds no earlier round used (profiles/r06_g/README.md)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06g; mkdir -p $O
case "$1" in
  a) BIGFUZZ_SECONDS=560 python tools/bigfuzz.py 700000 900000 2>&1 | tee $O/bigfuzz_auto.txt | grep -E "^seed .*00 |configs|time limit|^\(" ;
     MCMCX_GROUP=0 BIGFUZZ_SECONDS=420 python tools/bigfuzz.py 900000 990000 2>&1 | tee $O/bigfuzz_lane.txt | grep -E "^seed .*00 |configs|time limit|^\(" ;;
  b) POOLED_FUZZ_SECONDS=280 python tools/pooled_restate_fuzz.py 700000 720000 2>&1 | tee $O/pooled_restate_fuzz.txt | tail -n 3
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 720000 740000 ram 2>&1 | tee $O/pooled_ram_restate_fuzz.txt | tail -n 3
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 740000 760000 scam 2>&1 | tee $O/pooled_scam_restate_fuzz.txt | tail -n 3
     MCMCX_POOLED_WAVES=2 POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 760000 780000 2>&1 | tee $O/pooled_restate_fuzz_w2.txt | tail -n 3 ;;
  c) HOST_FUZZ_SECONDS=240 python tools/host_fuzz.py 700000 720000 2>&1 | tee $O/host_fuzz_fused.txt | tail -n 3
     BIGNPAR_SECONDS=330 python tools/bignpar_fuzz.py 700000 720000 2>&1 | tee $O/bignpar_fuzz.txt | tail -n 3
     BIGNPAR_SECONDS=300 python tools/bignpar_fuzz.py 720000 740000 scam 2>&1 | tee $O/scam_npar_fuzz.txt | tail -n 3 ;;
  d) BIGFUZZ_SECONDS=1000 python tools/bigfuzz.py 1000000 1300000 2>&1 | tee $O/bigfuzz_auto_2.txt | grep -E "^seed .*000 |configs|time limit|^\(" ;;
  e) POOLED_FUZZ_SECONDS=330 python tools/pooled_restate_fuzz.py 800000 830000 2>&1 | tee $O/pooled_restate_fuzz_2.txt | tail -n 3
     MCMCX_POOLED_WAVES=2 POOLED_FUZZ_SECONDS=330 python tools/pooled_restate_fuzz.py 830000 860000 ram 2>&1 | tee $O/pooled_ram_restate_fuzz_w2.txt | tail -n 3
     BIGNPAR_SECONDS=330 python tools/bignpar_fuzz.py 800000 830000 2>&1 | tee $O/bignpar_fuzz_2.txt | tail -n 3 ;;
  f) export MCMCX_POOLED_WAVES=2 MCMCX_POOLED_KS=1 POOLED_FUZZ_NPAR=41,42,43,44,45,47,48,49,50,51,52,53,55,56,57,59,60,61,63,64    # pooled_mfma_ks_kernel
     O=gpurun_out/r06k; mkdir -p $O
     POOLED_FUZZ_SECONDS=400 python tools/pooled_restate_fuzz.py 900000 930000 2>&1 | tee $O/pooled_restate_fuzz_ks.txt | tail -n 3
     POOLED_FUZZ_SECONDS=400 python tools/pooled_restate_fuzz.py 930000 960000 ram 2>&1 | tee $O/pooled_ram_restate_fuzz_ks.txt | tail -n 3 ;;
  g) export MCMCX_POOLED_WAVES=2 MCMCX_POOLED_KS=1 POOLED_FUZZ_NPAR=49,50,51,52    # pooled_mfma_ks_kernel<true>: the fourth block through the 4 x 4 x 4 instruction
     O=gpurun_out/r06m; mkdir -p $O
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 960000 980000 2>&1 | tee $O/pooled_restate_fuzz_s3.txt | tail -n 3
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 980000 999000 ram 2>&1 | tee $O/pooled_ram_restate_fuzz_s3.txt | tail -n 3 ;;
  h) O=gpurun_out/r06n; mkdir -p $O                # the final build (tree64 in moments_kernel: every pooled tick), the engine's own kernel choices
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 1100000 1120000 2>&1 | tee $O/pooled_restate_fuzz.txt | tail -n 3
     POOLED_FUZZ_SECONDS=150 python tools/pooled_restate_fuzz.py 1120000 1140000 ram 2>&1 | tee $O/pooled_ram_restate_fuzz.txt | tail -n 3
     POOLED_FUZZ_SECONDS=200 python tools/pooled_restate_fuzz.py 1140000 1160000 scam 2>&1 | tee $O/pooled_scam_restate_fuzz.txt | tail -n 3 ;;
esac
