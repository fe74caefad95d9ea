#!/bin/bash
# tools/profile_all.sh -- run on the GPU box from the repo root: every bench configuration through tools/profile_round.sh
# (kernel trace + separate PMC passes), summaries under gpurun_out/prof/<tag>/; then
#   python tools/install_profiles.py rNN_x c4_ram c4_ram_target c2_dram c3_dram c4_dram c4_pooled c5_pooled c5_pooled_fast
cd "$(dirname "$0")/.." || exit 1
bash tools/profile_round.sh c4_ram > /dev/null 2>&1;                         echo c4_ram done
bash tools/profile_round.sh c4_ram_target --start target > /dev/null 2>&1;   echo c4_ram_target done
bash tools/profile_round.sh c2_dram --workload c2 > /dev/null 2>&1;          echo c2 done
bash tools/profile_round.sh c3_dram --workload c3 > /dev/null 2>&1;          echo c3 done
bash tools/profile_round.sh c4_dram --method dram > /dev/null 2>&1;          echo c4_dram done
bash tools/profile_round.sh c4_pooled --pooled > /dev/null 2>&1;             echo c4_pooled done
bash tools/profile_round.sh c5_pooled --workload c5 > /dev/null 2>&1;        echo c5_pooled done
bash tools/profile_round.sh c5_pooled_fast --workload c5 --scam-fast > /dev/null 2>&1;   echo c5_pooled_fast done
for t in c4_ram c4_ram_target c2_dram c3_dram c4_dram c4_pooled c5_pooled c5_pooled_fast; do
  python3 - <<PY
import json
j = json.load(open("gpurun_out/prof/$t/summary.json"))
b = j.get("bench_line_kt", {})
print("$t", "value %.4g" % b.get("value", 0), "hbm B/prop %.0f (R %.0f W %.0f)" % (j.get("hbm_bytes_per_proposal", 0), j.get("hbm_read_bytes_per_proposal", 0), j.get("hbm_write_bytes_per_proposal", 0)),
      "valu/prop %.0f busy %.3f wait_any %.3f" % (j.get("valu_insts_per_proposal", 0), j.get("valu_busy", 0), j.get("wait_any", 0)), j.get("hbm_error", ""), j.get("sq_error", ""))
PY
done
