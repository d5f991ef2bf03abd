#!/bin/bash
# Instruction counters of the GGS kernel tools/pmc_target.py B 1 launches (pd_ggs_lane_kernel since round 4; separate rocprofv3 --pmc passes):
# VALU / SALU / LDS instructions, VALU busy, LDS bank conflicts.
# usage (GPU box): tools/pmc_instr.sh [B=64]  -> gpurun_out/ggs_pmc_instruction_counters_B<B>.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
B=${1:-64}; OUT=gpurun_out/ggs_pmc_instruction_counters_B$B.txt; mkdir -p gpurun_out; : > $OUT
echo "# rocprofv3 --pmc <set> --kernel-trace -- python tools/pmc_target.py $B 1 ; GGS kernel dispatches only; per launch = $B sequences x 700 iterations" >> $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "VALUBusy SALUBusy" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=gpurun_out/pmc_i; rm -rf $d
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python tools/pmc_target.py $B 1 > /dev/null 2>&1
  python - "$d" >> $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
per = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "pd_ggs_kernel" in row["Kernel_Name"] or "pd_ggs_lane_kernel" in row["Kernel_Name"]:
            per[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in per.items():
    print(f"{k}: avg per launch {sum(v) / len(v):.6g} over {len(v)} launches")
PY
  rm -rf $d
done
python - "$OUT" "$B" <<'PY'
import re, sys
t = open(sys.argv[1]).read(); B = int(sys.argv[2])
m = re.search(r"SQ_INSTS_VALU: avg per launch ([0-9.e+]+)", t)
if m:
    open(sys.argv[1], "a").write(f"# VALU wave-instructions per sequence and iteration (all phases): {float(m.group(1)) / (B * 700):.0f}\n")
PY
cat $OUT
