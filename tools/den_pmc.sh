#!/bin/bash
# Counters of the large-batch denoiser's kernels (round 5): effective shader clock (GRBM_GUI_ACTIVE / duration), MFMA-pipe busy, wave stall buckets -- separate
# rocprofv3 --pmc passes over tools/den_large.py 256 with the development library (PD_QA_BARE selects what pd_qkv_attn_kernel leaves out: 0 full, 3 no memory
# traffic in its K loop, 4 no MFMAs).  usage (GPU box): bash tools/den_pmc.sh -> gpurun_out/den_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
OUT=$R/gpurun_out/den_pmc.txt; mkdir -p $R/gpurun_out; : > $OUT
for bare in 0 3 4; do
 for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  d=$R/gpurun_out/pmc_d; rm -rf $d
  PD_ENGINE_LIB=$R/gpurun_ab/libpd_knobs.so PD_QA_BARE=$bare timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $R/tools/den_large.py 256 > /dev/null 2> $R/gpurun_out/pmc_d.err || { echo "BARE $bare set [$set] failed: $(tail -2 $R/gpurun_out/pmc_d.err | tr '\n' ' ')" >> $OUT; continue; }
  python - "$d" "$bare" >> $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, bare = sys.argv[1], sys.argv[2]
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(per):
    if not any(t in k for t in ("pd_qkv_attn", "pd_gemm_strip", "pd_ln_rows", "pd_gemm_dma")):
        continue
    n = len(dur.get(k, []))
    us = sum(dur[k]) / n if n else float("nan")
    c = {name: sum(v) / len(v) for name, v in per[k].items()}
    extra = f"; effective clock {c['GRBM_GUI_ACTIVE'] / us / 1e3:.2f} GHz" if "GRBM_GUI_ACTIVE" in c and n else ""
    short = k.split("(")[0][:60]
    print(f"BARE={bare} {short:60s} launches {n:4d} avg {us:7.1f} us (under the counters) " + " ".join(f"{a}={b:.3g}" for a, b in sorted(c.items())) + extra)
PY
  rm -rf $d
 done
done
cat $OUT
