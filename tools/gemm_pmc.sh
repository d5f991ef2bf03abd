#!/bin/bash
# Counters of the denoiser GEMM tiles (tools/gemm_probe): MFMA pipe busy, wave stall buckets, LDS conflicts -- separate rocprofv3 --pmc passes.
# usage (GPU box): tools/gemm_pmc.sh  -> gpurun_out/gemm_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/gemm_pmc.txt; mkdir -p gpurun_out; : > $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_ACTIVE_INST[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*" | sort -u | tr '\n' ' ' >> $OUT; echo >> $OUT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"; do
  d=gpurun_out/pmc_g; rm -rf $d
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- tools/gemm_probe > /dev/null 2> gpurun_out/pmc_g.err || { echo "set [$set] failed: $(tail -2 gpurun_out/pmc_g.err | tr '\n' ' ')" >> $OUT; continue; }
  python - "$d" >> $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "pd_gemm_stream" in row["Kernel_Name"]:
            key = row["Kernel_Name"].split("pd_gemm_stream_kernel")[1][:14] + " grid " + row.get("Grid_Size", "?")
            per[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(per):
    print(k, {c: round(sum(v) / len(v)) for c, v in per[k].items()})
PY
  rm -rf $d
done
cat $OUT
