#!/bin/bash
# round 6, call 21: fused QKV + attention kernel, block -> (group, head) mapped onto the XCDs (PD_QA_XCD_MAP 0 = neighbours, 1 = head pairs per XCD half, 2 = all heads of a group on one XCD):
# step alone + bitwise check, FETCH_SIZE of the kernel, three contexts / headline
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out
timeout 1200 python tools/den_large_ab.py gpurun_ab/libpd_map0.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_map2.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_qa_map.txt; cat gpurun_out/r6_qa_map.txt
for lib in gpurun_ab/libpd_map0.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_map2.so; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_q; rm -rf $d
  PD_ENGINE_LIB=$R/$lib timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -- python $R/tools/den_large.py 256 > /dev/null 2> $R/gpurun_out/pmc_q.err || { echo "$lib $ctr failed"; continue; }
  python - "$d" "$lib" "$ctr" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, lib, ctr = sys.argv[1:4]
per = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == ctr: per[row["Kernel_Name"].split("(")[0][:48]].append(float(row["Counter_Value"]))
print(lib, ctr, "KB per dispatch:", {k: round(sum(v) / len(v)) for k, v in per.items() if any(t in k for t in ("qkv_attn", "gemm_strip", "ln_rows"))})
PY
  done
done > gpurun_out/r6_qa_map_pmc.txt 2>&1; cat gpurun_out/r6_qa_map_pmc.txt
for lib in gpurun_ab/libpd_map0.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_map2.so gpurun_ab/libpd_map0.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_map2.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'value', round(d['value'],1), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1), 'all contexts', round(d['roofline_denoiser']['all_contexts_step_us'],1), 'ggs in pipe ms', round(d['roofline']['launch_ms'],3))"
done > gpurun_out/r6_qa_map_ab.txt 2>&1; cat gpurun_out/r6_qa_map_ab.txt
