#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for q in default 8 default 8 2; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 600 python bench.py --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['in_pipe']; print('GPU_MAX_HW_QUEUES=$q', 'value', round(d['value'],1), 'ggs in pipe mean', round(r['mean_ms'],3), 'busy', round(r['busy_ms_per_launch'],3), 'overlapping', r['overlapping_launches'], 'all contexts', round(d['roofline_denoiser']['all_contexts_step_us'],1))"
done > gpurun_out/r6_hwq.txt 2>&1; cat gpurun_out/r6_hwq.txt
