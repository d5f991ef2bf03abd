#!/bin/bash
# round 6: passes in flight (engine contexts / streams) with the final kernels: 2, 3 (default), 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation"
for rep in 0 1; do for d in 3 2 4; do
  timeout 600 python bench.py $F --pipeline-depth $d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('depth $d', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs in-pipe ms', round(r['launch_ms'],3), 'max', round(r['in_pipe']['max_ms'],2))"
done; done > gpurun_out/r6_depth_sweep.txt 2>&1; cat gpurun_out/r6_depth_sweep.txt
