#!/bin/bash
# round 6, call 24: bigger engine passes with the final kernels (a GGS launch of 512 / 768 workgroups back-fills the CUs as workgroups finish: one tail per launch instead of one per 256 sequences)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for cfg in "256 3 24" "512 2 32" "768 1 24" "768 2 48" "512 3 48" "256 3 24"; do
  set -- $cfg
  timeout 900 python bench.py --engine-batch $1 --pipeline-depth $2 --steps $3 --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2> gpurun_out/r6_eb.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('engine batch $1 depth $2 steps $3:', 'value', round(d['value'],1), 'ggs launch ms', round(r['launch_ms'],3), 'per 256 sequences', round(r['launch_ms']*256/$1,3), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1), 'passes', d['config']['engine_passes_in_timed_region'])" || tail -3 gpurun_out/r6_eb.err
done > gpurun_out/r6_engine_batch.txt 2>&1; cat gpurun_out/r6_engine_batch.txt
