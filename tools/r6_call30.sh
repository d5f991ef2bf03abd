#!/bin/bash
# round 6, call 30: one engine pass per context (contexts = passes of the run) against three contexts, 256-sequence passes, at 20 steps (the driver's command: 5 passes) and 24 (6 passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() {
  timeout 900 python bench.py --pipeline-depth $1 --steps $2 --warmup 5 --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2> gpurun_out/r6_eb.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('contexts $1 steps $2:', 'value', round(d['value'],1), '| passes', c['engine_passes_in_timed_region'], '| ggs in pipe mean', round(r['in_pipe']['mean_ms'],3), 'busy', round(r['in_pipe']['busy_ms_per_launch'],3), 'overlapping', r['in_pipe']['overlapping_launches'], '| all contexts step us', round(d['roofline_denoiser']['all_contexts_step_us'],1))" || tail -3 gpurun_out/r6_eb.err
}
for cfg in "3 20" "5 20" "3 20" "5 20" "4 20" "3 24" "6 24" "3 24" "6 24"; do run $cfg; done > gpurun_out/r6_contexts.txt 2>&1; cat gpurun_out/r6_contexts.txt
