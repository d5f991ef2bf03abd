#!/bin/bash
# round 6, call 27: engine batch x contexts x passes at the driver's run length (20 steps of 64 sequences = 1 280) and at the default (24 steps = 1 536), final kernels, one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
run() {  # engine batch, depth, steps, min passes
  timeout 900 python bench.py --engine-batch $1 --pipeline-depth $2 --steps $3 --min-passes $4 --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2> gpurun_out/r6_eb.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print('engine batch $1 contexts $2 steps $3 min-passes $4:', 'value', round(d['value'],1), '| passes', c['engine_passes_in_timed_region'], 'of', c['sequences_per_engine_pass'], '| ggs launch ms per 256 sequences', round(r['launch_ms']*256/c['sequences_per_engine_pass'],3), '| denoiser step us per 5120 rows', round(d['roofline_denoiser']['step_us']*256/c['sequences_per_engine_pass'],1), '| slots equal alone', c['headline_slots_equal_alone'])" || tail -3 gpurun_out/r6_eb.err
}
for cfg in "256 3 20 2" "640 1 20 1" "640 2 20 2" "1280 1 20 1" "320 2 20 2" "256 3 20 2" "256 3 24 2" "768 1 24 1" "1536 1 24 1" "512 3 24 2" "256 3 24 2"; do run $cfg; done > gpurun_out/r6_engine_batch2.txt 2>&1; cat gpurun_out/r6_engine_batch2.txt
