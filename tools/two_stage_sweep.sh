#!/bin/bash
# Same-box sweep of the pipe shape (round 4): whole-pass streams (every context runs its unguided half, then its guided half, on its own
# stream: the default) against the two-stage pipe of SamplingPipeline (--unguided-streams u: u streams run unguided halves, --ggs-slots s
# streams run guided halves) at engine passes of 256 / 192 / 128 sequences.  With ONE guided slot the GGS launches of different contexts never
# meet, and an engine pass of 192 sequences leaves 64 CUs to the other contexts' denoiser kernels while a launch iterates.
# usage (on the GPU box): bash tools/two_stage_sweep.sh > gpurun_out/two_stage_sweep.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe"
run() {   # label, then bench flags
  local label=$1; shift
  timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']; r = d['roofline']
print('$label:', round(d['value'], 1), 'seq/s, ms/step', round(d['ms_per_step'], 2), '| passes', c['engine_passes_in_timed_region'], 'x', c['sequences_per_engine_pass'],
      '| contexts', c['pipeline_depth'], 'guided slots', c.get('guided_slots'), 'unguided streams', c.get('unguided_streams'),
      '| ggs launch ms', round(r['launch_ms'], 2), '| den step us', round(d['roofline_denoiser']['step_us']), '| cold', round(d['cold_single_batch']['latency_ms'], 1) if d.get('cold_single_batch') else None)" || echo "$label: FAILED"
}
S=${STEPS:-24}
run "whole-pass 256 x 3 (default)"        --steps $S
run "two-stage 192, 3 ctx, 1 slot, 2 u"   --steps $S --engine-batch 192 --pipeline-depth 3 --ggs-slots 1 --unguided-streams 2
run "two-stage 192, 3 ctx, 1 slot, 1 u"   --steps $S --engine-batch 192 --pipeline-depth 3 --ggs-slots 1 --unguided-streams 1
run "two-stage 192, 4 ctx, 1 slot, 2 u"   --steps $S --engine-batch 192 --pipeline-depth 4 --ggs-slots 1 --unguided-streams 2
run "two-stage 128, 4 ctx, 1 slot, 2 u"   --steps $S --engine-batch 128 --pipeline-depth 4 --ggs-slots 1 --unguided-streams 2
run "whole-pass 192 x 3"                  --steps $S --engine-batch 192 --pipeline-depth 3
