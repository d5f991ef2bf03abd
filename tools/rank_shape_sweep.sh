#!/bin/bash
# One rank of the driver's 8- / 4-GPU strong-scaling run (--steps 20: 160 / 320 sequences per rank) emulated on one GPU, by pass shape:
# the default cuts a rank's run into at least TWO passes (a second context overlaps the first one's GGS launches); --min-passes 1 runs the
# 160 sequences of an 8-GPU rank as ONE pass (160 GGS workgroups per launch instead of 2 x 80 side by side, denoiser steps of 3 200 rows
# instead of 2 x 1 600), --min-passes 3 / 4 as shorter passes.  NOT YET MEASURED with the lane kernel (round 3, wave kernels: 475 as one pass
# of 160 against 465 as 2 x 80).  usage (on the GPU box): bash tools/rank_shape_sweep.sh > gpurun_out/rank_shape_sweep.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
Q="--scaling weak --steps 20 --warmup 5 --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe"
run() {
  local label=$1; shift
  timeout 300 python bench.py $Q "$@" 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); c = d['config']; r = d['roofline']
print('$label:', round(d['value'], 1), 'seq/s per rank, ms/step', round(d['ms_per_step'], 2), '| passes', c['engine_passes_in_timed_region'], 'x', c['sequences_per_engine_pass'],
      '| contexts', c['pipeline_depth'], '| ggs wgs', c['ggs_workgroups_per_sequence'], '| ggs launch ms', round(r['launch_ms'], 2), '| pass latency ms', round(c['engine_pass_latency_ms_unpipelined'], 1))" || echo "$label: FAILED"
}
run "8-GPU rank, 2 x 80 (default)"      --seqs-per-step 8
run "8-GPU rank, 1 x 160"               --seqs-per-step 8 --min-passes 1
run "8-GPU rank, 3 passes (56/56/48)"   --seqs-per-step 8 --min-passes 3
run "8-GPU rank, 4 x 40"                --seqs-per-step 8 --min-passes 4 --pipeline-depth 4
