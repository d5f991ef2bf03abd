#!/bin/bash
# sweep of (contexts in flight, guided slots, GGS workgroups/sequence) for bench.py; one summary line each
# usage: tools/pipeline_sweep.sh depth,slots,wgs ...   (Qn = set GPU_MAX_HW_QUEUES=n for the following runs)
out=gpurun_out/pipeline_sweep.txt
mkdir -p gpurun_out
: > $out
for cfg in "$@"; do
  if [[ $cfg == Q* ]]; then export GPU_MAX_HW_QUEUES=${cfg#Q}; echo "## GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES" >> $out; continue; fi
  IFS=, read d s w pr nu dw <<< "$cfg"; pr=${pr:-0}
  echo "== depth=$d slots=$s wgs=$w ustreams=${nu:-2} den_wgs=${dw:-0}" >> $out
  timeout 300 python bench.py --steps 24 --warmup $d --pipeline-depth $d --ggs-slots $s --ggs-wgs $w --unguided-streams ${nu:-2} --denoiser-wgs-per-xcd ${dw:-0} --cpu-budget-s 0 2>&1 | tail -1 | \
    python -c "import sys,json; l=sys.stdin.read().strip(); 
try:
    j=json.loads(l); print('%.1f seq/s %.2f ms/pass; lat %.1f; ggs %.3f ms (%.2f us/it); den %.1f us; iters %s finite %s' % (j['value'], j['ms_per_step'], j['config']['pass_latency_ms_unpipelined'], j['per_step_ms']['ggs_guided_step'], j['per_step_ms']['ggs_iteration_us'], j['per_step_ms']['denoiser_step']*1e3, j['config']['ggs_iterations_per_sequence_run'], j['config']['outputs_finite']))
except Exception as e: print('FAILED', l[-300:])" >> $out
done
cat $out
