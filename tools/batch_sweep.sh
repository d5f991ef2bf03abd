#!/bin/bash
# throughput against sequences in flight: bench.py at several (sequences per context, contexts); usage: tools/batch_sweep.sh "B depth" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "$@"; do
  set -- $cfg
  timeout 280 python bench.py --seqs-per-gpu $1 --pipeline-depth $2 --ggs-slots $2 --no-image-features --cpu-budget-s 0 --steps ${3:-12} --warmup ${2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print(json.dumps({'B':$1,'contexts':$2,'in_flight':$1*$2,'seq_per_s':round(d['value'],1),'ms_per_step':round(d['ms_per_step'],2),'wgs_per_seq':c['ggs_workgroups_per_sequence'],'pass_latency_ms':round(c['pass_latency_ms_unpipelined'],1),'den_us':round(d['roofline_denoiser']['step_us'],1),'ggs_ms':round(d['roofline']['launch_ms'],2),'ggs_set_ms':round(d['roofline']['all_launches_ms'],2),'frac_all':round(d['roofline']['frac_all_launches'],4),'iters':c['ggs_iterations_per_sequence_run'],'finite':c['outputs_finite']}))"
done
