#!/bin/bash
# round 6, call 6: the refactored bench.py (legs in bench_legs.py; roofline.frac from the timed region's own launches) end to end + the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
T0=$SECONDS; timeout 1500 python bench.py --dry-dist --cpu-budget-s 10 > gpurun_out/r6_bench6.json 2> gpurun_out/r6_bench6.err; echo "rc $? wall $((SECONDS - T0)) s"; tail -5 gpurun_out/r6_bench6.err
python -c "
import json; d=json.load(open('gpurun_out/r6_bench6.json')); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'launch_ms', r['launch_ms'], 'in_pipe', r['in_pipe'], 'alone', r['alone']['launch_ms'], r['alone']['frac'])
print('den', d['roofline_denoiser']['step_us'], d['roofline_denoiser']['frac'])
print('ranks', json.dumps(d.get('rank_emulation'))[:1500])
print('images', json.dumps(d.get('from_images'))[:1200])
print('cpu', json.dumps(d.get('cpu_baseline'))[:900])
print('exact', d.get('exact_mode',{}).get('value'), 'fresh', d.get('fresh_inputs',{}).get('value'), 'slots', d['config']['headline_slots_equal_alone'])
for k,v in (d.get('per_config') or {}).items(): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -25 > gpurun_out/r6_pytest6.txt; tail -25 gpurun_out/r6_pytest6.txt
