#!/bin/bash
# Round 5, fourth GPU call: GPU suite, the two-hop GGS kernel at N = 50 with the pipelined per-frame sum (phase clocks + launch time), bench line with per_config.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c4; rm -rf $O; mkdir -p $O
timeout 300 python tools/ggs_prof_n50.py 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > $O/ggs_prof_n50.txt; cat $O/ggs_prof_n50.txt
timeout 1200 python -m pytest tests -m gpu -q -rfE --tb=short -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids" | tail -60 > $O/pytest.txt; tail -4 $O/pytest.txt
timeout 900 python bench.py --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_line.json')); r=d['roofline']; e=d['roofline_denoiser']
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms', round(r['launch_ms'],2), 'frac', round(r['frac'],3), 'den us', round(e['step_us'],1), 'all ctx', e['all_contexts_step_us'], 'slots_equal', d['config'].get('headline_slots_equal_alone'))
print('fabric', {k: v for k, v in r['fabric'].items() if ('GBps' in k or 'frac' in k or 'ratio' in k) and 'note' not in k})
for k,v in (d.get('per_config') or {}).items(): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
