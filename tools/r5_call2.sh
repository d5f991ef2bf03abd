#!/bin/bash
# Round 5, second GPU call: the GPU suite after the fixes, the LDS-DMA ring rows of the stream probe, the in-kernel latency legs of the B = 1 denoiser
# chain, what bounds pd_qkv_attn_kernel (BARE variants), the phase clocks of the two-hop GGS kernel at N = 50, small-batch bitwise check against
# round 4's library, and the full default bench line (per_config with dropin_sample_ms).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c2; rm -rf $O; mkdir -p $O
NEW=$R/posediffusion_amd/lib/libpd_engine.so; OLD=$R/gpurun_ab/libpd_r4.so
timeout 1200 python -m pytest tests -m gpu -q -rfE --tb=short -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids" | tail -80 > $O/pytest.txt; tail -4 $O/pytest.txt
timeout 200 tools/stream_probe > $O/stream_probe.txt 2>&1; tail -8 $O/stream_probe.txt
PD_ENGINE_LIB=$R/gpurun_ab/libpd_stamps.so timeout 200 python tools/den_small_legs.py 1 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > $O/den_small_legs.txt; tail -12 $O/den_small_legs.txt
for b in 0 1 2 3 4 5; do
  echo "PD_QA_BARE=$b" >> $O/qa_bare.txt
  PD_ENGINE_LIB=$R/gpurun_ab/libpd_knobs.so PD_QA_BARE=$b timeout 120 python tools/den_large.py 256 2>&1 | grep "denoiser step" | tail -1 >> $O/qa_bare.txt
done
cat $O/qa_bare.txt
timeout 300 python tools/ggs_prof_n50.py 2>&1 | grep "us/it" > $O/ggs_prof_n50.txt; cat $O/ggs_prof_n50.txt
timeout 200 python tools/den_ab.py $OLD $NEW 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > $O/den_small_ab.txt; cat $O/den_small_ab.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_line.json')); r=d['roofline']; e=d['roofline_denoiser']
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms', round(r['launch_ms'],2), 'frac', round(r['frac'],3), 'den us', round(e['step_us'],1), 'all ctx', e['all_contexts_step_us'], 'slots_equal', d['config'].get('headline_slots_equal_alone'))
print('fabric', {k: v for k, v in r['fabric'].items() if 'GBps' in k or 'frac' in k or 'exceeded' in k})
for k,v in (d.get('per_config') or {}).items(): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
print('exact', d.get('exact_mode',{}).get('value'), 'fresh', d.get('fresh_inputs',{}).get('value'), 'cold', d.get('cold_single_batch',{}).get('latency_ms'), 'cpu', d['cpu_baseline']['value'])"
