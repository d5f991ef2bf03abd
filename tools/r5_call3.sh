#!/bin/bash
# Round 5, third GPU call: GPU suite with the SCC-clobber fix and the seeds-test bound, the fixed stream probe (LDS-DMA ring rows), the L2 carry probe,
# phase clocks of the two-hop GGS kernel at N = 50, the fused attention kernel with its weight fragments a chunk ahead (DEEP) against half a chunk.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c3; rm -rf $O; mkdir -p $O
NEW=$R/posediffusion_amd/lib/libpd_engine.so; OLD=$R/gpurun_ab/libpd_r4.so; KN=$R/gpurun_ab/libpd_knobs.so
timeout 60 tools/stream_probe > $O/stream_probe.txt 2>&1; tail -7 $O/stream_probe.txt
timeout 60 tools/l2_carry_probe > $O/l2_carry_probe.txt 2>&1; cat $O/l2_carry_probe.txt
for d in 0 1 0 1; do
  echo "PD_QA_DEEP=$d" >> $O/qa_deep.txt
  PD_ENGINE_LIB=$KN PD_QA_DEEP=$d timeout 120 python tools/den_large.py 256 2>&1 | grep "denoiser step" | tail -1 >> $O/qa_deep.txt
done
cat $O/qa_deep.txt
timeout 300 python tools/den_large_ab.py $OLD $NEW 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > $O/den_large_ab.txt; cat $O/den_large_ab.txt
timeout 300 python tools/ggs_prof_n50.py 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > $O/ggs_prof_n50.txt; cat $O/ggs_prof_n50.txt
timeout 1200 python -m pytest tests -m gpu -q -rfE --tb=short -s 2>&1 | grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids" | tail -60 > $O/pytest.txt; tail -4 $O/pytest.txt
timeout 600 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench_line.json')); r=d['roofline']; e=d['roofline_denoiser']
print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms', round(r['launch_ms'],2), 'frac', round(r['frac'],3), 'den us', round(e['step_us'],1), 'all ctx', e['all_contexts_step_us'], 'slots_equal', d['config'].get('headline_slots_equal_alone'))
print('fabric', {k: v for k, v in r['fabric'].items() if ('GBps' in k or 'frac' in k or 'exceeded' in k) and 'note' not in k})"
