#!/bin/bash
# Round 5, sixth GPU call: pd_gemm_big_kernel -- bitwise test, step alone per mask, the pipe with and without it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c6; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_r5.py -m gpu -q -rfE --tb=short -k "fused or hoist" 2>&1 | grep -v "Warning\|warnings.warn\|^$\|amdgpu.ids" | tail -30 > $O/pytest.txt; tail -6 $O/pytest.txt
timeout 300 python tools/den_large_ab.py $R/posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" | head -24 > $O/den_large_ab.txt; cat $O/den_large_ab.txt
Q="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe"
for m in 0 7 2 0 7; do
  PD_BENCH_BIG_GEMM=$m timeout 300 python bench.py $Q 2>$O/bench_$m.err | tail -1 > $O/bench_$m.json
  python -c "
import json; d=json.load(open('$O/bench_$m.json')); r=d['roofline']; e=d['roofline_denoiser']
print('mask $m', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms', round(r['launch_ms'],2), 'den us', round(e['step_us'],1), 'all ctx', round(e['all_contexts_step_us'],1), 'slots_equal', d['config'].get('headline_slots_equal_alone'))" | tee -a $O/bench_ab.txt
done
