#!/bin/bash
# round 6, call 5: zigzag match stream in the lane kernel -- A/B against round 5's library (base) and the same sources without zigzag (nozz);
# lane tests; headline A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8" timeout 900 python tools/ab_ggs.py gpurun_ab/libpd_base.so gpurun_ab/libpd_nozz.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_ab_zigzag.txt; cat gpurun_out/r6_ab_zigzag.txt
timeout 300 python tools/lane_prof.py 256 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_lane_prof5.txt; cat gpurun_out/r6_lane_prof5.txt
timeout 900 python -m pytest tests -m gpu -q -k "lane or headline or ggs or guided or device_built or ingest or decode" 2>&1 | grep -v Warning | tail -40 > gpurun_out/r6_pytest5.txt; tail -12 gpurun_out/r6_pytest5.txt
for rep in 0 1; do for lib in gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', 'value', d['value'], 'ms/step', d['ms_per_step'], 'ggs alone ms', r.get('launch_ms'), 'den us', d.get('roofline_denoiser',{}).get('step_us'))"
done; done > gpurun_out/r6_headline_ab5.txt 2>&1; cat gpurun_out/r6_headline_ab5.txt
