#!/bin/bash
# round 5, call 14: the headline with 14 against 12 register-resident steps in the lane kernel (gpurun_ab/libpd_rv12.so = the same sources with -DPD_LANE_RV=12), same box, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for lib in gpurun_ab/libpd_rv12.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_rv12.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_rv12.so posediffusion_amd/lib/libpd_engine.so; do
  echo -n "$lib: "; PD_ENGINE_LIB=$PWD/$lib timeout 300 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ggs launch alone ms', round(d['roofline']['launch_ms'],2), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1))"
done | tee gpurun_out/ab_rv14_bench.txt
