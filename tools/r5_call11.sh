#!/bin/bash
# round 5, call 11: the final library against the previous one (lane kernel source unchanged, two-hop kernel changed) on the bench's GGS launch, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8" timeout 600 python tools/ab_ggs.py gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/ab_final_vs_base.txt; cat gpurun_out/ab_final_vs_base.txt
for lib in gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so; do
  echo "== $lib"; PD_ENGINE_LIB=$PWD/$lib timeout 300 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ggs ms', round(d['roofline']['launch_ms'],2), 'den us', round(d['roofline_denoiser']['step_us'],1))"
done | tee gpurun_out/ab_final_vs_base_bench.txt
