#!/bin/bash
# round 6, call 17: the residual tile of the EPI 2 strip GEMMs requested during the last K chunk (PD_STRIP_RES_AHEAD) against the library before it: legs, step alone + bitwise check, tests, headline A / B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
(echo "== PD_STRIP_RES_AHEAD=0"; tools/strip_legs_probe_nores; echo "== PD_STRIP_RES_AHEAD=1"; tools/strip_legs_probe) > gpurun_out/r6_strip_legs_res.txt 2>&1; grep -v "^ *$" gpurun_out/r6_strip_legs_res.txt | cut -c1-330 | head -30
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_den_res.txt; cat gpurun_out/r6_den_res.txt
timeout 900 python -m pytest tests -m gpu -q -k "bench_launch_shapes or fused_qkv or fp16_plane or first_layer or adversarial or denoiser or wide_tile" 2>&1 | tail -4
for lib in gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'value', round(d['value'],1), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1), 'ggs in pipe ms', round(d['roofline']['launch_ms'],3))"
done > gpurun_out/r6_res_ab.txt 2>&1; cat gpurun_out/r6_res_ab.txt
