#!/bin/bash
# round 6, call 8: 96-row tiles for the 512-wide strip GEMMs of the large-batch denoiser (out-projection, FF2; rt3ff1: FF1 too) against 64-row tiles (nort3): same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_nort3.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_rt3ff1.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_den_rt3.txt; cat gpurun_out/r6_den_rt3.txt
for rep in 0 1; do for lib in gpurun_ab/libpd_nort3.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs in-pipe ms', round(r['launch_ms'],3), 'den us', round(d['roofline_denoiser']['step_us'],1), 'all ctx', round(d['roofline_denoiser']['all_contexts_step_us'],1))"
done; done > gpurun_out/r6_headline_rt3.txt 2>&1; cat gpurun_out/r6_headline_rt3.txt
timeout 600 python -m pytest tests -m gpu -q -k "bench_launch_shapes or fused_qkv or fp16_plane or first_layer" 2>&1 | tail -3
