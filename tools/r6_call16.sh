#!/bin/bash
# round 6: the exact-fp32 LDS-DMA GEMM kernel's epilogue with 16-byte accesses (exact mode's encoder GEMMs; _first / _last.0 of the default mode): step alone + bitwise check, tests, exact-mode headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_prewide.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_den_dma_wide.txt; cat gpurun_out/r6_den_dma_wide.txt
timeout 900 python -m pytest tests -m gpu -q -k "bench_launch_shapes or fp16_plane or first_layer or denoiser or exact or wide_tile" 2>&1 | tail -4
for lib in gpurun_ab/libpd_prewide.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_prewide.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exact_mode']; print('$lib', 'value', round(d['value'],1), 'exact_mode', round(e['value'],1), 'exact step us', round(e['denoiser_step_us_alone'],1), 'default step us', round(e['denoiser_step_us_alone_default_mode'],1))"
done > gpurun_out/r6_exact_ab.txt 2>&1; cat gpurun_out/r6_exact_ab.txt
