#!/bin/bash
# round 6, call 18: last K chunk of every LDS-DMA GEMM loop requests nothing (strip, fused QKV + attention, exact-fp32 DMA kernel) + residual tile a chunk ahead: step alone + bitwise check vs the library before, denoiser / ViT / exact tests, exact-mode step
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_den_lastchunk.txt; cat gpurun_out/r6_den_lastchunk.txt
timeout 1200 python -m pytest tests -m gpu -q -k "bench_launch_shapes or fused_qkv or fp16_plane or first_layer or adversarial or denoiser or wide_tile or exact or vit or smoke or sample" 2>&1 | tail -4
for lib in gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_preres.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exact_mode']; print('$lib', 'value', round(d['value'],1), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1), 'all contexts', round(d['roofline_denoiser']['all_contexts_step_us'],1), 'ggs in pipe ms', round(d['roofline']['launch_ms'],3), 'exact_mode', round(e['value'],1), 'exact step us', round(e['denoiser_step_us_alone'],1))"
done > gpurun_out/r6_lastchunk_ab.txt 2>&1; cat gpurun_out/r6_lastchunk_ab.txt
