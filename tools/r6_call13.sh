#!/bin/bash
# round 6, call 19: fp16-plane activations stored as row-local hi / lo PLANES (no un-zip in the K loops) against the one-word-per-element layout (words = the previous commit):
# step alone + bitwise check, the denoiser / ViT / fused-attention tests, the headline
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_words.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_den_planes.txt; cat gpurun_out/r6_den_planes.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "bench_launch_shapes or fused_qkv or fp16_plane or first_layer or vit or adversarial or denoiser or feature_extractor or forward_api or pipeline or two_engines" 2>&1 | tail -6
for rep in 0 1; do for lib in gpurun_ab/libpd_words.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs in-pipe ms', round(r['launch_ms'],3), 'den us', round(d['roofline_denoiser']['step_us'],1), 'all ctx', round(d['roofline_denoiser']['all_contexts_step_us'],1))"
done; done > gpurun_out/r6_headline_planes.txt 2>&1; cat gpurun_out/r6_headline_planes.txt
