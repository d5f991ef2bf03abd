#!/bin/bash
# round 6, call 13: fragment reads of the strip kernel issued one 16-k step ahead by hand (PD_STRIP_PIPE) against the same sources without it (nopipe): legs, step alone + bitwise check, tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
tools/strip_legs_probe > gpurun_out/r6_strip_legs_pipe.txt 2>&1; head -12 gpurun_out/r6_strip_legs_3buf.txt | cut -c1-400
timeout 900 python tools/den_large_ab.py gpurun_ab/libpd_nopipe.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_den_pipe.txt; cat gpurun_out/r6_den_3buf.txt
timeout 900 python -m pytest tests -m gpu -q -k "bench_launch_shapes or fused_qkv or fp16_plane or first_layer or vit or adversarial or denoiser" 2>&1 | tail -4
