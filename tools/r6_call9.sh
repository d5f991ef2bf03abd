#!/bin/bash
# round 6, call 9: 32-row tiles (rt1: out-projection + FF2; rt1ff1: FF1 too) against the 96-row tiles of the product library, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python tools/den_large_ab.py posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_rt1.so gpurun_ab/libpd_rt1ff1.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|FUSED_ATTN" > gpurun_out/r6_den_rt1.txt; cat gpurun_out/r6_den_rt1.txt
