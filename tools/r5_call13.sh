#!/bin/bash
# round 5, call 13: lane kernel with 14 register-resident steps (no spill since the P3b thread roles are formed inside the iteration) against 12, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8" timeout 600 python tools/ab_ggs.py gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/ab_lane_rv14.txt; cat gpurun_out/ab_lane_rv14.txt
