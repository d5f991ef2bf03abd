#!/bin/bash
# round 5, call 9: the lane kernel with three waves per SIMD (12 waves x 168 VGPRs: every wave 38 steps at the bench shape, no SIMD carries a
# long and a short wave) against the 8-wave default, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8" timeout 600 python tools/ab_ggs.py gpurun_ab/libpd_base.so gpurun_ab/libpd_w12rv6r3l0.so gpurun_ab/libpd_w12rv6r2l1.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/ab_lane_w12.txt; cat gpurun_out/ab_lane_w12.txt
