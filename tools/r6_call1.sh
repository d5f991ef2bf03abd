#!/bin/bash
# round 6, call 1: lane tables with k more cuts for the spare / k longest pairs (k by the modelled pass) + jac_all on the last wave,
# against round 5's library (base) and the same sources with k = 1 only (k1): same box, alternating; phase clocks; the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8;80,1,8" timeout 900 python tools/ab_ggs.py gpurun_ab/libpd_base.so gpurun_ab/libpd_k1.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_ab_lane_balance.txt; cat gpurun_out/r6_ab_lane_balance.txt
timeout 300 python tools/lane_prof.py 256 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_lane_prof.txt; cat gpurun_out/r6_lane_prof.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r6_pytest1.txt; cat gpurun_out/r6_pytest1.txt
