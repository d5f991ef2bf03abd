#!/bin/bash
# round 6, call 2 (and 3: + the (k, d) rule that puts the long | short boundary on a wave boundary): + masked steps two at a time (mixed waves) -- A/B against round 5's library, phase clocks
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="256,1,8;64,1,8" timeout 900 python tools/ab_ggs.py gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_ab_lane_balance2.txt; cat gpurun_out/r6_ab_lane_balance2.txt
timeout 300 python tools/lane_prof.py 256 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_lane_prof2.txt; cat gpurun_out/r6_lane_prof2.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "lane or headline or ggs or guided or device_built or ingest" 2>&1 | tail -5 > gpurun_out/r6_pytest2.txt; cat gpurun_out/r6_pytest2.txt
