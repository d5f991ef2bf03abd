#!/bin/bash
# round 5, call 8: the lane-item cut rule that balances the SIMDs (pd_lane_extra): same-box A / B against the previous library on the GGS launch
# shapes, then every GPU test that touches the lane kernel or its tables
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_AB_SHAPES="64,1,8;256,1,8;8,1,8" timeout 300 python tools/ab_ggs.py gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so > gpurun_out/ab_lane_cuts.txt 2>&1; tail -30 gpurun_out/ab_lane_cuts.txt
timeout 200 python tools/lane_prof.py 256 > gpurun_out/lane_prof_cuts.txt 2>&1; cat gpurun_out/lane_prof_cuts.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/pytest_cuts.txt
