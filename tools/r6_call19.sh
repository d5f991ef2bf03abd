#!/bin/bash
# round 6, call 19: pd_ggs_kernel's quaternion Jacobian (jac_all) formed on an idle wave during the pair backward instead of at the top of the iteration: A / B on the GGS launch shapes + GGS tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1200 python tools/ab_ggs.py gpurun_ab/libpd_prejac.so posediffusion_amd/lib/libpd_engine.so 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids" > gpurun_out/r6_jac_ab.txt; cat gpurun_out/r6_jac_ab.txt
timeout 1500 python -m pytest tests -m gpu -q -k "ggs or lane or guide or workgroup or invarian or sample" 2>&1 | tail -4
