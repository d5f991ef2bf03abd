#!/bin/bash
# round 5, call 10: the two-hop GGS kernel's serial phases at N = 50 (totals on three waves, dL/dA sums only where the stage moves the focal
# lengths, decode of what the stage moves): phase clocks, same-box A / B against the previous library, GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/c10; rm -rf $O; mkdir -p $O
for lib in gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so; do
  echo "== $lib"; PD_ENGINE_LIB=$PWD/$lib timeout 300 python tools/ggs_prof_n50.py 2>&1 | grep -v "Warning\|TransformerEncoder\|amdgpu.ids\|wgs_per_seq=64\|^  .*64 ->" | head -3
done > $O/ggs_prof_n50.txt; cat $O/ggs_prof_n50.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
