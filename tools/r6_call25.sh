#!/bin/bash
# round 6, call 25: GGS launches of more than 256 workgroups against the sequence alone (bench.py --engine-batch 768 --pipeline-depth 1 reported a mismatch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for B in 256 512 768 1024; do timeout 900 python tools/big_launch_check.py $B 3 2>&1 | grep -v "Warn\|Transformer\|amdgpu.ids"; done > gpurun_out/r6_big_launch.txt 2>&1; cat gpurun_out/r6_big_launch.txt
