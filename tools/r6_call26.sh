#!/bin/bash
# round 6, call 26: the more-workgroups-than-CUs test; bench.py with one context and a 768-sequence engine batch (its slot check now runs the single sequence on the kernel family of the big launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_r6.py -m gpu -q 2>&1 | tail -3
timeout 900 python bench.py --engine-batch 768 --pipeline-depth 1 --steps 24 --no-per-config --no-fresh-inputs --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation --no-exact-mode 2> gpurun_out/r6_eb.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('engine batch 768 depth 1 steps 24:', 'value', round(d['value'],1), 'ggs launch ms', round(r['launch_ms'],3), 'per 256 sequences', round(r['launch_ms']/3,3), 'denoiser step us', round(d['roofline_denoiser']['step_us'],1), 'slots equal alone', d['config']['headline_slots_equal_alone'])" || tail -5 gpurun_out/r6_eb.err
