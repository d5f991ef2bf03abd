#!/bin/bash
# round 6, call 7: why the refactored bench's timed region ran at 653 sequences/s in call 6: with / without the launch-stamp readout (now a one-workgroup
# kernel into preallocated rows instead of hipMemcpyAsync into a fresh tensor); then the ViT tests (fp16-plane default)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
F="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe --no-from-images --no-rank-emulation"
for rep in 0 1; do for v in "--no-launch-stamps" "" "--dry-dist"; do
  timeout 600 python bench.py $F $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v]', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'launch_ms', round(r['launch_ms'],3), 'in_pipe', r['in_pipe'], 'alone', round(r['alone']['launch_ms'],3))"
done; done > gpurun_out/r6_stamps_ab.txt 2>&1; cat gpurun_out/r6_stamps_ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "vit or feature_extractor or forward_api" 2>&1 | grep -v Warning | tail -30 > gpurun_out/r6_pytest7.txt; tail -30 gpurun_out/r6_pytest7.txt
