#!/bin/bash
# round 6, call 4: zigzag stream probe; the lane-table tests in full; headline A/B (round 5's library vs the (k, d) cut rule)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 tools/stream_probe 2>&1 | grep "ZIGZAG\|RING\|FAILED\|failed" > gpurun_out/r6_zigzag_probe.txt; cat gpurun_out/r6_zigzag_probe.txt
timeout 900 python -m pytest tests -m gpu -q -k "lane or headline or ggs or guided or device_built or ingest or decode" 2>&1 | grep -v Warning | tail -60 > gpurun_out/r6_pytest4.txt; tail -40 gpurun_out/r6_pytest4.txt
for rep in 0 1; do for lib in gpurun_ab/libpd_base.so posediffusion_amd/lib/libpd_engine.so; do
  PD_ENGINE_LIB=$PWD/$lib timeout 600 python bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$lib', 'value', d['value'], 'ms/step', d['ms_per_step'], 'ggs alone ms', r.get('launch_ms'), 'den us', d.get('roofline_denoiser',{}).get('step_us'))"
done; done > gpurun_out/r6_headline_ab.txt 2>&1; cat gpurun_out/r6_headline_ab.txt
