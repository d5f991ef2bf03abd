#!/bin/bash
# End-of-round evidence with the final binary (on the GPU box): full GPU suite, smoke, counters, the default bench line and
# a kernel trace of the same command.  Results under gpurun_out/final/ -> copy into profiles/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final/smoke.txt
bash tools/collect_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json gpurun_out/final/pmc_summary.json
cp gpurun_out/pmc_summary.json profiles/round1_pmc_summary.json      # bench.py reads the traffic figures from here
timeout 600 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/trace -o bench -- python $R/bench.py --no-image-features --cpu-budget-s 0 > $R/gpurun_out/final/bench_traced.json 2>/dev/null
cd $R; python tools/rocpd_stats.py gpurun_out/final/trace/bench_results.db 16 > gpurun_out/final/kernel_stats.txt; rm -rf gpurun_out/final/trace
python -c "
import json; d=json.load(open('gpurun_out/final/bench_line.json')); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'ggs launch ms', r['launch_ms'], 'set', r['all_launches_ms'], 'frac', r['frac'], 'frac_all', r['frac_all_launches'], 'traffic', r['traffic'])
print('from_images', d.get('from_images', {}).get('value'), 'cpu', d['cpu_baseline']['value'], 'den', d['roofline_denoiser'])"
head -4 gpurun_out/final/kernel_stats.txt
