#!/bin/bash
# The trace-derived part of tools/round_end.sh alone (kernel statistics of the bench command + the co-resident analysis), e.g. after
# tools/rocpd_stats.py / tools/coresident_from_trace.py changed.  Results under gpurun_out/final/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; F=gpurun_out/final; mkdir -p $F
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$F/trace -o bench -- python $R/bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 > $R/$F/bench_traced.json 2>/dev/null
cd $R; python tools/rocpd_stats.py $F/trace/bench_results.db 16 > $F/kernel_stats.txt
python tools/coresident_from_trace.py $F/trace/bench_results.db $(python -c "import json; print(json.load(open('$F/bench_traced.json'))['roofline']['algorithmic_flops_per_launch'])") > $F/coresident.txt 2>&1
rm -rf $F/trace; cat $F/kernel_stats.txt | tail -8; cat $F/coresident.txt
python -c "
import json; d=json.load(open('$F/bench_traced.json')); r=d['roofline']; print('traced run: value', d['value'], 'launch_ms alone', r['launch_ms'], r['launch_ms_each'])"
