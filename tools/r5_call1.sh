#!/bin/bash
# Round 5, first GPU call: the whole GPU suite (new: headline-launch lane tests, fused attention, _first hoist), same-box A / B of the denoiser
# against round 4's library (gpurun_ab/libpd_r4.so = sha 185960b2...), clean per-kernel statistics of a large-batch denoiser step for both,
# the headline for both, and the rank-shape sweep round 4 left unrun.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c1; rm -rf $O; mkdir -p $O
NEW=$R/posediffusion_amd/lib/libpd_engine.so; OLD=$R/gpurun_ab/libpd_r4.so
timeout 1200 python -m pytest tests -m gpu -q -rfE --tb=short -s 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -120 > $O/pytest.txt; tail -5 $O/pytest.txt
timeout 300 python tools/den_large_ab.py $OLD $NEW > $O/den_large_ab.txt 2>&1; cat $O/den_large_ab.txt
timeout 200 python tools/den_ab.py $OLD $NEW > $O/den_small_ab.txt 2>&1; cat $O/den_small_ab.txt
cd /tmp && export TMPDIR=/tmp
for tag in r4 r5; do
  lib=$NEW; [ $tag = r4 ] && lib=$OLD
  PD_ENGINE_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/tr_$tag -o den -- python $R/tools/den_large.py 256 > $R/$O/den_large_$tag.txt 2>&1
  python $R/tools/rocpd_stats.py $R/$O/tr_$tag/den_results.db 24 > $R/$O/denoiser_step_kernel_stats_$tag.txt 2>&1; rm -rf $R/$O/tr_$tag
done
cd $R
Q="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe"
for tag in r4 r5; do
  lib=$NEW; [ $tag = r4 ] && lib=$OLD
  PD_ENGINE_LIB=$lib timeout 300 python bench.py $Q 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); r=d['roofline']; e=d['roofline_denoiser']
print('$tag', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms', round(r['launch_ms'],2), 'den us', round(e['step_us'],1), 'all ctx', e['all_contexts_step_us'], 'slots_equal', d['config'].get('headline_slots_equal_alone'))" | tee -a $O/bench_ab.txt
done
bash tools/rank_shape_sweep.sh > $O/rank_shape_sweep.txt 2>&1; cat $O/rank_shape_sweep.txt
