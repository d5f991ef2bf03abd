#!/bin/bash
# Round 5: the lane kernel with LDS-resident steps beside a shallower ring, in the full pipe (bench.py, 256 x 3 in flight), libraries alternating on one box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=gpurun_out/c7; rm -rf $O; mkdir -p $O
Q="--no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe"
for rep in 0 1; do
 for tag in ring6 r4l2 r3l3; do
  lib=$R/gpurun_ab/libpd_$tag.so; [ $tag = ring6 ] && lib=$R/posediffusion_amd/lib/libpd_engine.so
  PD_ENGINE_LIB=$lib timeout 300 python bench.py $Q 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); r=d['roofline']; e=d['roofline_denoiser']
print('$tag', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'ggs ms alone', round(r['launch_ms'],2), 'three together', round(r['co_resident']['wall_ms'],1), 'den all ctx', round(e['all_contexts_step_us'],1), 'slots_equal', d['config'].get('headline_slots_equal_alone'))" | tee -a $O/bench_ab.txt
 done
done
