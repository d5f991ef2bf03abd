#!/bin/bash
# End-of-round evidence with the final binary (on the GPU box): full GPU suite, smoke, counters, the default bench line, a kernel
# trace of the same command (+ the co-resident analysis), a trace of ONE 256-sequence GGS launch that fills the chip, and the
# reference's demo.py running unchanged on the drop-in when a staged copy of the reference tree travelled along (_ref_stage/).
# Results under gpurun_out/final/ -> copy into profiles/round6_*.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; F=gpurun_out/final; rm -rf $F; mkdir -p $F
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $F/pytest_gpu.txt; cat $F/pytest_gpu.txt; cp gpurun_out/pose_group_errors.json $F/ 2>/dev/null
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "vit_large_batch" 2>&1 | grep "ViT mode" > $F/vit_modes.txt; cat $F/vit_modes.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $F/smoke.txt
bash tools/collect_pmc.sh > /dev/null 2>&1; cp gpurun_out/pmc_summary.json $F/pmc_summary.json
mkdir -p profiles; cp gpurun_out/pmc_summary.json profiles/round6_pmc_summary.json      # bench.py reads the traffic figures from here (hash-checked)
[ -d _ref_stage/pose_diffusion ] && export PD_REFERENCE_ROOT=$R/_ref_stage
T0=$SECONDS; timeout 1200 python bench.py --dry-dist > $F/bench_line.json 2> $F/bench.err; echo "default bench.py run: $((SECONDS - T0)) s wall" | tee $F/bench_wall.txt; tail -2 $F/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/$F/trace -o bench -- python $R/bench.py --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-from-images --no-rank-emulation --no-stream-probe > $R/$F/bench_traced.json 2>/dev/null
cd $R; python tools/rocpd_stats.py $F/trace/bench_results.db 16 > $F/kernel_stats.txt
python tools/coresident_from_trace.py $F/trace/bench_results.db $(python -c "import json; print(json.load(open('$F/bench_traced.json'))['roofline']['algorithmic_flops_per_launch'])") > $F/coresident.txt 2>&1; cat $F/coresident.txt; rm -rf $F/trace
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$F/trace256 -o t256 -- python $R/tools/pmc_target.py 256 1 > /dev/null 2>&1
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$F/trace_den -o den -- python $R/tools/den_large.py 256 > $R/$F/den_large.txt 2>&1
cd $R; python tools/rocpd_stats.py $F/trace_den/den_results.db 20 > $F/denoiser_step_kernel_stats.txt 2>&1; rm -rf $F/trace_den       # steps alone, no GGS in the process
python tools/rocpd_stats.py $F/trace256/t256_results.db 6 > $F/fullchip_launch_stats.txt
python tools/coresident_from_trace.py $F/trace256/t256_results.db $(python -c "print(256*57000*100.0*700)") >> $F/fullchip_launch_stats.txt 2>&1; rm -rf $F/trace256; cat $F/fullchip_launch_stats.txt | tail -4
if [ -d _ref_stage/pose_diffusion ]; then
  python tools/make_synthetic_ckpt.py /tmp/synth.pth --cfg _ref_stage/cfgs/default.yaml > /dev/null 2>&1
  (cd _ref_stage/pose_diffusion && PYTHONPATH=$R timeout 600 python -m posediffusion_amd.run_reference demo.py image_folder=samples/apple ckpt=/tmp/synth.pth GGS.enable=False 2>&1 | grep -v Warning | tail -8) > $F/demo_ggs_off.log; cat $F/demo_ggs_off.log | tail -5
fi
# cpu_baseline reproducibility: the same leg twice per kind -- the reference files in place (staged copy) and the oracle port -- on this box's host cores
for kind in reference port; do for rep in 0 1; do
  if [ $kind = reference ] && [ -d _ref_stage/pose_diffusion ]; then export PD_REFERENCE_ROOT=$R/_ref_stage; else export PD_REFERENCE_ROOT=/nonexistent; fi
  timeout 600 python -c "
import json, bench_legs as L
r = L.cpu_baseline(30.0); print(json.dumps({k: r[k] for k in ('kind', 'value', 'cores', 'host_cores', 'denoiser_ms_per_step', 'guided_step_s', 'guided_step_iterations_timed')}))" 2>/dev/null | tail -1
done; done > $F/cpu_baseline_repeat.txt; cat $F/cpu_baseline_repeat.txt
python -c "
import json; d=json.load(open('$F/bench_line.json')); r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'ggs launch ms (in pipe)', r['launch_ms'], 'frac', r['frac'], 'in_pipe', r['in_pipe'], 'alone', r['alone']['launch_ms'], r['alone']['frac'], 'traffic', r['traffic'])
print('denoiser', d['roofline_denoiser']['step_us'], d['roofline_denoiser']['frac'], d['roofline_denoiser'].get('all_contexts_step_us'))
print('rank_emulation', {k: (v.get('predicted_sequences_per_s') if isinstance(v, dict) else v) for k, v in (d.get('rank_emulation') or {}).items()})
print('from_images', d.get('from_images'))
print('fresh', d.get('fresh_inputs',{}).get('value'), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])
for k,v in (d.get('per_config') or {}).items(): print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items()})"
head -8 $F/kernel_stats.txt
