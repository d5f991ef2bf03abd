cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -s 2>&1 | grep -E "^E|free-running|configs\[4\]|passed|failed|^FAILED|^tests/.*[0-9]+: in" | cut -c1-300 | head -60 | tee $O/pytest.txt
tools/valu_probe 2>&1 | tee $O/valu_probe.txt
timeout 300 python tools/ggs_prof_k1.py 2>&1 | grep -v "Warn\|amdgpu.ids\|return nn" | tee $O/ggs_prof_k1.txt
timeout 400 python bench.py --steps 16 --warmup 4 --no-per-config --cpu-budget-s 0 > $O/bench_64x4.json 2> $O/bench_64x4.err
python -c "
import json; d=json.load(open('$O/bench_64x4.json')); r=d['roofline']; print('bench 64x4: value', round(d['value'],1), 'ggs launch ms', round(r['launch_ms'],2), 'co-res', round(r['co_resident']['wall_ms'],2), 'den us', round(d['roofline_denoiser']['step_us'],1), 'fresh', d.get('fresh_inputs',{}).get('value'), 'same', d.get('fresh_inputs',{}).get('first_pass_bitwise_equals_resident_pass'))" || tail -5 $O/bench_64x4.err
