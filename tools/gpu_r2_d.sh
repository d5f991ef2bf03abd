cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r2d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -E "^E|passed|failed|^FAILED|^tests/.*[0-9]+: in" | cut -c1-300 | head -30 | tee $O/pytest.txt
for B in 8 64 256; do echo "== B=$B"; timeout 300 python tools/ggs_prof_k1.py $B 0,4,2 2>&1 | grep -v "Warn\|amdgpu.ids\|return nn"; done | tee $O/ggs_prof.txt
for r in 0 4; do
PD_GGS_RESERVED=$r timeout 400 python bench.py --steps 16 --warmup 4 --no-per-config --no-fresh-inputs --cpu-budget-s 0 > $O/bench_r$r.json 2> $O/bench_r$r.err
python -c "
import json; d=json.load(open('$O/bench_r$r.json')); r=d['roofline']; print('bench reserved=$r: value', round(d['value'],1), 'ggs launch ms', round(r['launch_ms'],2), 'co-res', round(r['co_resident']['wall_ms'],2), 'den us', round(d['roofline_denoiser']['step_us'],1))" || tail -5 $O/bench_r$r.err
done
