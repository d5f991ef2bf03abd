cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests -m gpu -q -x -k "ggs or GGS or guided or async or graph_replay or full_size or bench_default or sampson or threshold or free_running or long_sequence" 2>&1 | tail -3
timeout 300 python tools/ggs_prof_k1.py 64 0 2>&1 | grep -v "Warn\|amdgpu.ids\|return nn"
timeout 400 python bench.py --steps 16 --warmup 4 --no-per-config --no-fresh-inputs --cpu-budget-s 0 > /tmp/b.json 2> /tmp/b.err
python -c "
import json; d=json.load(open('/tmp/b.json')); r=d['roofline']; rd=d['roofline_denoiser']; print('bench: value', round(d['value'],1), 'ggs ms', round(r['launch_ms'],2), 'co-res', round(r['co_resident']['wall_ms'],2), 'den us', round(rd['step_us'],1))" || tail -5 /tmp/b.err
