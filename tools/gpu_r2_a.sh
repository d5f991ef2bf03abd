#!/bin/bash
# round 2, GPU call A: new parity tests first, then the whole GPU suite, then three bench shapes (same binary).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_demo_dropin.py -m gpu -x -q -s 2>&1 | tail -40 > $O/pytest_r2.txt; tail -15 $O/pytest_r2.txt
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity_r2.py --deselect tests/test_demo_dropin.py 2>&1 | tail -15 > $O/pytest_all.txt; tail -5 $O/pytest_all.txt
for cfg in "64 4" "256 1" "128 2"; do set -- $cfg
  timeout 400 python bench.py --seqs-per-gpu $1 --pipeline-depth $2 --steps $((8*$2)) --warmup $2 --no-image-features --cpu-budget-s 0 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1x$2.json")); r=d["roofline"]
    print("bench $1x$2: value", round(d["value"],1), "ms/step", round(d["ms_per_step"],1), "ggs launch ms", round(r["launch_ms"],2), "set", round(r["all_launches_ms"],2), "den step us", d["roofline_denoiser"]["step_us"], "lat", d["config"]["pass_latency_ms_unpipelined"])
except Exception as e: print("bench $1x$2 failed", e); print(open("$O/bench_$1x$2.err").read()[-1500:])
PY
done
# the same default shape without LDS staging (reserved flag 2) for the A/B
PD_GGS_RESERVED=2 timeout 400 python bench.py --steps 16 --warmup 4 --no-image-features --cpu-budget-s 0 > $O/bench_64x4_nostage.json 2> $O/bench_64x4_nostage.err
python -c "
import json; d=json.load(open('$O/bench_64x4_nostage.json')); print('no staging: value', round(d['value'],1), 'ggs launch ms', round(d['roofline']['launch_ms'],2))" || tail -5 $O/bench_64x4_nostage.err
