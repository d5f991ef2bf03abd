mkdir -p gpurun_out/r4k
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4k/pytest_gpu.txt
tail -8 gpurun_out/r4k/pytest_gpu.txt
for rep in 0 1; do
PD_ENGINE_LIB=$PWD/gpurun_ab/libpd_engine_r3.so timeout 600 python bench.py --steps 20 --warmup 5 --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe > gpurun_out/r4k/bench_r3_$rep.json 2> gpurun_out/r4k/bench_r3_$rep.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 --no-stream-probe > gpurun_out/r4k/bench_new_$rep.json 2> gpurun_out/r4k/bench_new_$rep.err
done
python - <<'PY'
import json
for n in ("r3_0","new_0","r3_1","new_1"):
    try:
        d=json.load(open(f"gpurun_out/r4k/bench_{n}.json"))
        print(n, "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "ggs launch ms", round(d["roofline"]["launch_ms"],2), "den step us", round(d["roofline_denoiser"]["step_us"],1), "all ctx", d["roofline_denoiser"]["all_contexts_step_us"], "cold", d.get("cold_single_batch",{}).get("latency_ms"), "iters", d["config"]["ggs_iterations_per_sequence_run"])
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r4k/bench_{n}.err").read()[-1500:])
PY
