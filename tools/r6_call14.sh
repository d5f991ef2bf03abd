#!/bin/bash
# round 6, call: bench.py --gpus 2 under torch.distributed.run with both ranks on this box's ONE GPU, collectives on gloo -- a functional check of the refactored
# bench's N > 1 flow (shard, barrier, max over ranks, gather, rank-0 line), not a measurement
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2 --no-per-config --no-fresh-inputs --no-fast-mode --cpu-budget-s 0 > gpurun_out/r6_bench_2ranks.json 2> gpurun_out/r6_bench_2ranks.err; echo "rc $?"; tail -3 gpurun_out/r6_bench_2ranks.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6_bench_2ranks.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling')})
c = d['config']; print({k: c[k] for k in ('sequences_per_gpu_per_step', 'steps_per_engine_pass', 'sequences_per_engine_pass', 'engine_passes_in_timed_region', 'ggs_iterations_per_sequence_run', 'outputs_finite', 'headline_slots_equal_alone', 'parallelism')})
print('roofline frac', d['roofline']['frac'], d['roofline']['in_pipe'], 'cpu', d['cpu_baseline']['sample'][:60], 'rank_emulation' in d, 'from_images' in d)
PY
