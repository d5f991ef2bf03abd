#!/bin/bash
# round 5, call 12: the N > 1 flow of bench.py as the driver launches it (torch.distributed.run, 2 ranks), functional check on a 1-GPU box: both ranks
# share GPU 0 and the collective runs on gloo (PD_DIST_BACKEND=gloo; RCCL needs a device per rank) -- NOT a performance figure
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
PD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2 > gpurun_out/bench_2ranks_one_gpu.json 2> gpurun_out/bench_2ranks_one_gpu.err; echo rc=$?
tail -3 gpurun_out/bench_2ranks_one_gpu.err; python -c "
import json; d=json.load(open('gpurun_out/bench_2ranks_one_gpu.json')); print({k: d[k] for k in ('metric','value','n_gpus','steps','warmup','ms_per_step','scaling')}); print(d['config'])"
