#!/usr/bin/env python
"""bench.py -- sequences/sec of the PoseDiffusion sampling hot path on MI355X (BASELINE.json metric).

A "step" = one full pass of the hot path over one batch of synthetic sequences:
  100 DDPM steps (transformer denoiser + posterior update) and, on the last `cond_start_step`=10
  steps, Geometry-Guided Sampling (5 optimisations = 700 clipped-momentum-SGD iterations per guided
  step, 7000 per sequence) over M = 57 000 pairwise matches per 20-frame sequence.
Workload (config.workload): BASELINE.json configs[3], "batch of 64 independent 20-frame sequences, GGS on"
  (224^2, 190 pairs x 300 matches each), the whole batch on ONE GPU ("also run 64 on 1/2/4 GPUs"); weak
  scaling: every GPU runs its own 64-sequence batches (total = 64 x n_gpus per step).  Four batches are in flight per
  GPU (posediffusion_amd/pipeline.py), i.e. 256 sequences, one GGS workgroup = one CU per sequence; throughput
  against the number in flight is in DESIGN.md section 5 (`--seqs-per-gpu 8`: 272 sequences/s at 73 ms latency).
  Inputs are resident in HBM before the timed region.

Launch: `python bench.py --gpus N --steps K --warmup W`; for N > 1 under torch.distributed.run
(one rank per GPU, RCCL): ranks shard the sequences, no collective on the data path, one final
all_gather of the [B_local,20,9] poses inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      -- dominant kernel (pd_ggs_kernel): algorithmic FLOPs / hipEvent-timed launch
  cpu_baseline  -- the oracle (torch-CPU restatement of the reference path, kind "port") timed on the
                   host cores of this box on a bounded sample, extrapolated to sequences/s
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from posediffusion_amd import shard, synth  # noqa: E402

N_FRAMES = 20
IMG = 224
PER_PAIR = 300
SEQS_PER_GPU = 64                    # BASELINE configs[3]: one batch of 64 independent sequences
COND_START = 10                      # cfgs/default.yaml:8
FLOP_PER_MATCH_ITER = 100.0          # 36 fwd + 64 bwd
DENOISER_PARAMS = 17_298_697         # fp32 -> 69.19 MB read per denoiser step
DENOISER_MFLOP_PER_TOKEN = 34.73     # SURVEY.md section 8(d), N = 20
PD_STREAM_MIN_ROWS = 1024            # csrc/pd_gemm_stream.h
FP32_PEAK_TFLOPS = 157.3             # MI355X fp32 vector ALU = dense fp32 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
MATCH_BYTES = 16                     # kp1, kp2: 2 x float2 per match (pair indices are per work item)
L2_TOTAL_BYTES = 8 * 4 * 2 ** 20     # 8 XCDs x 4 MiB


# per-XCD persistent denoiser: workgroups per XCD when [alone on the chip, several batches in flight]; 0 = per-launch kernels
DEFAULT_DEN_WGS = {False: 0, True: 0}


def build_inputs(eng, diff, B, dev, seed0):
    """Synthetic inputs for B local sequences (global indices seed0 .. seed0+B-1), all resident on the
    device: z, reference-order noise, and matches that are epipolar-consistent with the engine's own
    unguided model mean at the first guided step (so every guided step runs its full 700 iterations,
    as it does with a trained checkpoint and real SuperGlue matches)."""
    from posediffusion_amd.host import draw_noise
    T = diff.num_timesteps
    z = torch.cat([synth.make_z(1, N_FRAMES, seed=1000 + seed0 + b) for b in range(B)]).to(dev)
    noise = torch.empty(T + 1, B, N_FRAMES, 9, device=dev)
    for b in range(B):
        g = torch.Generator(device=dev).manual_seed(seed0 + b)            # cfg.seed (+ sequence index)
        noise[:, b] = draw_noise((N_FRAMES, 9), T, dev, COND_START, True, generator=g)
    # unguided run -> model mean at t = COND_START-1 is what GGS first sees
    _, process, _ = eng.sample(z, noise, 0, None, use_graph=False)
    x_at = process[T - COND_START]                                         # x_t for t = COND_START-1
    mean, _ = eng.p_mean(x_at, z, COND_START - 1)
    mean = mean.cpu().numpy().astype(np.float64)
    for b in range(B):
        md = synth.make_epipolar_matches(mean[b], IMG, IMG, PER_PAIR, seed=2000 + seed0 + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    return z, noise


def _vit_flops(n, size, sf, D=384, L=12, FF=1536, P=16):
    hs = size if sf == 1 else int(size * sf)
    p = (hs // P) ** 2
    t = p + 1
    return n * (2 * p * 3 * P * P * D + L * (2 * t * (D * 3 * D + D * D + 2 * D * FF) + 4 * t * t * D))


def pmc_traffic():
    """HBM-side traffic per launch from the committed rocprofv3 PMC summary (bench.py cannot collect counters
    itself: they need their own `rocprofv3 --pmc` passes).  -> (ggs_bytes, denoiser_step_bytes, provenance) or Nones."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return (d["ggs_launch"]["traffic_bytes_corrected"], d["denoiser_step"]["traffic_bytes_corrected"],
                "profiles/round1_pmc_summary.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, separate --pmc passes, "
                "gfx950 FETCH_SIZE x2 correction; fabric-side counters (Infinity-Cache hits included); measured on one "
                f"{SEQS_PER_GPU}-sequence batch, one GGS workgroup per sequence")
    except Exception:
        return None, None, None


def cpu_baseline(budget_s: float):
    """Oracle (torch-CPU port of the reference path) on a bounded sample -> sequences/s."""
    from oracle import pd_oracle as O
    # tiny GEMMs oversubscribe badly on a many-core host (128 threads: 80 ms/step vs 24 ms on 8): probe a few
    # thread counts on the denoiser and keep the fastest for the whole sample
    diff = synth.make_diffuser(seed=0)
    sd = O.cast_state_dict(diff.model.state_dict(), torch.float32)
    z = synth.make_z(1, N_FRAMES)
    x = torch.randn(1, N_FRAMES, 9, generator=torch.Generator().manual_seed(0))
    tt = torch.full((1,), 50, dtype=torch.long)
    max_threads = torch.get_num_threads()
    best = (float("inf"), max_threads)
    with torch.no_grad():
        for n in sorted({min(max_threads, c) for c in (4, 8, 16, 32, max_threads)}):
            torch.set_num_threads(n)
            O.denoiser_forward(sd, x, tt, z)
            t0 = time.time()
            for _ in range(3):
                O.denoiser_forward(sd, x, tt, z)
            best = min(best, ((time.time() - t0) / 3, n))
    threads = best[1]
    torch.set_num_threads(threads)
    with torch.no_grad():
        O.denoiser_forward(sd, x, tt, z)                                   # warm-up
        n_den, t0 = 0, time.time()
        while n_den < 3 or (time.time() - t0 < 0.25 * budget_s and n_den < 50):
            O.denoiser_forward(sd, x, tt, z)
            n_den += 1
        t_den = (time.time() - t0) / n_den
    enc = synth.make_cameras(N_FRAMES, seed=2000)
    md = synth.make_matches(enc, IMG, IMG, per_pair=PER_PAIR, seed=2000)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = synth.perturb_pose(enc, seed=7)
    O.ggs_optimize(x0.clone(), pm, iter_num=1)                             # warm-up (2 iterations)
    n_it = max(4, min(200, int(0.75 * budget_s / 0.04)))
    t0 = time.time()
    _, _, steps = O.ggs_optimize(x0.clone(), pm, update_R=True, update_T=False, update_FL=False, iter_num=n_it)
    t_it = (time.time() - t0) / max(steps, 1)
    t_seq = 100 * t_den + 7000 * t_it
    torch.set_num_threads(max_threads)
    return {"value": 1.0 / t_seq, "unit": "sequences/s", "cores": threads, "kind": "port",
            "sample": f"{n_den} denoiser steps (B=1,N=20) + {steps} GGS iterations (M=57000) of oracle/pd_oracle.py "
                      f"(torch {torch.__version__} CPU, {threads} threads): {t_den * 1e3:.1f} ms/step, {t_it * 1e3:.1f} ms/iter; "
                      f"extrapolated to 100 steps + 7000 iterations per sequence"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--seqs-per-gpu", type=int, default=SEQS_PER_GPU)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ggs-slots", type=int, default=4,
                    help="how many of the batches in flight may be in their guided (GGS) half at once")
    ap.add_argument("--unguided-streams", type=int, default=0,
                    help="0: every stream runs whole passes; u > 0: two-stage pipeline with u streams for the unguided halves")
    ap.add_argument("--ggs-wgs", type=int, default=0, help="override GGS workgroups per sequence (0 = from the slot count)")
    ap.add_argument("--pipeline-depth", type=int, default=4,
                    help="engine contexts / HIP streams per GPU; consecutive passes (different batches) overlap: the next "
                         "batch's unguided denoiser steps run on the CUs the persistent GGS kernel leaves free. 1 = serial")
    ap.add_argument("--denoiser-wgs-per-xcd", type=int, default=-1,
                    help="per-XCD persistent denoiser kernel: workgroups per XCD (0 = per-launch kernels, -1 = default)")
    ap.add_argument("--trace", action="store_true", help="print the pipeline timeline (per-batch phase times) to stderr")
    ap.add_argument("--no-image-features", action="store_true",
                    help="skip the extra (untimed for `value`) images -> features -> poses measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0, help="CPU seconds for the cpu_baseline sample (0 = skip)")
    args = ap.parse_args()

    rank, world, local = shard.init_distributed()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the sampling path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from posediffusion_amd.engine import make_ggs_cfg
    from posediffusion_amd.host import get_engine

    B = args.seqs_per_gpu
    total = B * world
    g0, g1 = shard.partition(total, world, rank)
    assert g1 - g0 == B
    diff = synth.make_diffuser(seed=0).to(dev)
    depth = max(1, args.pipeline_depth)
    slots = min(depth, max(1, args.ggs_slots))
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    from posediffusion_amd.pipeline import SamplingPipeline
    eng = get_engine(diff.model, diff, B, N_FRAMES)
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    engines = [eng] + [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N_FRAMES) for _ in range(depth - 1)]
    den_wgs = args.denoiser_wgs_per_xcd if args.denoiser_wgs_per_xcd >= 0 else DEFAULT_DEN_WGS[depth > 1]
    for e in engines:
        e.set_denoiser_wgs_per_xcd(den_wgs)
    pipe = SamplingPipeline(engines, slots, dev, unguided_streams=args.unguided_streams, trace=args.trace)
    # one resident batch per context (different sequences: seeds offset by the global batch size)
    inputs = [build_inputs(engines[j], diff, B, dev, seed0=g0 + j * total) for j in range(depth)]
    z, noise = inputs[0]
    # GGS workgroups per sequence: alone on the chip -> one work item per wave (24 WGs/sequence, lowest latency);
    # pipelined -> sized so that the persistent kernels of ALL batches that may be in their guided half are
    # co-resident (4 batches x 8 sequences x 8 workgroups = 256 CUs; see posediffusion_amd/pipeline.py)
    wgs = args.ggs_wgs if args.ggs_wgs > 0 else pipe.wgs_per_seq(B)
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs)
    use_graph = not args.no_graph
    torch.cuda.synchronize()

    last_pose = {}

    def one_step(i):
        j = pipe.next_context()
        pend = pipe.submit(inputs[j][0], inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False)
        last_pose[j] = pend.pose
        return pend.pose, pend.stats

    for j in range(depth):          # setup: every context captures its hipGraphs before anything is timed
        with torch.cuda.stream(pipe.u_stream):
            out = engines[j].sample(inputs[j][0], inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False, phase=1)
            engines[j].sample(inputs[j][0], inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False, phase=2, out=out)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = [one_step(i) for i in range(args.steps)]
    torch.cuda.synchronize()
    # the one data-path collective: every rank's poses of all K passes in ONE all_gather over xGMI (still timed)
    gathered = shard.gather_poses(torch.stack([r[0] for r in results], dim=1).contiguous(), total)   # [total, K, N, 9]
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dev)
    for e in engines:
        e.check_async()
    if args.trace and rank == 0:
        for i, (a, b2, c, d) in enumerate(pipe.timeline()):
            print(f"  sub {i:2d}: U {a:7.1f} -> {b2:7.1f} ({b2 - a:5.1f})   G {c:7.1f} -> {d:7.1f} ({d - c:5.1f})", file=sys.stderr)
    assert gathered.shape[0] == total and gathered.shape[1] == args.steps
    # un-overlapped latency of one pass (outside the timed region, reported next to the throughput)
    for rep in range(2):        # the first call captures the whole-loop graph
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(pipe.u_stream):
            engines[0].sample(z, noise, COND_START, cfg, use_graph=use_graph, want_process=False)
        torch.cuda.synchronize()
        pass_latency_ms = (time.perf_counter() - t1) * 1e3
    ms_per_step = dt / args.steps * 1e3
    value = total * args.steps / dt

    # ---- extra, outside the timed region of `value`: the same pipe fed from images (DINO ViT-S/16 at three
    # scales on every frame, csrc/pd_vit.hip) -- each batch's features are computed on its context's stream right before its pass
    feat = None
    if not args.no_image_features and pipe.whole_pass_streams:
        from posediffusion_amd.vit import VitEngine, vit_state
        torch.manual_seed(1)
        ext = synth._dropin().MultiScaleImageFeatureExtractor().to(dev)              # random-init DINO-shaped parameters
        vits = [VitEngine(vit_state(ext._net), dev) for _ in range(depth)]
        images = [torch.rand(B * N_FRAMES, 3, IMG, IMG, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + j))
                  for j in range(depth)]
        scales = (1, 1 / 2, 1 / 3)

        def one_step_images():
            j = pipe.next_context()
            with torch.cuda.stream(pipe.next_stream()):
                zj = vits[j].multiscale(images[j], scales).reshape(B, N_FRAMES, -1)
            return pipe.submit(zj, inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False)

        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(pipe.u_stream):
            for rep in range(3):
                if rep == 1:
                    e0.record()
                vits[0].multiscale(images[0], scales)
            e1.record()
        torch.cuda.synchronize()
        feat_ms = e0.elapsed_time(e1) / 2
        for _ in range(depth):
            one_step_images()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pend = [one_step_images() for _ in range(args.steps)]
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        for e in engines:
            e.check_async()
        feat_flops = sum(_vit_flops(B * N_FRAMES, IMG, sf) for sf in scales)
        feat = {"value": B * args.steps / dt2, "unit": "sequences/s on this GPU, images resident in HBM", "steps": args.steps,
                "features_ms_per_batch_alone": feat_ms, "features_tflops_alone": feat_flops / (feat_ms * 1e-3) / 1e12,
                "features_gflop_per_batch": feat_flops / 1e9, "frames_per_batch": B * N_FRAMES, "scales": "1, 1/2, 1/3",
                "outputs_finite": bool(all(torch.isfinite(p.pose).all().item() for p in pend[-depth:])),
                "precision": "split (bf16 hi + lo, three bf16 MFMA products, fp32 accumulate; z within 1e-5 of the fp32 network)",
                "note": "ViT-S/16 with random-init weights (no checkpoint offline); TFLOP/s are fp32-equivalent; not part of `value`"}
        for v in vits:
            v.close()

    # every guided step must have run its full 700 iterations (no data-dependent early exit skipped work)
    iters = torch.stack([r[1][:, :, :, 1].sum(dim=(0, 2)).cpu() for r in results])    # [pass, local sequence]
    finite = bool(torch.isfinite(gathered).all().item())

    # ---- roofline of the dominant kernel + the denoiser step, timed with hipEvents on the launch stream
    ggs_ms = eng.time_kernel(1, B, N_FRAMES, cfg, reps=3)
    den_ms = eng.time_kernel(2 if den_wgs > 0 else 0, B, N_FRAMES, cfg, reps=20)
    M = N_FRAMES * (N_FRAMES - 1) // 2 * PER_PAIR
    ggs_flops = B * M * FLOP_PER_MATCH_ITER * 7 * cfg.iter_num             # one pd_ggs_guide launch = 700 iterations
    ggs_tflops = ggs_flops / (ggs_ms * 1e-3) / 1e12
    # all contexts' GGS kernels together, as they run in the pipe: `depth` co-resident launches, wall time of the set
    evs = []
    torch.cuda.synchronize()
    for rep in range(2):
        evs = []
        for j in range(depth):
            st = pipe.g_streams[j % len(pipe.g_streams)]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st):
                e0.record(st)
                engines[j].ggs_guide(last_pose.get(j, results[-1][0]), 0, cfg)
                e1.record(st)
            evs.append((e0, e1))
        torch.cuda.synchronize()
    ggs_set_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs)
    ggs_set_tflops = depth * ggs_flops / (ggs_set_ms * 1e-3) / 1e12
    den_gbs = DENOISER_PARAMS * 4 / (den_ms * 1e-3) / 1e9
    # the weight stream (69 MB per step, HBM) bounds the denoiser up to ~50 tokens, the exact-fp32 matrix pipe above
    tokens = B * N_FRAMES
    den_flops = tokens * DENOISER_MFLOP_PER_TOKEN * 1e6
    den_tflops = den_flops / (den_ms * 1e-3) / 1e12
    if tokens >= PD_STREAM_MIN_ROWS:
        den_kernels = "one denoiser step = 59 launches (pd_gemm_stream_kernel x32, pd_ln_rows_kernel x16, pd_gemm_kernel x2, pd_attn_kernel x8, pd_tail_kernel)"
    else:
        den_kernels = "one denoiser step = 43 launches (pd_gemm_kernel x34, pd_attn_kernel x8, pd_tail_kernel)"

    ggs_traffic, den_traffic, traffic_src = pmc_traffic() if B == SEQS_PER_GPU else (None, None, None)
    # Which roof: with few sequences in flight the matches of a sequence stay in registers (24 workgroups per sequence) or in
    # the L2s and the kernel is priced against the fp32 vector ALU.  With the default 256 in flight (one workgroup
    # = one CU per sequence) their 16 B per match (kp1, kp2 as 2 x float2; the pair index is per work item) total
    # 233 MB and are streamed from the fabric EVERY iteration -- the PMC traffic equals these algorithmic bytes -- so the
    # kernel is priced against HBM.
    match_bytes = float(B) * M * MATCH_BYTES * 7 * cfg.iter_num                 # one launch = 700 iterations
    streams_matches = B * depth * M * MATCH_BYTES > L2_TOTAL_BYTES and (wgs or 24) < 24
    alu = {"achieved": ggs_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ggs_tflops / FP32_PEAK_TFLOPS,
           "achieved_all_launches": ggs_set_tflops, "frac_all_launches": ggs_set_tflops / FP32_PEAK_TFLOPS,
           "algorithmic_flops_per_launch": ggs_flops}
    roofline = {
        "kernel": "pd_ggs_kernel (one launch = one guided step = 700 iterations x %d sequences)" % B,
        "traffic": ggs_traffic, "traffic_source": traffic_src, "launch_ms": ggs_ms,
        "co_resident_launches": depth, "all_launches_ms": ggs_set_ms,
    }
    if streams_matches:
        gbs, set_gbs = match_bytes / (ggs_ms * 1e-3) / 1e9, depth * match_bytes / (ggs_set_ms * 1e-3) / 1e9
        roofline.update({
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "achieved_all_launches": set_gbs, "frac_all_launches": set_gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": match_bytes, "fp32_alu": alu,
            "note": f"{B * depth} sequences in flight: their matches ({B * depth * M * MATCH_BYTES / 1e6:.0f} MB) exceed the L2s "
                    "and stream from Infinity Cache / HBM every iteration, 16 B per match. `achieved` is ONE launch alone on the "
                    "chip (its workgroups cover a quarter of the CUs); `achieved_all_launches` is the set of co-resident "
                    "launches of all contexts, as they run in the pipe. `fp32_alu` prices the same launches against the "
                    "vector ALU (100 FLOP per match and iteration, SURVEY 8d)"})
    else:
        roofline.update(dict(alu, bound="mfma", bound_detail="fp32 vector ALU; its peak equals the dense fp32 MFMA peak (157.3 TFLOP/s)",
                             note=("matches stay in registers for the whole launch" if (wgs or 24) >= 24 else
                                   "matches are re-read from L2 every iteration (more than one work item per wave)") +
                                  "; the fabric traffic is the per-iteration cross-workgroup exchange, not match streaming. "
                                  "`achieved` is ONE launch; `achieved_all_launches` is the set of co-resident launches of all "
                                  "contexts, as in the pipe"))
    # the same step on all contexts at once, as in the unguided halves of the pipe: `reps` steps per context, each on its stream
    den_set_ms = None
    if depth > 1:
        reps = 10
        xs = [torch.randn(B, N_FRAMES, 9, device=dev) for _ in range(depth)]
        for rep in range(2):
            evs = []
            for j in range(depth):
                st = pipe.g_streams[j % len(pipe.g_streams)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    e0.record(st)
                    for _ in range(reps):
                        engines[j].denoise(xs[j], inputs[j][0], 50)
                    e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
        den_set_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs) / reps
    if tokens > 50:
        roofline_den = {"kernel": den_kernels, "bound": "mfma", "bound_detail": "exact-fp32 matrix instruction (157.3 TFLOP/s)",
                        "achieved": den_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": den_tflops / FP32_PEAK_TFLOPS,
                        "traffic": den_traffic, "step_us": den_ms * 1e3, "algorithmic_flops_per_step": den_flops,
                        "weights_GBps": den_gbs, "all_contexts_step_us": None if den_set_ms is None else den_set_ms * 1e3,
                        "achieved_all_contexts": None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12,
                        "frac_all_contexts": None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "note": f"{tokens} token rows per step; `achieved` is one context alone, `achieved_all_contexts` the "
                                f"{depth} contexts' steps running together as in the unguided halves of the pipe"}
    else:
        roofline_den = {"kernel": den_kernels, "bound": "hbm", "achieved": den_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": den_gbs / HBM_PEAK_GBS, "traffic": den_traffic, "step_us": den_ms * 1e3,
                        "algorithmic_bytes_per_step": DENOISER_PARAMS * 4}
    out = {
        "metric": "sequences/sec (20-frame, GGS on)", "value": value, "unit": "sequences/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: batches of {B} independent 20-frame sequences, one batch per step and GPU "
                        f"({total} sequences per step in total), {depth} batches in flight per GPU; 100 DDPM steps, GGS on for the last {COND_START} steps "
                        f"(7000 iterations/sequence), M={M} matches/sequence (190 pairs x {PER_PAIR}), {IMG}x{IMG}; "
                        "random-init reference-rule weights, matches epipolar-consistent with the engine's own "
                        "unguided model mean at t=9",
            "sequences_per_gpu": B, "sequences_in_flight_per_gpu": B * depth, "frames": N_FRAMES, "matches_per_sequence": M, "diffusion_steps": 100,
            "ggs_iterations_per_sequence_run": float(iters.min().item()), "hip_graph": use_graph,
            "denoiser_wgs_per_xcd": den_wgs, "pipeline_depth": depth, "ggs_slots": slots, "unguided_streams": 0 if pipe.whole_pass_streams else len(pipe.u_streams), "pass_latency_ms_unpipelined": pass_latency_ms, "ggs_workgroups_per_sequence": wgs or 24,
            "parallelism": f"dp{world} (independent sequences, one final all_gather)", "outputs_finite": finite,
        },
        "roofline": roofline,
        "roofline_denoiser": roofline_den,
        "per_step_ms": {"denoiser_step": den_ms, "ggs_guided_step": ggs_ms, "ggs_iteration_us": ggs_ms * 1e3 / (7 * cfg.iter_num)},
    }
    if feat is not None:
        out["from_images"] = feat
    if rank == 0:
        if args.cpu_budget_s > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_budget_s)
            except Exception as e:  # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "sequences/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        else:
            out["cpu_baseline"] = {"value": None, "unit": "sequences/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "skipped (measured on rank 0 at N=1 only)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
