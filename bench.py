#!/usr/bin/env python
"""bench.py -- sequences/sec of the PoseDiffusion sampling hot path on MI355X (BASELINE.json metric).

A "step" = one full pass of the hot path over ONE batch of 64 independent synthetic sequences (BASELINE.json configs[3]):
  100 DDPM steps (transformer denoiser + posterior update) and, on the last `cond_start_step` = 10 steps,
  Geometry-Guided Sampling (5 optimisations = 700 clipped-momentum-SGD iterations per guided step, 7 000 per sequence)
  over M = 57 000 pairwise matches per 20-frame sequence (190 pairs x 300, 224 x 224).  Inputs (z, noise, matches) are
  resident in HBM before the timed region.

Scaling (`--scaling`, default strong = what configs[3] names: "batch of 64 ... sharded across 8 x MI355X"):
  strong  every step's 64 sequences are block-partitioned over the N ranks (64 / N each).  A rank runs the shards of
          several consecutive steps as ONE engine pass of up to `--engine-batch` = 256 sequences (a GGS launch of 256
          workgroups holds every CU once), `--pipeline-depth` passes in flight, the run's passes balanced
          (shard.steps_per_pass) -- so the per-GPU launch shape is the single-GPU one as long as a rank has that many
          sequences, and the pipeline fill / drain of a short run is what costs.  Total work is fixed: K x 64 sequences.
  weak    every rank runs its own 64-sequence batch per step (64 x N sequences per step).
There is no collective on the data path; ONE all_gather of the poses of all K steps ends the timed region.

Launch: `python bench.py --gpus N --steps K --warmup W`; for N > 1 under torch.distributed.run (one rank per GPU, RCCL).

Prints ONE JSON line on rank 0 (contract in the task statement) with, besides the required keys:
  roofline        dominant kernel (pd_ggs_lane_kernel<14>): `frac` = algorithmic fp32 FLOPs of ONE launch / the MEAN duration of the launches
                  the timed region itself ran (every one of them stamped by the kernel: pd_ggs_launch_stamps) / 157.3 TFLOP/s (SURVEY 8d
                  prices it against the fp32 vector ALU); sub-objects `in_pipe` (those launches), `alone` (a launch on an idle chip,
                  hipEvents: a side figure), `co_resident`, `fabric` (match bytes streamed per iteration -- Infinity-Cache / fabric
                  bandwidth, NOT HBM); `traffic` from the committed PMC summary, only when that summary was collected with THIS
                  libpd_engine.so (sha256 recorded there)
  roofline_denoiser
  per_config      BASELINE configs[1], [2], [3]-shard and [4], each alone on the chip (ms per pass, sequences/s)
  exact_mode      the same pipe with the denoiser's encoder GEMMs on the exact-fp32 matrix instruction (PD_OPT_DENOISER_SPLIT = 0)
                  instead of the default fp16 hi + lo operand pairs (three fp16 MFMA products, fp32 accumulation)
  fresh_inputs    the same pipe with every pass uploading NEW z / noise / matches inside the timed region
                  (pinned host -> device copies + asynchronous device-side match ingestion)
  cold_single_batch  ONE batch of 64 sequences alone on an idle chip (latency, sequences/s): what `value`'s steady-state rate is NOT
  dry_dist        (--dry-dist on a single-GPU box) the N > 1 collectives executed on RCCL with a world of one rank
  from_images     the drop-in PoseDiffusionModel.forward on 20 images (DINO features at three scales + sampler + decode), GGS off / on:
                  the one workload the reference publishes a wall time for (README.md:45)
  rank_emulation  one rank of the driver's 2 / 4 / 8-GPU strong-scaling run emulated on this GPU, with the predicted curve
  cpu_baseline    the reference files verbatim (kind "reference") when the reference tree is present, else the oracle
                  port (kind "port"), on the host cores of this box: fixed thread counts, 20 denoiser steps + ONE whole guided step

The legs live in bench_legs.py (one callable each); this file owns the command line, the timed region and the JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_legs as L  # noqa: E402
from bench_legs import COND_START, IMG, N_FRAMES, PD_STREAM_MIN_ROWS, PER_PAIR, STEP_SEQS  # noqa: E402
from bench_legs import cpu_baseline, lane_stream_fraction, make_batch_inputs, pmc_traffic, stream_ceiling  # noqa: E402,F401  (tests import them from here)
from posediffusion_amd import shard, synth  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--seqs-per-step", type=int, default=STEP_SEQS, help="sequences per step: in total (strong) / per GPU (weak)")
    ap.add_argument("--engine-batch", type=int, default=256,
                    help="most sequences per engine pass (a rank groups the shards of consecutive steps up to this; 256 = one GGS workgroup per CU)")
    ap.add_argument("--min-passes", type=int, default=2,
                    help="fewest engine passes a rank's run is cut into (shard.steps_per_pass; 2 = a second context overlaps the first; "
                         "1 lets a short run -- an 8-GPU rank's 160 sequences -- go as ONE pass)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ggs-wgs", type=int, default=0, help="override GGS workgroups per sequence (0 = from the pipeline shape)")
    ap.add_argument("--pipeline-depth", type=int, default=3, help="engine contexts / HIP streams per GPU (engine passes in flight); 1 = serial")
    ap.add_argument("--unguided-streams", type=int, default=0,
                    help="0 = every context's stream runs whole passes; u > 0 = the two-stage pipe of SamplingPipeline: u streams run the unguided "
                         "halves, --ggs-slots streams the guided halves (their GGS launches then never meet another slot's)")
    ap.add_argument("--ggs-slots", type=int, default=0, help="streams that run guided halves (0 = one per context)")
    ap.add_argument("--trace", action="store_true", help="print the pipeline timeline (per-pass phase times) to stderr")
    ap.add_argument("--no-per-config", action="store_true", help="skip the per-BASELINE-config measurements")
    ap.add_argument("--no-fast-mode", "--no-exact-mode", dest="no_fast_mode", action="store_true",
                    help="skip the comparison run with the encoder GEMMs on the exact-fp32 matrix instruction")
    ap.add_argument("--no-fresh-inputs", action="store_true", help="skip the fresh-inputs (upload inside the timed region) measurement")
    ap.add_argument("--no-from-images", action="store_true", help="skip the images -> cameras leg (DINO features + sampler + decode)")
    ap.add_argument("--no-rank-emulation", action="store_true", help="skip the emulation of one rank of the 2 / 4 / 8-GPU runs")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="CPU seconds the cpu_baseline's one guided step may take before it is cut short (0 = skip)")
    ap.add_argument("--dry-dist", action="store_true",
                    help="single-GPU box: initialise RCCL (backend nccl) with a world of ONE rank and run the barrier / all_gather / "
                         "max-over-ranks call sites of the N > 1 path anyway, so that they execute at least once on GPU hardware")
    ap.add_argument("--no-stream-probe", action="store_true", help="skip tools/stream_probe (this box's match-stream reference rates)")
    ap.add_argument("--no-launch-stamps", action="store_true",
                    help="do not read the GGS launch stamps of the timed region (roofline.frac then falls back to the launch alone; A/B of the readout's cost)")
    return ap.parse_args(argv)


def setup(args, rank, world, dev):
    """Engines, pipe, resident inputs and the run's schedule -> bench_legs.Bench."""
    from posediffusion_amd import _lib
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state, get_engine
    from posediffusion_amd.pipeline import SamplingPipeline
    K = args.steps
    strong = args.scaling == "strong"
    step_total = args.seqs_per_step * (1 if strong else world)                  # sequences of one step over all ranks
    if strong:
        g0, g1, group, _ = shard.strong_schedule(K, step_total, world, rank, args.engine_batch, args.min_passes)
    else:
        g0, g1 = rank * args.seqs_per_step, (rank + 1) * args.seqs_per_step
        group = shard.steps_per_pass(K, args.seqs_per_step, args.engine_batch, args.min_passes)
    B_step = g1 - g0                                                            # this rank's sequences of one step
    if B_step <= 0:
        raise SystemExit(f"rank {rank} has no sequences: {step_total} per step over {world} ranks")
    EB = B_step * group                                                         # sequences per engine pass
    depth = max(1, args.pipeline_depth)
    diff = synth.make_diffuser(seed=0).to(dev)
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    eng = get_engine(diff.model, diff, EB, N_FRAMES)
    engines = [eng] + [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=EB, max_N=N_FRAMES) for _ in range(depth - 1)]
    slots = depth if args.ggs_slots <= 0 else min(args.ggs_slots, depth)
    pipe = SamplingPipeline(engines, slots, dev, unguided_streams=max(0, args.unguided_streams), trace=args.trace)
    # one resident engine batch per context (different sequences: seeds offset per context and rank)
    want_fresh = not args.no_fresh_inputs
    check_slots = sorted({0, min(EB - 1, ((EB - 1) // 3) | 1), EB - 1})        # headline_slots_equal_alone: these slots of context 0 are re-run alone
    inputs = [make_batch_inputs(engines[j], diff, EB, dev, seed0=100_000 * rank + j * EB,
                                keep_host=True if want_fresh else (check_slots if j == 0 else False)) for j in range(depth)]
    wgs = args.ggs_wgs if args.ggs_wgs > 0 else pipe.wgs_per_seq(EB)
    flags = int(os.environ.get("PD_GGS_RESERVED", "0"))                         # A/B switch, pd_engine.h
    if wgs == 1 and not (flags & _lib.PD_GGS_CFG_NO_LANE_ITEMS):
        flags |= _lib.PD_GGS_CFG_LANE_ITEMS       # one workgroup per sequence: the lane-per-item kernel (SamplingPipeline.make_cfg does the same)
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs, reserved=flags)
    bc = L.Bench(args=args, rank=rank, world=world, dev=dev, diff=diff, tables=tables, eng=eng, engines=engines, pipe=pipe, inputs=inputs, cfg=cfg,
                 wgs=wgs, EB=EB, depth=depth, slots=slots, B_step=B_step, group=group, K=K, strong=strong, step_total=step_total,
                 use_graph=not args.no_graph, check_slots=check_slots, want_fresh=want_fresh, sliced={})

    def pass_inputs(j, b):                                                      # resident tensors per pass size: no copies in the timed region
        if b == EB:
            return inputs[j][0], inputs[j][1]
        if (j, b) not in bc.sliced:
            bc.sliced[(j, b)] = (inputs[j][0][:b].contiguous(), inputs[j][1][:, :b].contiguous())
        return bc.sliced[(j, b)]

    def submit(b_seqs):
        """One engine pass of b_seqs sequences on the next context; the launch stamps of its GGS launches are copied out behind it, on its
        stream (context j always runs on stream j here: the copy is ordered before the context's next pass)."""
        j = pipe.next_context()
        z, noise = pass_inputs(j, b_seqs)
        p = pipe.submit(z, noise, COND_START, cfg, use_graph=bc.use_graph, want_process=False)
        p.stamps = None
        if bc.stamp_rows is not None and bc.stamps_used < bc.stamp_rows.shape[0]:
            with torch.cuda.stream(p.stream):
                p.stamps = engines[p.context].ggs_launch_stamps(COND_START, out=bc.stamp_rows[bc.stamps_used])
            bc.stamps_used += 1
        return p

    bc.pass_inputs, bc.submit = pass_inputs, submit
    # one preallocated row of launch stamps per pass of the warm-up and the timed region (nothing is allocated between the passes of the pipe)
    bc.stamp_rows = torch.zeros(4 * (K + args.warmup) + 64, COND_START, 2, dtype=torch.int64, device=dev) if (hasattr(eng.lib, "pd_ggs_launch_stamps") and not args.no_launch_stamps) else None
    bc.stamps_used = 0
    return bc


def passes_for(bc, k_steps):
    """The passes of a k-step run: pass p covers steps [p * group, min(k, (p + 1) * group)) -> that many shards of B_step sequences."""
    return [min(bc.group, k_steps - s0) * bc.B_step for s0 in range(0, k_steps, bc.group)]


def timed_region(bc, dry):
    """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, max over ranks; one all_gather of the poses inside."""
    args, pipe, K = bc.args, bc.pipe, bc.K
    stagger_ms = float(os.environ.get("PD_BENCH_STAGGER_MS", "0"))      # experiment knob, default off (measured: profiles/round2_overlap_probe.txt)
    sleep_cycles_per_ms = 0.0
    if stagger_ms > 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1_000_000)
        e0.record()
        torch.cuda._sleep(20_000_000)
        e1.record()
        torch.cuda.synchronize()
        sleep_cycles_per_ms = 20_000_000 / e0.elapsed_time(e1)
    for b in passes_for(bc, args.warmup):
        bc.submit(b)
    torch.cuda.synchronize()
    shard.barrier(dry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stagger_ms > 0 and bc.depth > 1:     # context j starts j x stagger late (inside the timed region)
        for j in range(1, bc.depth):
            with torch.cuda.stream(pipe.g_streams[j % len(pipe.g_streams)]):
                torch.cuda._sleep(int(j * stagger_ms * sleep_cycles_per_ms))
    pend = [bc.submit(b) for b in passes_for(bc, K)]
    torch.cuda.synchronize()
    # poses of step s = rows [(s % group) * B_step, +B_step) of pass s // group; ONE all_gather of all K steps (still timed)
    per_step = []
    for s_ in range(K):
        p_, r0, r1 = shard.step_rows(s_, bc.group, bc.B_step)
        per_step.append(pend[p_].pose[r0:r1])
    local = torch.stack(per_step, dim=1).contiguous()
    gathered = shard.gather_poses(local, bc.step_total, dry)                    # [step_total, K, N, 9]
    torch.cuda.synchronize()
    shard.barrier(dry)
    torch.cuda.synchronize()
    dt = shard.max_over_ranks(time.perf_counter() - t0, bc.dev, dry)
    if dry:      # the gather of a one-rank group must hand back exactly the local rows
        assert torch.equal(gathered, local), "RCCL all_gather (world 1) changed the poses"
    for e in bc.engines:
        e.check_async()
    assert gathered.shape[0] == bc.step_total and gathered.shape[1] == K
    # every guided step must have run its full 700 iterations (no data-dependent early exit skipped work)
    iters = torch.cat([p.stats[:, :, :, 1].sum(dim=(0, 2)).cpu() for p in pend])
    return {"dt": dt, "pend": pend, "iters": iters, "finite": bool(torch.isfinite(gathered).all().item()),
            "in_pipe_ms": L.in_pipe_launches([p.stamps for p in pend if getattr(p, "stamps", None) is not None]),
            "in_pipe_busy": L.in_pipe_busy([p.stamps for p in pend if getattr(p, "stamps", None) is not None])}


def main():
    args = parse_args()
    # stdout carries exactly one line, the JSON result: RCCL prints its version banner to file descriptor 1 when the first communicator is
    # made, so descriptor 1 points at stderr until the result is ready
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    dry = bool(args.dry_dist) and int(os.environ.get("WORLD_SIZE", "1")) == 1
    rank, world, local = shard.init_distributed(backend="nccl" if dry else None, force=dry)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the sampling path has no CPU fallback)")
    if os.environ.get("PD_DIST_BACKEND") == "gloo":
        local = 0                      # functional check: several ranks share the one GPU of the box (shard.init_distributed)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    bc = setup(args, rank, world, dev)
    K, EB, depth, cfg = bc.K, bc.EB, bc.depth, bc.cfg
    torch.cuda.synchronize()
    # setup: every context captures its hipGraphs (full engine batch, and the tail batch of a K that is no multiple of `group`)
    L.capture_all(bc, sorted(set(passes_for(bc, K)) | set(passes_for(bc, max(args.warmup, 1))) | {EB}))

    tr = timed_region(bc, dry)
    dt, pend, iters = tr["dt"], tr["pend"], tr["iters"]
    if args.trace and rank == 0:
        for i, (a, b2, c, d) in enumerate(bc.pipe.timeline()):
            print(f"  sub {i:2d}: U {a:7.1f} -> {b2:7.1f} ({b2 - a:5.1f})   G {c:7.1f} -> {d:7.1f} ({d - c:5.1f})", file=sys.stderr)
    ms_per_step = dt / K * 1e3
    value = bc.step_total * K / dt

    # ---- everything below is outside the timed region
    pass_latency_ms, full_pose = L.pass_latency(bc)
    cold = L.cold_single_batch(bc)
    n_passes = len(passes_for(bc, K))
    fast = L.exact_mode(bc, full_pose, n_passes) if (not args.no_fast_mode and EB * N_FRAMES >= PD_STREAM_MIN_ROWS) else None
    fresh = L.fresh_inputs(bc, full_pose, n_passes) if bc.want_fresh else None
    slots_equal = L.headline_slots_equal_alone(bc, full_pose) if rank == 0 else None
    roofline, ggs_alone_ms = L.roofline_ggs(bc, full_pose, tr["in_pipe_ms"], tr["in_pipe_busy"])
    roofline_den, den_ms = L.roofline_denoiser(bc)
    ranks = L.rank_emulation(bc) if (rank == 0 and world == 1 and not args.no_rank_emulation and bc.strong and EB >= 160) else None
    per_config = L.per_config(bc) if (not args.no_per_config and rank == 0) else None      # (closes the extra contexts: keep it after the legs that use them)
    images = None
    if not args.no_from_images and rank == 0 and world == 1:
        try:
            images = L.from_images(bc)
        except Exception as e:  # noqa: BLE001  (the headline must still be reported)
            images = {"error": repr(e)}

    M = N_FRAMES * (N_FRAMES - 1) // 2 * PER_PAIR
    k_eff = bc.wgs or 24
    out = {
        "metric": "sequences/sec (20-frame, GGS on)", "value": value, "unit": "sequences/s", "n_gpus": world,
        "steps": K, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32" if EB * N_FRAMES < PD_STREAM_MIN_ROWS else
                 "f32 (GGS, attention, LayerNorm, DDPM update, every accumulation); the encoder GEMMs multiply fp16 hi + lo operand pairs (22 bits, "
                 "static power-of-two scales) with fp32 accumulation -- `exact_mode` is the same pipe with them on the exact-fp32 matrix instruction",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: one step = one batch of {bc.step_total} independent 20-frame sequences"
                        + (f" block-partitioned over {world} GPU(s) ({bc.B_step} per GPU and step; a GPU runs the shards of {bc.group} consecutive "
                           f"steps as one engine pass of {EB} sequences)" if bc.strong else f" ({args.seqs_per_step} per GPU)")
                        + f", {depth} engine passes in flight per GPU; 100 DDPM steps, GGS on for the last {COND_START} steps "
                        f"(7000 iterations/sequence), M={M} matches/sequence (190 pairs x {PER_PAIR}), {IMG}x{IMG}; random-init reference-rule "
                        "weights, matches epipolar-consistent with the engine's own unguided model mean at t=9; inputs resident in HBM",
            "sequences_per_step": bc.step_total, "sequences_per_gpu_per_step": bc.B_step, "steps_per_engine_pass": bc.group, "sequences_per_engine_pass": EB,
            "engine_passes_in_timed_region": len(pend), "sequences_in_flight_per_gpu": EB * depth, "frames": N_FRAMES,
            "matches_per_sequence": M, "diffusion_steps": 100, "ggs_iterations_per_sequence_run": float(iters.min().item()),
            "hip_graph": bc.use_graph, "pipeline_depth": depth, "guided_slots": bc.slots, "unguided_streams": max(0, args.unguided_streams),
            "ggs_workgroups_per_sequence": k_eff, "engine_pass_latency_ms_unpipelined": pass_latency_ms,
            "parallelism": f"dp{world} (independent sequences, one final all_gather)", "outputs_finite": tr["finite"],
            "headline_slots_equal_alone": slots_equal, "headline_slots_checked": bc.check_slots,
            "multi_gpu": "no hardware scaling curve was measured by the builder (every lease had one GPU); `rank_emulation` holds the predicted one",
        },
        "roofline": roofline,
        "roofline_denoiser": roofline_den,
        "per_step_ms": {"denoiser_step": den_ms, "ggs_guided_step": roofline["launch_ms"], "ggs_guided_step_alone": ggs_alone_ms,
                        "ggs_iteration_us": roofline["launch_ms"] * 1e3 / (7 * cfg.iter_num)},
    }
    if cold is not None:
        out["cold_single_batch"] = cold
    if dry:
        out["dry_dist"] = {"backend": torch.distributed.get_backend(), "world": 1,
                           "executed": "barrier, all_gather of the poses, all_reduce(MAX) of the time -- the N > 1 call sites on RCCL with one rank; "
                                       "RCCL ACROSS GPUs over xGMI is still unexecuted (no multi-GPU box was available to the builder)"}
    for key, val in (("exact_mode", fast), ("fresh_inputs", fresh), ("rank_emulation", ranks), ("from_images", images), ("per_config", per_config)):
        if val is not None:
            out[key] = val
    if rank == 0:
        skipped = {"value": None, "unit": "sequences/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port"}
        if args.cpu_budget_s > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_budget_s)
            except Exception as e:  # noqa: BLE001  (the GPU number must still be reported)
                out["cpu_baseline"] = dict(skipped, sample=f"failed: {e!r}")
        else:
            out["cpu_baseline"] = dict(skipped, sample="skipped (measured on rank 0 at N=1 only)")
        sys.stdout.flush()
        C.CDLL(None).fflush(None)           # RCCL's banner sits in libc's stdout buffer: out with it while descriptor 1 is still stderr
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1 or dry:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
