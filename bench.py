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
  roofline        dominant kernel (pd_ggs_kernel<..>): `frac` = algorithmic fp32 FLOPs of ONE launch / its hipEvent-timed
                  duration / 157.3 TFLOP/s (SURVEY 8d prices it against the fp32 vector ALU); sub-objects `co_resident`
                  (the `--pipeline-depth` launches that run together, as in the pipe) and `fabric` (match bytes streamed per
                  iteration -- Infinity-Cache / fabric bandwidth, NOT HBM); `traffic` from the committed PMC summary, only
                  when that summary was collected with THIS libpd_engine.so (sha256 recorded there)
  roofline_denoiser
  per_config      BASELINE configs[1], [2], [3]-shard and [4], each alone on the chip (ms per pass, sequences/s)
  exact_mode      the same pipe with the denoiser's encoder GEMMs on the exact-fp32 matrix instruction (PD_OPT_DENOISER_SPLIT = 0)
                  instead of the default fp16 hi + lo operand pairs (three fp16 MFMA products, fp32 accumulation)
  fresh_inputs    the same pipe with every pass uploading NEW z / noise / matches inside the timed region
                  (pinned host -> device copies + asynchronous device-side match ingestion)
  cold_single_batch  ONE batch of 64 sequences alone on an idle chip (latency, sequences/s): what `value`'s steady-state rate is NOT
  dry_dist        (--dry-dist on a single-GPU box) the N > 1 collectives executed on RCCL with a world of one rank
  cpu_baseline    the reference files verbatim (kind "reference") when the reference tree is present, else the oracle
                  port (kind "port"), on the host cores of this box, bounded sample, all five GGS stage types timed
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from posediffusion_amd import shard, synth  # noqa: E402

N_FRAMES = 20
IMG = 224
PER_PAIR = 300
STEP_SEQS = 64                       # BASELINE configs[3]: one batch of 64 independent sequences
COND_START = 10                      # cfgs/default.yaml:8
FLOP_PER_MATCH_ITER = 100.0          # 36 fwd + 64 bwd
DENOISER_PARAMS = 17_298_697         # fp32 -> 69.19 MB read per denoiser step
DENOISER_MFLOP_PER_TOKEN = 34.73     # SURVEY.md section 8(d), N = 20
PD_STREAM_MIN_ROWS = 1024            # csrc/pd_gemm_stream.h
FP32_PEAK_TFLOPS = 157.3             # MI355X fp32 vector ALU = dense fp32 MFMA peak (MI355X_MICROARCH.md)
F16_PEAK_TFLOPS = 2500.0            # dense fp16 / bf16 MFMA peak (same guide; the sparsity figure is never used)
HBM_PEAK_GBS = 8000.0
MATCH_BYTES = 16                     # kp1, kp2: 2 x float2 per match (pair indices are per work item)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round5_pmc_summary.json")


# ------------------------------------------------------------------------------------------------------------ inputs
def make_batch_inputs(eng, diff, B, dev, seed0, n_frames=N_FRAMES, img=IMG, per_pair=PER_PAIR, keep_host=False, upload=True):
    """Synthetic inputs for B sequences (seeds seed0 .. seed0+B-1): z, reference-order noise, and matches that are
    epipolar-consistent with the engine's own unguided model mean at the first guided step (so every guided step runs its
    full 700 iterations, as it does with a trained checkpoint and real SuperGlue matches).  Matches go to the engine's
    slots; with keep_host the per-sequence matches_dicts are returned too (fresh-inputs mode packs them into pinned memory)."""
    from posediffusion_amd.host import draw_noise
    T = diff.num_timesteps
    z = torch.cat([synth.make_z(1, n_frames, seed=1000 + seed0 + b) for b in range(B)]).to(dev)
    noise = torch.empty(T + 1, B, n_frames, 9, device=dev)
    for b in range(B):
        g = torch.Generator(device=dev).manual_seed(seed0 + b)            # cfg.seed (+ sequence index)
        noise[:, b] = draw_noise((n_frames, 9), T, dev, COND_START, True, generator=g)
    _, process, _ = eng.sample(z, noise, 0, None, use_graph=False)         # unguided run -> what GGS first sees
    x_at = process[T - COND_START]                                         # x_t for t = COND_START-1
    mean, _ = eng.p_mean(x_at, z, COND_START - 1)
    mean = mean.cpu().numpy().astype(np.float64)
    mds = []
    for b in range(B if (upload or keep_host) else 0):
        md = synth.make_epipolar_matches(mean[b], img, img, per_pair, seed=2000 + seed0 + b)
        if upload:
            eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        if keep_host is True or (keep_host and b in keep_host):      # True: every slot; a collection: those slots (None elsewhere)
            mds.append(md)
        elif keep_host:
            mds.append(None)
    return z, noise, mds


def lib_sha256():
    from posediffusion_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def pmc_traffic(which, eb=None):
    """Fabric-side traffic per launch from the committed rocprofv3 PMC summary (counters need their own `rocprofv3 --pmc`
    passes: tools/collect_pmc.sh).  Valid only for the binary it was collected with: the summary records the sha256 of
    libpd_engine.so and a mismatch voids it LOUDLY (stderr + reason in the JSON) instead of quoting a stale number."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, f"no PMC summary ({e.__class__.__name__}): run tools/collect_pmc.sh on the GPU box"
    have, want = d.get("libpd_engine_sha256"), lib_sha256()
    if have != want:
        msg = (f"STALE: {os.path.relpath(PMC_SUMMARY, ROOT)} was collected with libpd_engine.so sha256 {str(have)[:12]}..., the "
               f"running library is {want[:12]}...: traffic not reported (re-run tools/collect_pmc.sh)")
        print("bench.py: " + msg, file=sys.stderr)
        return None, msg
    if eb is not None and d.get("batch_sequences") != eb:
        return None, f"{os.path.relpath(PMC_SUMMARY, ROOT)} is for an engine batch of {d.get('batch_sequences')} sequences, this run uses {eb}"
    return d[which]["traffic_bytes_corrected"], (
        f"{os.path.relpath(PMC_SUMMARY, ROOT)}: (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, separate --pmc passes, gfx950 "
        "FETCH_SIZE x2 correction; fabric-side counters (Infinity-Cache hits are counted: an upper bound on HBM bytes); same "
        "library hash as this run")


def lane_stream_fraction(pair_sizes, lanes=512, resident_steps=17):
    """Share of the algorithmic match bytes the lane-per-item GGS kernel pulls through the fabric per iteration (reporting only; the rule
    is pd_ggs_set_matches' in csrc/pd_ggs.hip: the smallest item length that leaves <= `lanes` lane items, k = 1..3 more cuts for the spare / k - d
    pairs with the longest items ((k, d) by the modelled match pass), items ordered by length, 64 per wave, a wave's stream padded to its longest item; the first
    `resident_steps` steps (two matches per lane each) of every wave live on chip for the whole launch: 14 in registers + 3 in LDS, PD_LANE_RV + PD_LANE_RL)."""
    ms = [m for m in pair_sizes if m > 0]
    lo, hi = 1, max(ms)
    while lo < hi:
        mid = (lo + hi) // 2
        if sum(-(-m // mid) for m in ms) <= lanes:
            hi = mid
        else:
            lo = mid + 1
    base = [-(-m // lo) for m in ms]
    spare = lanes - sum(base)
    order = sorted(range(len(ms)), key=lambda p: (-(-(-ms[p] // base[p])), p))
    rank = {p: r for r, p in enumerate(order)}

    def cuts(k, d):                                                        # k more cuts for the spare // k - d pairs with the longest items
        return [base[p] + (min(k, ms[p] - base[p]) if ms[p] > base[p] and rank[p] < spare // k - d else 0) for p in range(len(ms))]

    def wave_steps(nch):
        st = sorted(((-(-ms[p] // nch[p]) + 1) // 2 for p in range(len(ms)) for _ in range(nch[p])), reverse=True)
        return st, [st[w] for w in range(0, len(st), 64)], [st[min(w + 63, len(st) - 1)] for w in range(0, len(st), 64)]

    def cost(nch):                                                         # pd_lane_pass_cost (csrc/pd_internal.h): waves w and w + 4 share a SIMD
        _, tmax, tmin = wave_steps(nch)
        t = [100 * a + 15 * (a - b) for a, b in zip(tmax, tmin)] + [0] * (8 - len(tmax))
        return max(max(t[s], (45 * t[s]) // 100 + t[s + 4]) for s in range(4))

    cands = [cuts(k, d) for k in (1, 2, 3) for d in range(16) if d == 0 or spare // k - d > 0]
    nch = min(cands, key=cost)                                             # ties: the smaller k, then the smaller d (min keeps the first)
    steps = wave_steps(nch)[0]
    waves = [steps[w] for w in range(0, len(steps), 64)]                   # a wave runs (and streams) as many steps as its longest item
    streamed = sum(max(t - resident_steps, 0) for t in waves) * 64 * 32     # bytes per iteration and sequence
    return streamed / (16.0 * sum(ms)), len(steps), waves


def stream_ceiling():
    """This box's ceiling for the lane kernel's OWN access pattern (tools/stream_probe.hip, `RING` rows; built by __graft_entry__.build()):
    one workgroup of 8 waves per CU, every wave streaming its contiguous share of a private region in 2 KiB steps through an LDS ring fed by
    global_load_lds_dwordx4 (rings of 4 / 6 / 8 steps in flight; 704 KB and 912 KB per CU and pass), with no arithmetic beside it.
    -> ((min, max) GB/s over those rows, (min, max) over the probe's plain-load 912 KB rows), source text; or (None, reason)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "stream_probe")
    if not os.path.isfile(exe):
        return None, "tools/stream_probe not built"
    try:
        txt = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180).stdout
    except Exception as e:  # noqa: BLE001
        return None, f"tools/stream_probe failed: {e!r}"
    ring, plain = [], []
    for line in txt.splitlines():
        f = line.split()
        try:
            if len(f) >= 5 and f[0] == "RING":
                ring.append(float(f[-1]) * 1e3)
            elif len(f) >= 5 and f[0] == "912" and f[1] == "KB":
                plain.append(float(f[-1]) * 1e3)
        except ValueError:
            pass
    if not ring:
        return None, "tools/stream_probe printed no RING rows"
    return ((min(ring), max(ring)), (min(plain), max(plain)) if plain else None), (
        "tools/stream_probe on THIS box, after the timed region: 256 workgroups x 8 waves, each wave streaming its share of a private 704 / 912 KB "
        "region through an LDS ring of 4 / 6 / 8 x 2 KiB fed by global_load_lds_dwordx4 -- the lane kernel's own pattern, without its arithmetic")


# ------------------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(budget_s: float):
    """The reference's own files executed in place (kind "reference") when the reference tree is present
    (PD_REFERENCE_ROOT or /root/reference), otherwise the oracle port (kind "port") -- on a bounded sample: denoiser steps
    at B = 1, N = 20 and GGS iterations at M = 57 000 for each of the four stage types (all / FL / R / T), extrapolated to
    100 steps + 10 x (400 all + 100 FL + 100 R + 100 T) iterations per sequence."""
    import contextlib
    import io
    from oracle import pd_oracle as O
    from oracle import ref_stubs as RS
    use_ref = RS.available()
    diff = synth.make_diffuser(seed=0)
    sd = O.cast_state_dict(diff.model.state_dict(), torch.float32)
    z = synth.make_z(1, N_FRAMES)
    x = torch.randn(1, N_FRAMES, 9, generator=torch.Generator().manual_seed(0))
    tt = torch.full((1,), 50, dtype=torch.long)
    if use_ref:
        ref = RS.load_reference()
        rdiff = RS.build_reference_diffuser(seed=0)
        den = lambda: rdiff.model(x, tt, z)                                           # noqa: E731  models/denoiser.py verbatim
    else:
        den = lambda: O.denoiser_forward(sd, x, tt, z)                                # noqa: E731
    host_cores = os.cpu_count()
    max_threads = torch.get_num_threads()
    # tiny GEMMs oversubscribe badly on a many-core host: probe a few thread counts on the denoiser, keep the fastest
    best = (float("inf"), max_threads)
    with torch.no_grad():
        for n in sorted({min(max_threads, c) for c in (4, 8, 16, 32, max_threads)}):
            torch.set_num_threads(n)
            den()
            t0 = time.time()
            for _ in range(3):
                den()
            best = min(best, ((time.time() - t0) / 3, n))
    threads = best[1]
    torch.set_num_threads(threads)
    with torch.no_grad():
        den()
        n_den, t0 = 0, time.time()
        while n_den < 3 or (time.time() - t0 < 0.2 * budget_s and n_den < 50):
            den()
            n_den += 1
        t_den = (time.time() - t0) / n_den
    enc = synth.make_cameras(N_FRAMES, seed=2000)
    md = synth.make_matches(enc, IMG, IMG, per_pair=PER_PAIR, seed=2000)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = synth.perturb_pose(enc, seed=7)
    stage_iters = {"all": 400, "fl": 100, "r": 100, "t": 100}                          # per guided step (:48-63, :86-87)
    flags = {"all": (True, True, True), "fl": (False, False, True), "r": (True, False, False), "t": (False, True, False)}
    t_stage, n_stage = {}, {}
    per_stage_budget = 0.8 * budget_s / 4

    def make_run(name, uR, uT, uF):
        if use_ref:
            pmr = {"kp1_homo": pm["kp1_homo"], "kp2_homo": pm["kp2_homo"], "i1": pm["i1"], "i2": pm["i2"], "h": IMG, "w": IMG,
                   "pair_idx": pm["pair_idx"]}
            return lambda n: ref.GGS_optimize(x0.clone(), 0, pmr, update_R=uR, update_T=uT, update_FL=uF,
                                              **dict(synth.GGS_CFG, iter_num=(n // 2 if name == "all" else n)))
        return lambda n: O.ggs_optimize(x0.clone(), pm, update_R=uR, update_T=uT, update_FL=uF, iter_num=(n // 2 if name == "all" else n))

    # the GGS iterations are 98 % of a sequence on the CPU and their operators (gathers and products over 57 000 matches) like another
    # thread count than the denoiser's 20-row GEMMs: probed separately on the `all` stage, the fastest kept for the four stages
    probe = make_run("all", True, True, True)
    best_g = (float("inf"), threads)
    with contextlib.redirect_stdout(io.StringIO()):
        for n in sorted({min(max_threads, c) for c in (4, 8, 16, 32, 64, max_threads)}):
            torch.set_num_threads(n)
            probe(2)
            t0 = time.time()
            probe(4)
            best_g = min(best_g, ((time.time() - t0) / 4, n))
    den_threads, threads = threads, best_g[1]
    torch.set_num_threads(threads)
    for name, (uR, uT, uF) in flags.items():
        n_it = 2
        run = make_run(name, uR, uT, uF)
        with contextlib.redirect_stdout(io.StringIO()):
            run(2)                                                                     # warm-up
            t0 = time.time()
            run(n_it)
            dt = time.time() - t0
            n_it = int(max(2, min(400, per_stage_budget / max(dt / n_it, 1e-4)))) // 2 * 2      # fills the budget (~ 0.2 budget_s per stage)
            t0 = time.time()
            run(n_it)
            dt = time.time() - t0
        t_stage[name], n_stage[name] = dt / n_it, n_it
    t_guided_step = sum(stage_iters[k] * t_stage[k] for k in stage_iters)
    t_seq = 100 * t_den + COND_START * t_guided_step
    torch.set_num_threads(max_threads)
    what = ("the reference files executed in place (models/denoiser.py, util/geometry_guided_sampling.py + restated pytorch3d "
            "helpers, oracle/ref_stubs.py)" if use_ref else "oracle/pd_oracle.py (torch-CPU restatement of the reference path)")
    return {"value": 1.0 / t_seq, "unit": "sequences/s", "cores": threads, "host_cores": host_cores,
            "kind": "reference" if use_ref else "port",
            "sample": f"{what}, torch {torch.__version__} CPU, {threads} threads for the GGS iterations and {den_threads} for the denoiser "
                      f"(each the fastest of 4/8/16/32/64/all on its own operator; the host has {host_cores} cores; `cores` = the GGS figure, "
                      f"98 % of the time): {n_den} denoiser steps (B=1, N=20): {t_den * 1e3:.1f} ms/step; GGS iterations at "
                      f"M=57000: " + ", ".join(f"{k} x{n_stage[k]}: {t_stage[k] * 1e3:.1f} ms/it" for k in stage_iters) +
                      "; extrapolated to 100 steps + 10 x (400 all + 100 FL + 100 R + 100 T) iterations per sequence"}


# ------------------------------------------------------------------------------------------------------------ per-config
def measure_config(diff, dev, B, n_frames, img, ggs_on, reps=3):
    """One BASELINE config alone on the chip: ms per pass (hipGraph replay, inputs resident), sequences/s."""
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B,
                     max_N=n_frames)
    z, noise, mds = make_batch_inputs(eng, diff, B, dev, seed0=7000, n_frames=n_frames, img=img, upload=ggs_on, keep_host=(ggs_on and B == 1))
    cfg = make_ggs_cfg(synth.GGS_CFG) if ggs_on else None
    cs = COND_START if ggs_on else 0
    eng.sample(z, noise, cs, cfg, use_graph=True, want_process=False)              # captures
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        _, _, st = eng.sample(z, noise, cs, cfg, use_graph=True, want_process=False)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    eng.check_async()
    iters = float(st[:, :, :, 1].sum(dim=(0, 2)).min().item()) if ggs_on else 0.0
    ms = min(times) * 1e3
    out = {"B": B, "frames": n_frames, "image": img, "ggs": ggs_on, "matches_per_sequence": n_frames * (n_frames - 1) // 2 * PER_PAIR if ggs_on else 0,
           "ms_per_pass": ms, "sequences_per_s": B / (ms * 1e-3), "ggs_iterations_per_sequence_run": iters}
    if ggs_on:
        g = eng.time_kernel(1, B, n_frames, cfg, reps=2)
        out["ggs_guided_step_ms"] = g
        out["ggs_iteration_us"] = g * 1e3 / 700
    out["denoiser_step_us"] = eng.time_kernel(0, B, n_frames, cfg if ggs_on else make_ggs_cfg(synth.GGS_CFG), reps=20) * 1e3
    eng.close()
    if B == 1:
        # the seam a user of the reference calls (models/gaussian_diffuser.py:284-306): GaussianDiffusion.sample(shape, z, cond_fn, cond_start_step)
        # of the drop-in module, END TO END -- the noise drawn in the reference's order by torch's generator, the matches_dict (numpy, as demo.py
        # holds it) recognised and uploaded (cached after the first call, like the reference's five calls per guided step share one dict), the
        # graph replayed, the `t=.. | sampson=..` lines printed, the result synchronised.  `ms_per_pass` above feeds resident, pre-drawn noise.
        import contextlib
        import functools
        import io
        synth._dropin()
        from util.geometry_guided_sampling import geometry_guided_sampling
        cond_fn, md = None, None
        if ggs_on:
            md = mds[0]
            cond_fn = functools.partial(geometry_guided_sampling, matches_dict=md, GGS_cfg=dict(synth.GGS_CFG))
        zs = z[:1]
        ts = []
        for rep in range(4):                                    # the first call builds the engine for these modules, uploads and captures
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                pose, _ = diff.sample([1, n_frames, 9], zs, cond_fn=cond_fn, cond_start_step=cs)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        out["dropin_sample_ms"] = min(ts[1:])
        out["dropin_sample_first_call_ms"] = ts[0]
        out["dropin_sample_finite"] = bool(torch.isfinite(pose).all().item())
    return out


# ------------------------------------------------------------------------------------------------------------ main
def main():
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
    ap.add_argument("--pipeline-depth", type=int, default=3,
                    help="engine contexts / HIP streams per GPU (engine passes in flight); 1 = serial")
    ap.add_argument("--unguided-streams", type=int, default=0,
                    help="0 = every context's stream runs whole passes; u > 0 = the two-stage pipe of SamplingPipeline: u streams run the unguided "
                         "halves, --ggs-slots streams the guided halves (their GGS launches then never meet another slot's)")
    ap.add_argument("--ggs-slots", type=int, default=0, help="streams that run guided halves (0 = one per context)")
    ap.add_argument("--trace", action="store_true", help="print the pipeline timeline (per-pass phase times) to stderr")
    ap.add_argument("--no-per-config", action="store_true", help="skip the per-BASELINE-config measurements")
    ap.add_argument("--no-fast-mode", "--no-exact-mode", dest="no_fast_mode", action="store_true",
                    help="skip the comparison run with the encoder GEMMs on the exact-fp32 matrix instruction")
    ap.add_argument("--no-fresh-inputs", action="store_true", help="skip the fresh-inputs (upload inside the timed region) measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0, help="CPU seconds for the cpu_baseline sample (0 = skip)")
    ap.add_argument("--dry-dist", action="store_true",
                    help="single-GPU box: initialise RCCL (backend nccl) with a world of ONE rank and run the barrier / all_gather / "
                         "max-over-ranks call sites of the N > 1 path anyway, so that they execute at least once on GPU hardware")
    ap.add_argument("--no-stream-probe", action="store_true", help="skip tools/stream_probe (this box's match-stream ceiling)")
    args = ap.parse_args()

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

    from posediffusion_amd import _lib
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state, get_engine, pack_matches
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
    check_slots = sorted({0, min(EB - 1, ((EB - 1) // 3) | 1), EB - 1})                       # headline_slots_equal_alone: these slots of context 0 are re-run alone
    inputs = [make_batch_inputs(engines[j], diff, EB, dev, seed0=100_000 * rank + j * EB, keep_host=True if want_fresh else (check_slots if j == 0 else False))
              for j in range(depth)]
    wgs = args.ggs_wgs if args.ggs_wgs > 0 else pipe.wgs_per_seq(EB)
    flags = int(os.environ.get("PD_GGS_RESERVED", "0"))                                                          # A/B switch, pd_engine.h
    if wgs == 1 and not (flags & _lib.PD_GGS_CFG_NO_LANE_ITEMS):
        flags |= _lib.PD_GGS_CFG_LANE_ITEMS       # one workgroup per sequence: the lane-per-item kernel (SamplingPipeline.make_cfg does the same)
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs, reserved=flags)
    use_graph = not args.no_graph
    torch.cuda.synchronize()

    # the passes of a K-step run: pass p covers steps [p*group, min(K, (p+1)*group)) -> that many shards of B_step sequences
    def passes_for(k_steps):
        return [min(group, k_steps - s0) * B_step for s0 in range(0, k_steps, group)]

    # setup: every context captures its hipGraphs (full engine batch, and the tail batch of a K that is no multiple of `group`)
    shapes = sorted(set(passes_for(K)) | set(passes_for(max(args.warmup, 1))) | {EB})
    sliced = {b: [(inputs[j][0][:b].contiguous(), inputs[j][1][:, :b].contiguous()) for j in range(depth)] for b in shapes if b != EB}

    def submit(b_seqs):                                                         # resident tensors per shape: no copies in the timed region
        j = pipe.next_context()
        z, noise = (inputs[j][0], inputs[j][1]) if b_seqs == EB else sliced[b_seqs][j]
        return pipe.submit(z, noise, COND_START, cfg, use_graph=use_graph, want_process=False)

    for j in range(depth):
        for b in shapes:
            z, noise = (inputs[j][0], inputs[j][1]) if b == EB else sliced[b][j]
            with torch.cuda.stream(pipe.u_stream):
                out = engines[j].sample(z, noise, COND_START, cfg, use_graph=use_graph, want_process=False, phase=1)
                engines[j].sample(z, noise, COND_START, cfg, use_graph=use_graph, want_process=False, phase=2, out=out)
            torch.cuda.synchronize()
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
    for b in passes_for(args.warmup):
        submit(b)
    torch.cuda.synchronize()
    shard.barrier(dry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stagger_ms > 0 and depth > 1:     # context j starts j x stagger late (inside the timed region): the contexts' guided halves
        for j in range(1, depth):        # then meet the other contexts' unguided halves instead of each other
            with torch.cuda.stream(pipe.g_streams[j % len(pipe.g_streams)]):
                torch.cuda._sleep(int(j * stagger_ms * sleep_cycles_per_ms))
    pend = [submit(b) for b in passes_for(K)]
    torch.cuda.synchronize()
    # poses of step s = rows [(s % group) * B_step, +B_step) of pass s // group; ONE all_gather of all K steps (still timed)
    per_step = []
    for s_ in range(K):
        p_, r0, r1 = shard.step_rows(s_, group, B_step)
        per_step.append(pend[p_].pose[r0:r1])
    gathered = shard.gather_poses(torch.stack(per_step, dim=1).contiguous(), step_total, dry)     # [step_total, K, N, 9]
    torch.cuda.synchronize()
    shard.barrier(dry)
    torch.cuda.synchronize()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dev, dry)
    if dry:      # the gather of a one-rank group must hand back exactly the local rows
        assert torch.equal(gathered, torch.stack(per_step, dim=1)), "RCCL all_gather (world 1) changed the poses"
    for e in engines:
        e.check_async()
    if args.trace and rank == 0:
        for i, (a, b2, c, d) in enumerate(pipe.timeline()):
            print(f"  sub {i:2d}: U {a:7.1f} -> {b2:7.1f} ({b2 - a:5.1f})   G {c:7.1f} -> {d:7.1f} ({d - c:5.1f})", file=sys.stderr)
    assert gathered.shape[0] == step_total and gathered.shape[1] == K
    ms_per_step = dt / K * 1e3
    value = step_total * K / dt
    # every guided step must have run its full 700 iterations (no data-dependent early exit skipped work)
    iters = torch.cat([p.stats[:, :, :, 1].sum(dim=(0, 2)).cpu() for p in pend])
    finite = bool(torch.isfinite(gathered).all().item())

    # un-overlapped latency of one engine pass and of ONE sequence (outside the timed region, reported next to the throughput)
    z, noise, _ = inputs[0]
    for rep in range(2):        # the first call captures the whole-loop graph
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(pipe.u_stream):
            full_pose = engines[0].sample(z, noise, COND_START, cfg, use_graph=use_graph, want_process=False)[0]
        torch.cuda.synchronize()
        pass_latency_ms = (time.perf_counter() - t1) * 1e3

    # ONE batch of 64 sequences alone on an idle chip (what configs[3] literally names), reported next to the streaming figure
    cold = None
    if EB >= STEP_SEQS:
        zc, nc = z[:STEP_SEQS].contiguous(), noise[:, :STEP_SEQS].contiguous()
        # the launch shape a caller with ONE batch in flight gets (SamplingPipeline.wgs_per_seq with one context: CUs // sequences = 4
        # workgroups per sequence on the wave-per-item kernels, 8.3 ms per GGS launch against 13.0 ms for one lane-kernel workgroup per
        # sequence on a quarter of the chip); the streaming shape's figure is reported beside it
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        cfg_cold = type(cfg).from_buffer_copy(cfg)
        cfg_cold.wgs_per_seq = max(1, cus // STEP_SEQS) if not args.ggs_wgs else args.ggs_wgs
        cfg_cold.reserved = 0
        lat_by = {}
        for tag, c in (("alone", cfg_cold), ("streaming_shape", cfg)):
            lat = []
            for rep in range(3):    # the first call captures this shape's graph
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                with torch.cuda.stream(pipe.u_stream):
                    engines[0].sample(zc, nc, COND_START, c, use_graph=use_graph, want_process=False)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
            lat_by[tag] = min(lat[1:])
        cold = {"sequences": STEP_SEQS, "latency_ms": lat_by["alone"], "sequences_per_s": STEP_SEQS / (lat_by["alone"] * 1e-3),
                "ggs_workgroups_per_sequence": int(cfg_cold.wgs_per_seq),
                "latency_ms_with_the_streaming_launch_shape": lat_by["streaming_shape"],
                "note": "one batch of 64 sequences, nothing else in flight: the latency of a single configs[3] batch with the launch shape a caller "
                        f"with one batch in flight gets ({int(cfg_cold.wgs_per_seq)} GGS workgroups per sequence fill the chip); with the streaming shape "
                        "(one workgroup per sequence: 64 of the 256 CUs busy) beside it; `value` is the steady-state rate with "
                        + str(EB * depth) + " sequences in flight"}

    # ---- exact mode (reported next to `value`): the same pipe with the encoder GEMMs on the exact-fp32 matrix instruction
    # (PD_OPT_DENOISER_SPLIT = 0) instead of the default fp16-plane kernels -- what rounds 1 and 2 reported as `value`
    fast = None
    if not args.no_fast_mode and EB * N_FRAMES >= PD_STREAM_MIN_ROWS:
        den_default_ms = eng.time_kernel(0, EB, N_FRAMES, cfg, reps=20)
        for e in engines:
            e.set_split_precision(0)
        for j in range(depth):                                                  # capture the exact-mode graphs
            with torch.cuda.stream(pipe.u_stream):
                out = engines[j].sample(inputs[j][0], inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False, phase=1)
                engines[j].sample(inputs[j][0], inputs[j][1], COND_START, cfg, use_graph=use_graph, want_process=False, phase=2, out=out)
            torch.cuda.synchronize()
        for _ in range(depth):
            submit(EB)
        torch.cuda.synchronize()
        n_fast = max(depth, min(len(passes_for(K)), 4 * depth))
        t3 = time.perf_counter()
        pfm = [submit(EB) for _ in range(n_fast)]
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        den_fast_ms = eng.time_kernel(0, EB, N_FRAMES, cfg, reps=20)
        itf = torch.cat([p.stats[:, :, :, 1].sum(dim=(0, 2)).cpu() for p in pfm])
        ctx0 = [p for p in pfm if p.context == 0]
        fast = {"value": EB * n_fast / dt3, "unit": "sequences/s on this GPU", "passes": n_fast,
                "dtype": "f32 everywhere, the encoder GEMMs on v_mfma_f32_32x32x2_f32 (PD_OPT_DENOISER_SPLIT = 0)",
                "denoiser_step_us_alone": den_fast_ms * 1e3, "denoiser_step_us_alone_default_mode": den_default_ms * 1e3,
                "ggs_iterations_per_sequence_run": float(itf.min().item()),
                "outputs_finite": bool(all(torch.isfinite(p.pose).all().item() for p in pfm[-depth:])),
                "pose_rel_deviation_from_the_default_mode_after_the_full_guided_pass": (
                    float(((ctx0[0].pose - full_pose).abs().max() / full_pose.abs().max()).item()) if ctx0 else None),
                "note": "the default (`value`) runs the four Linear layers of each encoder layer as fp16 hi + fp16 lo operands (22 "
                        "mantissa bits, power-of-two scales from static bounds), three fp16 MFMA products, fp32 accumulation: per-step error "
                        "against fp64 8e-7 .. 1.1e-6 (this exact mode: 1.0e-6 .. 1.1e-6), 100 free-running steps 6.43e-4 mean deviation "
                        "from fp64 over 52 sequences (exact mode: 6.43e-4) -- tests/test_gpu_parity_r3.py::"
                        "test_fp16_plane_denoiser_mode_is_fp32_grade, profiles/round3_fp16_plane_mode_study.json"}
        for e in engines:
            e.set_split_precision(2)

    # ---- fresh inputs: every pass brings NEW z / noise / matches from pinned host memory inside the timed region
    fresh = None
    if want_fresh:
        sets = []
        for j in range(min(2, depth)):
            zc, nc, mds = inputs[j]
            kp1, kp2, i12, off, shape = pack_matches(mds, pin=True)
            sets.append((zc.cpu().pin_memory(), nc.cpu().pin_memory(), kp1, kp2, i12, off, shape))
        hints = dict(max_pairs=N_FRAMES * (N_FRAMES - 1) // 2, max_matches_per_pair=PER_PAIR, one_order=True)
        staging = [tuple(torch.empty_like(t, device=dev) for t in sets[0][:5]) for _ in range(depth)]
        up_bytes = sum(t.numel() * t.element_size() for t in sets[0][:5])

        def submit_fresh(i):
            j = pipe.next_context()
            src = sets[i % len(sets)]
            with torch.cuda.stream(pipe.next_stream()):
                for dst, s_ in zip(staging[j], src[:5]):
                    dst.copy_(s_, non_blocking=True)                            # pinned host -> device on the pass's stream
                engines[j].set_matches_async(0, staging[j][2], staging[j][3], staging[j][4], src[5], src[6], **hints)
            return pipe.submit(staging[j][0], staging[j][1], COND_START, cfg, use_graph=use_graph, want_process=False)

        for i in range(depth):
            submit_fresh(i)                                                     # warm-up (allocates the slot buffers, captures the graph of the device-built plan)
        torch.cuda.synchronize()
        n_fresh = max(depth, min(len(passes_for(K)), 3 * depth))
        t2 = time.perf_counter()
        pf = [submit_fresh(i) for i in range(n_fresh)]
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        for e in engines:
            e.check_async()
        it2 = torch.cat([p.stats[:, :, :, 1].sum(dim=(0, 2)).cpu() for p in pf])
        # pass 0 of the fresh run carries the data of context 0's resident batch: the same bits are expected whichever context
        # runs it (identical engines; the device-built match tables sort exactly like the host-built ones)
        same = bool(torch.equal(pf[0].pose, full_pose))
        fresh = {"value": EB * n_fresh / dt2, "unit": "sequences/s on this GPU", "passes": n_fresh, "sequences_per_pass": EB,
                 "uploaded_bytes_per_pass": up_bytes, "upload": "pinned host -> device copy of z, noise, kp1, kp2 (fp64), i12 (int64) on the pass's "
                 "stream + pd_ggs_set_matches_csr_async (device-side stable sort and table build, no host synchronisation)",
                 "ggs_iterations_per_sequence_run": float(it2.min().item()),
                 "first_pass_bitwise_equals_resident_pass": same,
                 "note": "two distinct pre-packed input sets alternate; packing into pinned memory (the data producer's side) is outside the timed region"}
        # back to each context's own resident batch (host-built tables) for the roofline legs below
        for j in range(depth):
            for b, md in enumerate(inputs[j][2]):
                engines[j].set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])

    # ---- the headline kernel pinned at the headline launch: a guided step of the full engine batch -- the launch
    # shape of the timed region: EB workgroups, the lane-per-item kernel -- against the same sequences run ALONE (one workgroup on an idle
    # chip) on a second, single-slot engine: bit for bit, all 700 iterations.  (tests/test_gpu_parity_r5.py does this against the oracle too.)
    slots_equal = None
    if rank == 0:
        solo = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=1, max_N=N_FRAMES)
        big, big_st = eng.ggs_guide(full_pose, 0, cfg)
        eng.check_async()
        slots_equal = True
        for b in check_slots:
            md = inputs[0][2][b]
            solo.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
            one, one_st = solo.ggs_guide(full_pose[b:b + 1], 0, cfg)
            solo.check_async()
            slots_equal = slots_equal and bool(torch.equal(one[0], big[b])) and bool(torch.equal(one_st[0], big_st[b])) \
                and float(one_st[0, :, 1].sum().item()) == 7.0 * cfg.iter_num
        solo.close()
        assert slots_equal, f"slots {check_slots} of the {EB}-sequence GGS launch differ from the same sequences run alone"

    # ---- roofline of the dominant kernel + the denoiser step, timed with hipEvents on the launch stream
    eng.time_kernel(1, EB, N_FRAMES, cfg, reps=2)             # warm: the timed launches below start on a busy chip (clocks up), as in the pipe
    ggs_each = [eng.time_kernel(1, EB, N_FRAMES, cfg, reps=1) for _ in range(6)]   # each launch on its own: the spread is reported
    ggs_ms = sum(ggs_each) / len(ggs_each)
    den_ms = eng.time_kernel(0, EB, N_FRAMES, cfg, reps=20)
    M = N_FRAMES * (N_FRAMES - 1) // 2 * PER_PAIR
    ggs_flops = EB * M * FLOP_PER_MATCH_ITER * 7 * cfg.iter_num              # one pd_ggs_guide launch = 700 iterations
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
                engines[j].ggs_guide(full_pose, 0, cfg)
                e1.record(st)
            evs.append((e0, e1))
        torch.cuda.synchronize()
    ggs_set_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs)
    ggs_set_tflops = depth * ggs_flops / (ggs_set_ms * 1e-3) / 1e12
    match_bytes = float(EB) * M * MATCH_BYTES * 7 * cfg.iter_num             # streamed once per iteration at one workgroup per sequence
    # (the lane-per-item kernel keeps 14 steps of every lane item in registers and 3 in LDS: it streams ~70 % of these bytes -- `fabric.streamed_fraction`)
    ceil_rng, ceil_src = (None, "skipped (--no-stream-probe)") if (args.no_stream_probe or rank != 0) else stream_ceiling()
    ggs_traffic, traffic_src = pmc_traffic("ggs_launch", EB) if (wgs or 24) == 1 else (None, "PMC summary is for one workgroup per sequence")
    k_eff = wgs or 24
    import ctypes as _C
    plan8 = (_C.c_int * 8)()
    lane_kernel = False
    if hasattr(eng.lib, "pd_debug_ggs_plan") and eng.lib.pd_debug_ggs_plan(eng._h, EB, N_FRAMES, _C.byref(cfg), plan8) == 0:
        lane_kernel = bool(plan8[6])
    kname = ("pd_ggs_lane_kernel<14> (a lane per work item: 8 waves, 14 steps of every item resident in registers + 3 in LDS, the rest through an LDS ring fed by LDS-DMA)"
             if lane_kernel else f"pd_ggs_kernel<5, false, {plan8[4] or 12}> (a wave per work item)")
    streamed, lane_items, lane_wave_steps = lane_stream_fraction([PER_PAIR] * (N_FRAMES * (N_FRAMES - 1) // 2)) if lane_kernel else (1.0, 0, [])
    streamed_rate = match_bytes * streamed / (ggs_ms * 1e-3) / 1e9          # GB/s the launch pulls through the fabric
    roofline = {
        "kernel": f"{kname}: one launch = one guided diffusion step = 700 iterations x {EB} sequences, {k_eff} workgroup(s) per sequence",
        "bound": "valu", "bound_detail": "fp32 vector ALU, 157.3 TFLOP/s (SURVEY 8d names the arithmetic roofline for the Sampson kernel; the kernel issues no "
                                         "MFMA).  A full-chip launch of it waits for VALU issue (~80 % of the packed-fp32 issue rate the SIMDs sustain) and for "
                                         "its match stream (`fabric`: ~94 % of the HBM peak figure, out of the Infinity Cache) at once",
        "achieved": ggs_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ggs_tflops / FP32_PEAK_TFLOPS,
        "algorithmic_flops_per_launch": ggs_flops, "algorithmic_flop_per_match_iteration": FLOP_PER_MATCH_ITER,
        "launch_ms": ggs_ms, "launch_ms_each": ggs_each, "launch_timing": "hipEvents around the launch on its stream (pd_time_kernel), one launch alone on the chip: "
                                              f"its {EB * k_eff} workgroups take one CU each (256 CUs)",
        "traffic": ggs_traffic, "traffic_source": traffic_src,
        "co_resident": {"launches": depth, "wall_ms": ggs_set_ms, "achieved": ggs_set_tflops, "frac": ggs_set_tflops / FP32_PEAK_TFLOPS,
                        "note": f"the {depth} contexts' launches issued together on their streams, as in the pipe; reproducible from "
                                "profiles/ with tools/coresident_from_trace.py (union of the kernel's intervals in a rocprofv3 kernel trace)"},
        "fabric": {"algorithmic_bytes_per_launch": match_bytes, "streamed_fraction": streamed, "lane_items_per_sequence": lane_items, "lane_wave_steps": lane_wave_steps,
                   "streamed_bytes_per_launch": match_bytes * streamed,
                   "achieved_GBps_one_launch": match_bytes / (ggs_ms * 1e-3) / 1e9,
                   "streamed_GBps_one_launch": match_bytes * streamed / (ggs_ms * 1e-3) / 1e9,
                   "achieved_GBps_co_resident": depth * match_bytes / (ggs_set_ms * 1e-3) / 1e9,
                   "hbm_peak_GBps": HBM_PEAK_GBS, "frac_of_hbm_peak_one_launch": match_bytes / (ggs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "frac_of_hbm_peak_streamed_one_launch": streamed_rate / HBM_PEAK_GBS,
                   "probe_rates_GBps": None if ceil_rng is None else {"lds_dma_ring_like_the_kernel": [ceil_rng[0][0], ceil_rng[0][1]],
                                                                        "plain_loads": None if ceil_rng[1] is None else [ceil_rng[1][0], ceil_rng[1][1]]},
                   "probe_rates_source": ceil_src,
                   "ratio_to_best_probe_rate": None if ceil_rng is None else streamed_rate / max(ceil_rng[0][1], ceil_rng[1][1] if ceil_rng[1] else 0.0),
                   "probe_rates_note": "REFERENCE rates of two synthetic streams on this box, not ceilings: rounds 3-4 called the plain-load probe a 'measured ceiling' "
                                       "and the kernel exceeded it (1.06); round 5 rebuilt the probe with the kernel's own pattern (8 waves per CU, 2 KiB steps through "
                                       "LDS rings fed by global_load_lds_dwordx4) and the kernel still streams faster than it (ratio > 1: its requests are spread "
                                       "over the iteration by 59 VALU instructions per step instead of arriving in lockstep).  The hard bounds are the HBM peak "
                                       "(`hbm_peak_GBps`; the 233 MB working set is Infinity-Cache resident, so even that is not binding by itself) and the "
                                       "fp32 ALU peak of `roofline.peak`; STREAMED bytes = algorithmic x streamed_fraction = the PMC FETCH_SIZE of the launch "
                                       "(profiles/round5_pmc_summary.json)",
                   "note": f"{EB * depth} sequences in flight, at most 256 of them (one GGS workgroup per CU) iterating at a time: {min(EB * depth, 256)} x "
                           f"{M * MATCH_BYTES / 1e6:.2f} MB of matches = {min(EB * depth, 256) * M * MATCH_BYTES / 1e6:.0f} MB re-read every iteration at one workgroup "
                           "per sequence (a chosen trade: no replicated serial phase); that set fits the 256 MiB Infinity Cache, so this is fabric / "
                           "Infinity-Cache bandwidth, not an HBM measurement"},
    }
    den_set_ms = None
    if depth > 1:
        reps = 10
        xs = [torch.randn(EB, N_FRAMES, 9, device=dev) for _ in range(depth)]
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
    tokens = EB * N_FRAMES
    den_flops = tokens * DENOISER_MFLOP_PER_TOKEN * 1e6
    den_tflops = den_flops / (den_ms * 1e-3) / 1e12
    den_gbs = DENOISER_PARAMS * 4 / (den_ms * 1e-3) / 1e9
    den_traffic, den_src = pmc_traffic("denoiser_step", EB)
    if tokens > 50:      # the weight stream bounds the denoiser up to ~50 tokens, the exact-fp32 matrix pipe above
        split_default = tokens >= PD_STREAM_MIN_ROWS       # the fp16-plane kernels are the default there (PD_OPT_DENOISER_SPLIT = 2)
        den_peak = F16_PEAK_TFLOPS / 3.0 if split_default else FP32_PEAK_TFLOPS
        roofline_den = {"kernel": f"one denoiser step at {tokens} token rows (pd_gemm_strip_kernel<.., F16> / pd_ln_rows_kernel / pd_attn_mma_kernel / pd_gemm_dma_kernel for _first and _last.0 / "
                                  "pd_tail_kernel launches at >= 1 024 rows; pd_gemm_kernel / pd_attn_kernel below)",
                        "bound": "mfma", "bound_detail": ("fp16 matrix instruction, three products per fp32 product: 2 500 / 3 = 833 TFLOP/s of algorithmic fp32 FLOPs "
                                                          "(the kernels are bound by operand delivery from LDS / L2 well below that, DESIGN 3.1)") if split_default
                        else "exact-fp32 matrix instruction (157.3 TFLOP/s)",
                        "achieved": den_tflops, "peak": den_peak, "unit": "TFLOP/s", "frac": den_tflops / den_peak,
                        "frac_of_exact_fp32_mfma_peak": den_tflops / FP32_PEAK_TFLOPS,
                        "traffic": den_traffic, "traffic_source": den_src, "step_us": den_ms * 1e3, "algorithmic_flops_per_step": den_flops,
                        "weights_GBps": den_gbs, "all_contexts_step_us": None if den_set_ms is None else den_set_ms * 1e3,
                        "achieved_all_contexts": None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12,
                        "frac_all_contexts": None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12 / den_peak,
                        "frac_all_contexts_of_exact_fp32_mfma_peak": None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
    else:
        roofline_den = {"kernel": "one denoiser step", "bound": "hbm", "achieved": den_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": den_gbs / HBM_PEAK_GBS, "traffic": den_traffic, "step_us": den_ms * 1e3, "algorithmic_bytes_per_step": DENOISER_PARAMS * 4}

    # ---- the other BASELINE configs, each alone on the chip (driver-visible)
    per_config = None
    if not args.no_per_config and rank == 0:
        for e in engines[1:]:
            e.close()
        per_config = {}
        for name, (b_, n_, img_, ggs_) in {"configs[1] B=1 N=20 GGS off": (1, 20, 224, False), "configs[2] B=1 N=20 GGS on": (1, 20, 224, True),
                                           "configs[3] shard: 8 sequences N=20 GGS on": (8, 20, 224, True),
                                           "configs[4] B=1 N=50 M=367500 336x336 GGS on": (1, 50, 336, True)}.items():
            try:
                per_config[name] = measure_config(diff, dev, b_, n_, img_, ggs_)
                if b_ <= 8 and n_ == 20:
                    per_config[name]["denoiser_hbm_roofline_frac"] = DENOISER_PARAMS * 4 / (per_config[name]["denoiser_step_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
            except Exception as e:  # noqa: BLE001  (the headline must still be reported)
                per_config[name] = {"error": repr(e)}

    out = {
        "metric": "sequences/sec (20-frame, GGS on)", "value": value, "unit": "sequences/s", "n_gpus": world,
        "steps": K, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32" if EB * N_FRAMES < PD_STREAM_MIN_ROWS else "f32 (GGS, attention, LayerNorm, DDPM update, every accumulation); the encoder GEMMs multiply fp16 hi + lo operand pairs (22 bits, static power-of-two scales) with fp32 accumulation",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[3]: one step = one batch of {step_total} independent 20-frame sequences"
                        + (f" block-partitioned over {world} GPU(s) ({B_step} per GPU and step; a GPU runs the shards of {group} consecutive "
                           f"steps as one engine pass of {EB} sequences)" if strong else f" ({args.seqs_per_step} per GPU)")
                        + f", {depth} engine passes in flight per GPU; 100 DDPM steps, GGS on for the last {COND_START} steps "
                        f"(7000 iterations/sequence), M={M} matches/sequence (190 pairs x {PER_PAIR}), {IMG}x{IMG}; random-init reference-rule "
                        "weights, matches epipolar-consistent with the engine's own unguided model mean at t=9; inputs resident in HBM",
            "sequences_per_step": step_total, "sequences_per_gpu_per_step": B_step, "steps_per_engine_pass": group, "sequences_per_engine_pass": EB,
            "engine_passes_in_timed_region": len(pend), "sequences_in_flight_per_gpu": EB * depth, "frames": N_FRAMES,
            "matches_per_sequence": M, "diffusion_steps": 100, "ggs_iterations_per_sequence_run": float(iters.min().item()),
            "hip_graph": use_graph, "pipeline_depth": depth, "guided_slots": slots, "unguided_streams": max(0, args.unguided_streams), "ggs_workgroups_per_sequence": k_eff,
            "engine_pass_latency_ms_unpipelined": pass_latency_ms,
            "parallelism": f"dp{world} (independent sequences, one final all_gather)", "outputs_finite": finite,
            "headline_slots_equal_alone": slots_equal, "headline_slots_checked": check_slots,
        },
        "roofline": roofline,
        "roofline_denoiser": roofline_den,
        "per_step_ms": {"denoiser_step": den_ms, "ggs_guided_step": ggs_ms, "ggs_iteration_us": ggs_ms * 1e3 / (7 * cfg.iter_num)},
    }
    if cold is not None:
        out["cold_single_batch"] = cold
    if dry:
        out["dry_dist"] = {"backend": torch.distributed.get_backend(), "world": 1,
                           "executed": "barrier, all_gather of the poses, all_reduce(MAX) of the time -- the N > 1 call sites on RCCL with one rank; "
                                       "RCCL ACROSS GPUs over xGMI is still unexecuted (no multi-GPU box was available to the builder)"}
    if fast is not None:
        out["exact_mode"] = fast
    if fresh is not None:
        out["fresh_inputs"] = fresh
    if per_config is not None:
        out["per_config"] = per_config
    if rank == 0:
        if args.cpu_budget_s > 0 and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_budget_s)
            except Exception as e:  # noqa: BLE001  (the GPU number must still be reported)
                out["cpu_baseline"] = {"value": None, "unit": "sequences/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        else:
            out["cpu_baseline"] = {"value": None, "unit": "sequences/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(),
                                   "kind": "port", "sample": "skipped (measured on rank 0 at N=1 only)"}
        sys.stdout.flush()
        import ctypes as _Cf
        _Cf.CDLL(None).fflush(None)           # RCCL's banner sits in libc's stdout buffer: out with it while descriptor 1 is still stderr
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
    if world > 1 or dry:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
