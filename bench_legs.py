"""bench_legs.py -- the measurement legs of bench.py, one callable each ("split bench.py into callable legs").

bench.py owns the command line, the distributed setup, the timed region and the JSON line; everything it reports beside `value`
is produced by one of the functions below from a `Bench` context (engines, pipe, resident inputs, GGS configuration).  Only
`cpu_baseline` touches `oracle/` (the task's rule: the oracle is test infrastructure; bench.py's cpu_baseline leg may time it).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import functools
import hashlib
import io
import json
import os
import subprocess
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from posediffusion_amd import shard, synth  # noqa: E402

N_FRAMES = 20
IMG = 224
PER_PAIR = 300
STEP_SEQS = 64                       # BASELINE configs[3]: one batch of 64 independent sequences
COND_START = 10                      # cfgs/default.yaml:8
FLOP_PER_MATCH_ITER = 100.0          # 36 fwd + 64 bwd
DENOISER_PARAMS = 17_298_697         # fp32 -> 69.19 MB read per denoiser step
DENOISER_MFLOP_PER_TOKEN = 34.73     # SURVEY.md section 8(d), N = 20: every Linear of Denoiser.forward, `_first` at K = 702
FIRST_HOISTED_MFLOP_PER_TOKEN = 2 * 512 * (384 + 128) / 1e6    # the z and t_emb columns of `_first` (models/denoiser.py:56-70): at >= 1 024 rows
                                     # they are computed once per sampling call / read from a table, NOT in a step
PD_STREAM_MIN_ROWS = 1024            # csrc/pd_gemm_stream.h
FP32_PEAK_TFLOPS = 157.3             # MI355X fp32 vector ALU = dense fp32 MFMA peak (MI355X_MICROARCH.md)
F16_PEAK_TFLOPS = 2500.0             # dense fp16 / bf16 MFMA peak (same guide; the sparsity figure is never used)
HBM_PEAK_GBS = 8000.0
MATCH_BYTES = 16                     # kp1, kp2: 2 x float2 per match (pair indices are per work item)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round6_pmc_summary.json")
LANE_RESIDENT_STEPS = 17             # PD_LANE_RV + PD_LANE_RL (csrc/pd_ggs_lane.inc): 14 steps in registers + 3 in LDS for the whole launch


class Bench(SimpleNamespace):
    """What the legs share: args, rank / world / dry, dev, diff, tables, engines (eng = engines[0]), pipe, inputs [(z, noise, mds)] per context,
    cfg (pd_ggs_cfg), wgs, EB, depth, B_step, group, K, use_graph, check_slots."""


# ------------------------------------------------------------------------------------------------------------ inputs
def make_batch_inputs(eng, diff, B, dev, seed0, n_frames=N_FRAMES, img=IMG, per_pair=PER_PAIR, keep_host=False, upload=True, z=None):
    """Synthetic inputs for B sequences (seeds seed0 .. seed0+B-1): z (or the given one), reference-order noise, and matches that are
    epipolar-consistent with the engine's own unguided model mean at the first guided step (so every guided step runs its
    full 700 iterations, as it does with a trained checkpoint and real SuperGlue matches).  Matches go to the engine's
    slots; with keep_host the per-sequence matches_dicts are returned too (fresh-inputs mode packs them into pinned memory)."""
    from posediffusion_amd.host import draw_noise
    T = diff.num_timesteps
    if z is None:
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


def lane_stream_fraction(pair_sizes, lanes=512, resident_steps=LANE_RESIDENT_STEPS):
    """Share of the algorithmic match bytes the lane-per-item GGS kernel pulls through the fabric per iteration (reporting only; the rule
    is pd_ggs_set_matches' in csrc/pd_ggs.hip: the smallest item length that leaves <= `lanes` lane items, k = 1..3 more cuts for the
    spare / k - d pairs with the longest items ((k, d) by the modelled match pass, pd_lane_pass_cost), items ordered by length, 64 per wave,
    a wave's stream padded to its longest item; the first `resident_steps` steps (two matches per lane each) of every wave live on chip
    for the whole launch).  -> (streamed fraction, lane items, steps per wave)."""
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
    steps, waves, _ = wave_steps(nch)                                      # a wave runs (and streams) as many steps as its longest item
    streamed = sum(max(t - resident_steps, 0) for t in waves) * 64 * 32     # bytes per iteration and sequence
    return streamed / (16.0 * sum(ms)), len(steps), waves


def stream_ceiling():
    """Reference rates of this box for the lane kernel's OWN access pattern (tools/stream_probe.hip, `RING` rows; built by
    __graft_entry__.build()): one workgroup of 8 waves per CU, every wave streaming its contiguous share of a private region in 2 KiB
    steps through an LDS ring fed by global_load_lds_dwordx4, with no arithmetic beside it.  Rows the probe could not launch, and rates
    outside (0, 40 TB/s), are dropped.  -> ((min, max) GB/s over the ring rows, (min, max) over the plain 912 KB rows), source; or (None, reason)."""
    exe = os.path.join(ROOT, "tools", "stream_probe")
    if not os.path.isfile(exe):
        return None, "tools/stream_probe not built"
    try:
        txt = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240).stdout
    except Exception as e:  # noqa: BLE001
        return None, f"tools/stream_probe failed: {e!r}"
    ring, plain = [], []
    for line in txt.splitlines():
        f = line.split()
        try:
            rate = float(f[-1]) * 1e3
        except (ValueError, IndexError):
            continue
        if not (0.0 < rate < 40000.0):
            continue
        if len(f) >= 5 and f[0] == "RING":
            ring.append(rate)
        elif len(f) >= 5 and f[0] == "912" and f[1] == "KB":
            plain.append(rate)
    if not ring:
        return None, "tools/stream_probe printed no usable RING rows"
    return ((min(ring), max(ring)), (min(plain), max(plain)) if plain else None), (
        "tools/stream_probe on THIS box, after the timed region: 256 workgroups x 8 waves, each wave streaming its share of a private 704 / 912 KB "
        "region through an LDS ring of 4 / 6 / 8 x 2 KiB fed by global_load_lds_dwordx4 -- the lane kernel's own pattern, without its arithmetic")


# ------------------------------------------------------------------------------------------------------------ CPU baseline
CPU_GGS_THREADS = 16                 # fixed (a probed thread count swung the figure 2 x between calls); min(.., host cores)
CPU_DEN_THREADS = 8


def cpu_baseline(budget_s: float):
    """The reference's own files executed in place (kind "reference") when the reference tree is present (PD_REFERENCE_ROOT or
    /root/reference, or a staged copy), otherwise the oracle port (kind "port") -- FIXED thread counts, and one WHOLE guided step
    (geometry_guided_sampling: its five optimisations = 700 iterations at M = 57 000) timed, not an extrapolation from a few iterations:
    sequence = 100 denoiser steps (timed: 20) + 10 guided steps (timed: 1).  If one guided step would exceed ~ 2 x the budget, iter_num is
    cut and the sample says so."""
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
    host_cores = os.cpu_count() or 1
    max_threads = torch.get_num_threads()
    den_threads, ggs_threads = min(CPU_DEN_THREADS, host_cores), min(CPU_GGS_THREADS, host_cores)
    torch.set_num_threads(den_threads)
    with torch.no_grad():
        den()
        den()
        n_den, t0 = 20, time.time()
        for _ in range(n_den):
            den()
        t_den = (time.time() - t0) / n_den
    enc = synth.make_cameras(N_FRAMES, seed=2000)
    md = synth.make_matches(enc, IMG, IMG, per_pair=PER_PAIR, seed=2000)
    x0 = synth.perturb_pose(enc, seed=7)
    torch.set_num_threads(ggs_threads)

    def guided_step(iter_num):
        cfg = dict(synth.GGS_CFG, iter_num=iter_num)
        with contextlib.redirect_stdout(io.StringIO()):
            if use_ref:
                return ref.geometry_guided_sampling(x0.clone(), 3, md, cfg)
            return O.geometry_guided_sampling(x0.clone(), 3, md, cfg)

    t0 = time.time()
    guided_step(1)                                                                     # 7 iterations: warm-up and the estimate
    t_it = (time.time() - t0) / 7
    iter_num = 100 if 700 * t_it <= 2.0 * max(budget_s, 1.0) else max(1, int(2.0 * budget_s / (7 * t_it)))
    t0 = time.time()
    guided_step(iter_num)
    t_guided = (time.time() - t0) * (100.0 / iter_num)
    t_seq = 100 * t_den + COND_START * t_guided
    torch.set_num_threads(max_threads)
    what = ("the reference files executed in place (models/denoiser.py, util/geometry_guided_sampling.py + restated pytorch3d "
            "helpers, oracle/ref_stubs.py)" if use_ref else "oracle/pd_oracle.py (torch-CPU restatement of the reference path)")
    return {"value": 1.0 / t_seq, "unit": "sequences/s", "cores": ggs_threads, "host_cores": host_cores,
            "kind": "reference" if use_ref else "port", "threads": {"ggs": ggs_threads, "denoiser": den_threads, "fixed": True},
            "denoiser_ms_per_step": t_den * 1e3, "guided_step_s": t_guided, "guided_step_iterations_timed": 7 * iter_num,
            "sample": f"{what}, torch {torch.__version__} CPU, {ggs_threads} threads for GGS and {den_threads} for the denoiser (fixed; the host has "
                      f"{host_cores} cores; `cores` = the GGS figure, 98 % of the time): {n_den} denoiser steps (B=1, N=20) {t_den * 1e3:.1f} ms/step; ONE whole "
                      f"guided step = geometry_guided_sampling at M=57000 with iter_num={iter_num} ({7 * iter_num} iterations"
                      + ("" if iter_num == 100 else ", scaled to 700") + f") {t_guided:.1f} s; sequence = 100 steps + {COND_START} guided steps"}


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
    out = {"B": B, "frames": n_frames, "image": img, "ggs": ggs_on,
           "matches_per_sequence": n_frames * (n_frames - 1) // 2 * PER_PAIR if ggs_on else 0,
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
        synth._dropin()
        from util.geometry_guided_sampling import geometry_guided_sampling
        cond_fn = functools.partial(geometry_guided_sampling, matches_dict=mds[0], GGS_cfg=dict(synth.GGS_CFG)) if ggs_on else None
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


def per_config(bc: Bench):
    """The other BASELINE configs, each alone on the chip (driver-visible).  Closes the extra engine contexts first (memory)."""
    for e in bc.engines[1:]:
        e.close()
    out = {}
    for name, (b_, n_, img_, ggs_) in {"configs[1] B=1 N=20 GGS off": (1, 20, 224, False), "configs[2] B=1 N=20 GGS on": (1, 20, 224, True),
                                       "configs[3] shard: 8 sequences N=20 GGS on": (8, 20, 224, True),
                                       "configs[4] B=1 N=50 M=367500 336x336 GGS on": (1, 50, 336, True)}.items():
        try:
            out[name] = measure_config(bc.diff, bc.dev, b_, n_, img_, ggs_)
            if b_ <= 8 and n_ == 20:
                out[name]["denoiser_hbm_roofline_frac"] = DENOISER_PARAMS * 4 / (out[name]["denoiser_step_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        except Exception as e:  # noqa: BLE001  (the headline must still be reported)
            out[name] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------------------ images -> cameras
def from_images(bc: Bench, reps=3):
    """The one workload the reference publishes a time for (README.md:45: "0.8 s GGS-off / 80 s GGS-on" per 20-frame sequence, INCLUDING the
    DINO features): PoseDiffusionModel.forward(image [1, 20, 3, 224, 224], cond_fn, cond_start_step, training=False)
    (models/pose_diffusion_model.py:109-142) of the drop-in -- MultiScaleImageFeatureExtractor at three scales on csrc/pd_vit.hip
    (models/image_feature_extractor.py:28-87), GaussianDiffusion.sample, pose_encoding_to_camera -- END TO END, wall clock, images resident on the
    GPU, noise drawn by torch inside the call, GGS off and on (matches pre-extracted: hloc is out of scope).  Random-init weights: timing only."""
    models = synth._dropin()
    from posediffusion_amd.compat import AttrDict
    from posediffusion_amd.host import get_engine
    from util.geometry_guided_sampling import geometry_guided_sampling
    dev = bc.dev
    cfg = {"pose_encoding_type": "absT_quaR_logFL",
           "IMAGE_FEATURE_EXTRACTOR": AttrDict({"_target_": "models.MultiScaleImageFeatureExtractor", "freeze": False}),
           "DENOISER": AttrDict({"_target_": "models.Denoiser", "TRANSFORMER": AttrDict(synth.TRANSFORMER_CFG)}),
           "DIFFUSER": AttrDict({"_target_": "models.GaussianDiffusion", "beta_schedule": "custom"})}
    torch.manual_seed(0)
    model = models.PoseDiffusionModel(**cfg).to(dev).eval()
    img = torch.rand(1, N_FRAMES, 3, IMG, IMG, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        z = model.image_feature_extractor(img.reshape(N_FRAMES, 3, IMG, IMG)).reshape(1, N_FRAMES, -1)
    eng = get_engine(model.diffuser.model, model.diffuser, 1, N_FRAMES)
    _, _, mds = make_batch_inputs(eng, model.diffuser, 1, dev, seed0=9100, upload=False, keep_host=True, z=z)
    # (the matches are consistent with the model mean under make_batch_inputs' noise, not under the noise torch draws inside forward(): a guided
    #  step may then leave through the min_matches break -- the iterations actually run are reported)
    cond_fn = functools.partial(geometry_guided_sampling, matches_dict=mds[0], GGS_cfg=dict(synth.GGS_CFG))
    out = {"frames": N_FRAMES, "image": IMG, "scales": [1, 0.5, 1 / 3], "reference_published": "README.md:45: ~0.8 s GGS-off, ~80 s GGS-on (their GPU)"}
    for tag, fn, cs in (("ggs_off", None, 0), ("ggs_on", cond_fn, COND_START)):
        ts = []
        for rep in range(reps + 1):                             # the first call builds engines, uploads the matches, captures the graph
            torch.manual_seed(11)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()), torch.no_grad():
                res = model(image=img, cond_fn=fn, cond_start_step=cs, training=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        cams = res["pred_cameras"]
        out[tag] = {"ms_end_to_end": min(ts[1:]), "first_call_ms": ts[0], "sequences_per_s": 1e3 / min(ts[1:]),
                    "outputs_finite": bool(torch.isfinite(cams.R).all().item() and torch.isfinite(cams.focal_length).all().item())}
        if fn is not None and getattr(model.diffuser, "last_ggs_stats", None) is not None:
            out[tag]["ggs_iterations_run"] = float(model.diffuser.last_ggs_stats[:, :, :, 1].sum().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(5):
            model.image_feature_extractor(img.reshape(N_FRAMES, 3, IMG, IMG))
    torch.cuda.synchronize()
    out["feature_extractor_ms"] = (time.perf_counter() - t0) * 1e3 / 5
    return out


# ------------------------------------------------------------------------------------------------------------ latency legs
def pass_latency(bc: Bench):
    """Un-overlapped latency of one engine pass (outside the timed region) -> (ms, the pass's poses)."""
    z, noise, _ = bc.inputs[0]
    ms, pose = None, None
    for rep in range(2):        # the first call captures the whole-loop graph
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.cuda.stream(bc.pipe.u_stream):
            pose = bc.eng.sample(z, noise, COND_START, bc.cfg, use_graph=bc.use_graph, want_process=False)[0]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) * 1e3
    return ms, pose


def cold_single_batch(bc: Bench):
    """ONE batch of 64 sequences alone on an idle chip (what configs[3] literally names), reported next to the streaming figure."""
    if bc.EB < STEP_SEQS:
        return None
    z, noise, _ = bc.inputs[0]
    zc, nc = z[:STEP_SEQS].contiguous(), noise[:, :STEP_SEQS].contiguous()
    # the launch shape a caller with ONE batch in flight gets (SamplingPipeline.wgs_per_seq with one context: CUs // sequences = 4
    # workgroups per sequence on the wave-per-item kernels); the streaming shape's figure is reported beside it
    cus = torch.cuda.get_device_properties(bc.dev).multi_processor_count
    cfg_cold = type(bc.cfg).from_buffer_copy(bc.cfg)
    cfg_cold.wgs_per_seq = max(1, cus // STEP_SEQS) if not bc.args.ggs_wgs else bc.args.ggs_wgs
    cfg_cold.reserved = 0
    lat_by = {}
    for tag, c in (("alone", cfg_cold), ("streaming_shape", bc.cfg)):
        lat = []
        for rep in range(3):    # the first call captures this shape's graph
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(bc.pipe.u_stream):
                bc.eng.sample(zc, nc, COND_START, c, use_graph=bc.use_graph, want_process=False)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat_by[tag] = min(lat[1:])
    return {"sequences": STEP_SEQS, "latency_ms": lat_by["alone"], "sequences_per_s": STEP_SEQS / (lat_by["alone"] * 1e-3),
            "ggs_workgroups_per_sequence": int(cfg_cold.wgs_per_seq),
            "latency_ms_with_the_streaming_launch_shape": lat_by["streaming_shape"],
            "note": "one batch of 64 sequences, nothing else in flight: the latency of a single configs[3] batch with the launch shape a caller "
                    f"with one batch in flight gets ({int(cfg_cold.wgs_per_seq)} GGS workgroups per sequence fill the chip); with the streaming shape "
                    "(one workgroup per sequence: 64 of the 256 CUs busy) beside it; `value` is the steady-state rate with "
                    + str(bc.EB * bc.depth) + " sequences in flight"}


# ------------------------------------------------------------------------------------------------------------ other modes of the pipe
def capture_all(bc: Bench, shapes=None):
    """Every context captures its hipGraphs for the given pass sizes (default: the full engine batch)."""
    for j in range(bc.depth):
        for b in (shapes or [bc.EB]):
            z, noise = bc.pass_inputs(j, b)
            with torch.cuda.stream(bc.pipe.u_stream):
                out = bc.engines[j].sample(z, noise, COND_START, bc.cfg, use_graph=bc.use_graph, want_process=False, phase=1)
                bc.engines[j].sample(z, noise, COND_START, bc.cfg, use_graph=bc.use_graph, want_process=False, phase=2, out=out)
            torch.cuda.synchronize()


def exact_mode(bc: Bench, full_pose, n_passes):
    """The same pipe with the encoder GEMMs on the exact-fp32 matrix instruction (PD_OPT_DENOISER_SPLIT = 0) instead of the default fp16-plane
    kernels -- the figure that matches the reference's arithmetic with no argument (quote both, always)."""
    eng, depth, EB = bc.eng, bc.depth, bc.EB
    den_default_ms = eng.time_kernel(0, EB, N_FRAMES, bc.cfg, reps=20)
    for e in bc.engines:
        e.set_split_precision(0)
    capture_all(bc)
    for _ in range(depth):
        bc.submit(EB)
    torch.cuda.synchronize()
    n_fast = max(depth, min(n_passes, 4 * depth))
    t3 = time.perf_counter()
    pfm = [bc.submit(EB) for _ in range(n_fast)]
    torch.cuda.synchronize()
    dt3 = time.perf_counter() - t3
    den_fast_ms = eng.time_kernel(0, EB, N_FRAMES, bc.cfg, reps=20)
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
    for e in bc.engines:
        e.set_split_precision(2)
    return fast


def fresh_inputs(bc: Bench, full_pose, n_passes):
    """The same pipe with every pass bringing NEW z / noise / matches from pinned host memory inside the timed region (the PCIe-inclusive rate:
    never `value`).  Restores every context's resident batch (host-built tables) afterwards."""
    from posediffusion_amd.host import pack_matches
    depth, EB, dev, pipe, engines = bc.depth, bc.EB, bc.dev, bc.pipe, bc.engines
    sets = []
    for j in range(min(2, depth)):
        zc, nc, mds = bc.inputs[j]
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
        return pipe.submit(staging[j][0], staging[j][1], COND_START, bc.cfg, use_graph=bc.use_graph, want_process=False)

    for i in range(depth):
        submit_fresh(i)                                                     # warm-up (allocates the slot buffers, captures the graph of the device-built plan)
    torch.cuda.synchronize()
    n_fresh = max(depth, min(n_passes, 3 * depth))
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
             "uploaded_bytes_per_pass": up_bytes,
             "upload": "pinned host -> device copy of z, noise, kp1, kp2 (fp64), i12 (int64) on the pass's stream + pd_ggs_set_matches_csr_async "
                       "(device-side stable sort and table build, no host synchronisation)",
             "ggs_iterations_per_sequence_run": float(it2.min().item()),
             "first_pass_bitwise_equals_resident_pass": same,
             "note": "two distinct pre-packed input sets alternate; packing into pinned memory (the data producer's side) is outside the timed region"}
    for j in range(depth):          # back to each context's own resident batch (host-built tables) for the roofline legs
        for b, md in enumerate(bc.inputs[j][2]):
            engines[j].set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    return fresh


def headline_slots_equal_alone(bc: Bench, full_pose):
    """The headline kernel pinned at the headline launch: a guided step of the full engine batch -- the launch shape of
    the timed region: EB workgroups, the lane-per-item kernel -- against the same sequences run ALONE (one workgroup on an idle chip) on a
    second, single-slot engine: bit for bit, all 700 iterations.  (tests/test_gpu_parity_r5.py does this against the oracle too.)"""
    from posediffusion_amd import _lib
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state
    solo = PoseEngine(denoiser_state(bc.diff.model), bc.tables, device=bc.dev, max_B=1, max_N=N_FRAMES)
    big, big_st = bc.eng.ggs_guide(full_pose, 0, bc.cfg)
    bc.eng.check_async()
    # the single sequence must run on the kernel family the big launch ran on: with one context (--pipeline-depth 1) the configuration leaves the launch shape
    # to the engine (wgs_per_seq = 0), which picks the lane-per-item kernel for EB sequences and 24 workgroups of the wave-per-item kernel for one
    solo_cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=bc.cfg.wgs_per_seq, reserved=bc.cfg.reserved)
    plan8 = (C.c_int * 8)()
    if bc.cfg.wgs_per_seq == 0 and hasattr(bc.eng.lib, "pd_debug_ggs_plan") and bc.eng.lib.pd_debug_ggs_plan(bc.eng._h, bc.EB, N_FRAMES, C.byref(bc.cfg), plan8) == 0:
        solo_cfg.wgs_per_seq = int(plan8[0])
        if plan8[6]:
            solo_cfg.reserved |= _lib.PD_GGS_CFG_LANE_ITEMS
    ok = True
    for b in bc.check_slots:
        md = bc.inputs[0][2][b]
        solo.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        one, one_st = solo.ggs_guide(full_pose[b:b + 1], 0, solo_cfg)
        solo.check_async()
        ok = ok and bool(torch.equal(one[0], big[b])) and bool(torch.equal(one_st[0], big_st[b])) \
            and float(one_st[0, :, 1].sum().item()) == 7.0 * bc.cfg.iter_num
    solo.close()
    assert ok, f"slots {bc.check_slots} of the {bc.EB}-sequence GGS launch differ from the same sequences run alone"
    return ok


# ------------------------------------------------------------------------------------------------------------ rooflines
def in_pipe_launches(stamp_sets):
    """[(int64 tensor [COND_START, 2] of {start, end} ticks, ticks per ms)] per pass of the timed region -> durations in ms of every GGS launch the
    timed region ran (pd_ggs_launch_stamps: recorded by the kernel itself on every launch; {0, 0} slots = another kernel ran: dropped)."""
    ms = []
    for st, khz in stamp_sets:
        a = st.cpu().numpy().astype(np.int64)
        for s0, s1 in a:
            if s0 > 0 and s1 > s0 and khz > 0:
                ms.append((s1 - s0) / khz)
    return ms


def in_pipe_busy(stamp_sets):
    """The same stamps as in_pipe_launches -> (wall ms the chip spent inside AT LEAST ONE of those launches / number of launches, launches whose interval
    overlaps another's by more than a tenth of itself).  Two contexts' launches that reach the dispatcher together share the CUs workgroup by workgroup: each then
    lasts about two launch times by its own stamps while the chip did two launches' work -- the mean duration counts that wall time twice, this figure once."""
    iv = []
    for st, khz in stamp_sets:
        for s0, s1 in st.cpu().numpy().astype(np.int64):
            if s0 > 0 and s1 > s0 and khz > 0:
                iv.append((s0 / khz, s1 / khz))
    if not iv:
        return None, 0
    iv.sort()
    busy, cur0, cur1 = 0.0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur1:
            busy += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    busy += cur1 - cur0
    over = 0
    for i, (a, b) in enumerate(iv):
        o = max((min(b, d) - max(a, c) for j, (c, d) in enumerate(iv) if j != i), default=0.0)
        over += int(o > 0.1 * (b - a))
    return busy / len(iv), over


def roofline_ggs(bc: Bench, full_pose, in_pipe_ms, busy=(None, 0)):
    """Roofline of the dominant kernel.  `frac` = the launches of the TIMED REGION (in-kernel wall-clock stamps of every one of them, several
    contexts in flight, replayed from captured graphs); the launch alone on an idle chip (hipEvents, pd_time_kernel) is a side figure."""
    eng, EB, depth, cfg, pipe = bc.eng, bc.EB, bc.depth, bc.cfg, bc.pipe
    eng.time_kernel(1, EB, N_FRAMES, cfg, reps=2)             # warm: the timed launches below start on a busy chip (clocks up)
    ggs_each = [eng.time_kernel(1, EB, N_FRAMES, cfg, reps=1) for _ in range(6)]   # each launch on its own: the spread is reported
    alone_ms = sum(ggs_each) / len(ggs_each)
    M = N_FRAMES * (N_FRAMES - 1) // 2 * PER_PAIR
    ggs_flops = EB * M * FLOP_PER_MATCH_ITER * 7 * cfg.iter_num              # one pd_ggs_guide launch = 700 iterations
    have_pipe = len(in_pipe_ms) > 0
    pipe_ms = float(np.mean(in_pipe_ms)) if have_pipe else None
    ggs_ms = pipe_ms if have_pipe else alone_ms                              # what `achieved` / `frac` are made of
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
                bc.engines[j].ggs_guide(full_pose, 0, cfg)
                e1.record(st)
            evs.append((e0, e1))
        torch.cuda.synchronize()
    ggs_set_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs)
    ggs_set_tflops = depth * ggs_flops / (ggs_set_ms * 1e-3) / 1e12
    match_bytes = float(EB) * M * MATCH_BYTES * 7 * cfg.iter_num             # streamed once per iteration at one workgroup per sequence
    ceil_rng, ceil_src = (None, "skipped (--no-stream-probe)") if (bc.args.no_stream_probe or bc.rank != 0) else stream_ceiling()
    k_eff = bc.wgs or 24
    ggs_traffic, traffic_src = pmc_traffic("ggs_launch", EB) if k_eff == 1 else (None, "PMC summary is for one workgroup per sequence")
    plan8 = (C.c_int * 8)()
    lane_kernel = False
    if hasattr(eng.lib, "pd_debug_ggs_plan") and eng.lib.pd_debug_ggs_plan(eng._h, EB, N_FRAMES, C.byref(cfg), plan8) == 0:
        lane_kernel = bool(plan8[6])
    kname = ("pd_ggs_lane_kernel<14> (a lane per work item: 8 waves, 14 steps of every item resident in registers + 3 in LDS, the rest through an LDS "
             "ring fed by LDS-DMA)" if lane_kernel else f"pd_ggs_kernel<5, false, {plan8[4] or 12}> (a wave per work item)")
    streamed, lane_items, lane_wave_steps = lane_stream_fraction([PER_PAIR] * (N_FRAMES * (N_FRAMES - 1) // 2)) if lane_kernel else (1.0, 0, [])
    streamed_rate = match_bytes * streamed / (ggs_ms * 1e-3) / 1e9          # GB/s the launch pulls through the fabric
    return {
        "kernel": f"{kname}: one launch = one guided diffusion step = 700 iterations x {EB} sequences, {k_eff} workgroup(s) per sequence",
        "bound": "valu",
        "bound_detail": "fp32 vector ALU, 157.3 TFLOP/s (SURVEY 8d names the arithmetic roofline for the Sampson kernel; the kernel issues no MFMA).  A full-chip "
                        "launch is bound by VALU issue at the clock the chip's power budget leaves it (~ 1.8 GHz with 256 CUs busy against 2.3 GHz at a "
                        "quarter chip), with its match stream (`fabric`, out of the Infinity Cache) beside it",
        "achieved": ggs_tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ggs_tflops / FP32_PEAK_TFLOPS,
        "frac_basis": ("mean duration of the %d GGS launches the TIMED REGION itself ran (%d contexts in flight, replayed from captured graphs), each timed by "
                       "the kernel (wall_clock64 at the start of workgroup 0 / atomicMax at the end of every workgroup: pd_ggs_launch_stamps); the rocprofv3 "
                       "--kernel-trace average of the same command is profiles/round6_kernel_stats.txt" % (len(in_pipe_ms), depth)) if have_pipe
                      else "NO in-pipe stamps (another GGS kernel ran, or --no-graph): the launch alone on an idle chip, hipEvents",
        "algorithmic_flops_per_launch": ggs_flops, "algorithmic_flop_per_match_iteration": FLOP_PER_MATCH_ITER,
        "launch_ms": ggs_ms,
        "in_pipe": None if not have_pipe else {"launches": len(in_pipe_ms), "mean_ms": pipe_ms, "min_ms": float(np.min(in_pipe_ms)),
                                               "max_ms": float(np.max(in_pipe_ms)), "p50_ms": float(np.median(in_pipe_ms)),
                                               "busy_ms_per_launch": busy[0], "overlapping_launches": busy[1],
                                               "busy_note": "wall time inside at least one of these launches / launches (union of the stamped intervals): launches of two "
                                                            "contexts that reach the dispatcher together share the CUs and each lasts ~ 2 launch times by its own stamps; "
                                                            "`frac` keeps the plain mean (what rocprofv3's average duration shows)"},
        "alone": {"launch_ms": alone_ms, "launch_ms_each": ggs_each, "achieved": ggs_flops / (alone_ms * 1e-3) / 1e12,
                  "frac": ggs_flops / (alone_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                  "note": "six single launches after the timed region with nothing else on the chip (hipEvents on the launch's stream, pd_time_kernel): a cooler, "
                          "higher-clocked chip than the pipe's -- a side figure, never `frac`"},
        "traffic": ggs_traffic, "traffic_source": traffic_src,
        "co_resident": {"launches": depth, "wall_ms": ggs_set_ms, "achieved": ggs_set_tflops, "frac": ggs_set_tflops / FP32_PEAK_TFLOPS,
                        "note": f"the {depth} contexts' launches issued together on their streams; reproducible from "
                                "profiles/ with tools/coresident_from_trace.py (union of the kernel's intervals in a rocprofv3 kernel trace)"},
        "fabric": {"algorithmic_bytes_per_launch": match_bytes, "streamed_fraction": streamed, "lane_items_per_sequence": lane_items,
                   "lane_wave_steps": lane_wave_steps, "streamed_bytes_per_launch": match_bytes * streamed,
                   "achieved_GBps_one_launch": match_bytes / (ggs_ms * 1e-3) / 1e9,
                   "streamed_GBps_one_launch": streamed_rate,
                   "achieved_GBps_co_resident": depth * match_bytes / (ggs_set_ms * 1e-3) / 1e9,
                   "hbm_peak_GBps": HBM_PEAK_GBS, "frac_of_hbm_peak_one_launch": match_bytes / (ggs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "frac_of_hbm_peak_streamed_one_launch": streamed_rate / HBM_PEAK_GBS,
                   "probe_rates_GBps": None if ceil_rng is None else {"lds_dma_ring_like_the_kernel": [ceil_rng[0][0], ceil_rng[0][1]],
                                                                        "plain_loads": None if ceil_rng[1] is None else [ceil_rng[1][0], ceil_rng[1][1]]},
                   "probe_rates_source": ceil_src,
                   "ratio_to_best_probe_rate": None if ceil_rng is None else streamed_rate / max(ceil_rng[0][1], ceil_rng[1][1] if ceil_rng[1] else 0.0),
                   "probe_rates_note": "REFERENCE rates of two synthetic streams on this box, not ceilings (the kernel's requests are spread over the iteration by "
                                       "its arithmetic and it streams faster than the lock-step probe).  The hard bounds are the HBM peak figure (the 233 MB working "
                                       "set is Infinity-Cache resident, so even that is not binding by itself) and the fp32 ALU peak of `roofline.peak`",
                   "note": f"{EB * depth} sequences in flight, at most 256 of them (one GGS workgroup per CU) iterating at a time: {min(EB * depth, 256)} x "
                           f"{M * MATCH_BYTES / 1e6:.2f} MB of matches = {min(EB * depth, 256) * M * MATCH_BYTES / 1e6:.0f} MB re-read every iteration at one workgroup "
                           "per sequence; that set fits the 256 MiB Infinity Cache, so this is fabric / Infinity-Cache bandwidth, not an HBM measurement"},
    }, alone_ms


def roofline_denoiser(bc: Bench):
    """One denoiser step at the engine batch: alone (hipEvents, 20 steps) and all contexts together.  The algorithmic FLOPs of a STEP exclude the
    hoisted columns of `_first`: at >= 1 024 rows the z piece runs once per sampling call and the time piece is a table."""
    eng, EB, depth, dev = bc.eng, bc.EB, bc.depth, bc.dev
    den_ms = eng.time_kernel(0, EB, N_FRAMES, bc.cfg, reps=20)
    den_set_ms = None
    if depth > 1:
        reps = 10
        xs = [torch.randn(EB, N_FRAMES, 9, device=dev) for _ in range(depth)]
        evs = []
        for rep in range(2):
            evs = []
            for j in range(depth):
                st = bc.pipe.g_streams[j % len(bc.pipe.g_streams)]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(st):
                    e0.record(st)
                    for _ in range(reps):
                        bc.engines[j].denoise(xs[j], bc.inputs[j][0], 50)
                    e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
        den_set_ms = max(evs[0][0].elapsed_time(e1) for _, e1 in evs) / reps
    tokens = EB * N_FRAMES
    streamed = tokens >= PD_STREAM_MIN_ROWS
    mflop = DENOISER_MFLOP_PER_TOKEN - (FIRST_HOISTED_MFLOP_PER_TOKEN if streamed else 0.0)
    den_flops = tokens * mflop * 1e6
    den_tflops = den_flops / (den_ms * 1e-3) / 1e12
    den_gbs = DENOISER_PARAMS * 4 / (den_ms * 1e-3) / 1e9
    den_traffic, den_src = pmc_traffic("denoiser_step", EB)
    if tokens <= 50:      # the weight stream bounds the denoiser up to ~50 tokens, the matrix pipe above
        return {"kernel": "one denoiser step", "bound": "hbm", "achieved": den_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": den_gbs / HBM_PEAK_GBS, "traffic": den_traffic, "step_us": den_ms * 1e3, "algorithmic_bytes_per_step": DENOISER_PARAMS * 4}, den_ms
    den_peak = F16_PEAK_TFLOPS / 3.0 if streamed else FP32_PEAK_TFLOPS       # the fp16-plane kernels are the default at >= 1 024 rows
    tf_all = None if den_set_ms is None else depth * den_flops / (den_set_ms * 1e-3) / 1e12
    return {"kernel": f"one denoiser step at {tokens} token rows (pd_gemm_strip_kernel<.., F16> / pd_qkv_attn_kernel / pd_ln_rows_kernel / pd_gemm_dma_kernel for "
                      "_first and _last.0 / pd_tail_kernel launches at >= 1 024 rows; pd_gemm_kernel / pd_attn_kernel below)",
            "bound": "mfma",
            "bound_detail": ("fp16 matrix instruction, three products per fp32 product: 2 500 / 3 = 833 TFLOP/s of algorithmic fp32 FLOPs "
                             "(the kernels are bound by operand delivery and launch structure well below that, DESIGN 3.2)") if streamed
            else "exact-fp32 matrix instruction (157.3 TFLOP/s)",
            "achieved": den_tflops, "peak": den_peak, "unit": "TFLOP/s", "frac": den_tflops / den_peak,
            "frac_of_exact_fp32_mfma_peak": den_tflops / FP32_PEAK_TFLOPS,
            "traffic": den_traffic, "traffic_source": den_src, "step_us": den_ms * 1e3, "algorithmic_flops_per_step": den_flops,
            "algorithmic_mflop_per_token_step": mflop,
            "flops_note": (f"{DENOISER_MFLOP_PER_TOKEN} MFLOP per token (SURVEY 8d) minus the {FIRST_HOISTED_MFLOP_PER_TOKEN:.3f} of `_first`'s z and t_emb columns, which the "
                           "streamed path computes once per sampling call / reads from a table (models/denoiser.py:56-70): not in a step's time, so not in its FLOPs"
                           if streamed else "SURVEY 8d: every Linear of Denoiser.forward"),
            "weights_GBps": den_gbs, "all_contexts_step_us": None if den_set_ms is None else den_set_ms * 1e3,
            "achieved_all_contexts": tf_all, "frac_all_contexts": None if tf_all is None else tf_all / den_peak,
            "frac_all_contexts_of_exact_fp32_mfma_peak": None if tf_all is None else tf_all / FP32_PEAK_TFLOPS}, den_ms


# ------------------------------------------------------------------------------------------------------------ one rank of an N-GPU run
def rank_emulation(bc: Bench, worlds=(2, 4, 8), steps=20):
    """What ONE rank of the driver's N-GPU strong-scaling run does (`--steps 20`: 20 x 64 / N sequences per rank, shard.strong_schedule), emulated on
    this GPU with the engines of the timed region -- the 8-GPU shape (configs[3] as BASELINE names it: 8 sequences per GPU and step) as a
    first-class object.  predicted(N) = N x the rank's rate: the ranks share nothing but one final all_gather of 46 KB.
    NO multi-GPU box was available to the builder: these are predictions on record, to be laid beside the driver's SCALE file."""
    out = {"steps": steps, "note": "one rank's passes run on this GPU (same engines, graphs captured for the rank's pass size first); predicted_sequences_per_s = "
                                   "world x rank rate.  No hardware scaling curve was measured by the builder (every lease had one GPU)."}
    for world in worlds:
        g0, g1, group, passes = shard.strong_schedule(steps, STEP_SEQS, world, 0, bc.args.engine_batch, bc.args.min_passes)
        if not passes or max(passes) > bc.EB:
            out[f"rank_of_{world}"] = {"error": f"a pass of {max(passes) if passes else 0} sequences does not fit this run's engine batch of {bc.EB}"}
            continue
        shapes = sorted(set(passes))
        capture_all(bc, shapes)
        for b in passes[:bc.depth]:
            bc.submit(b)                                        # warm
        torch.cuda.synchronize()
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            pend = [bc.submit(b) for b in passes]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        for e in bc.engines:
            e.check_async()
        it = torch.cat([p.stats[:, :, :, 1].sum(dim=(0, 2)).cpu() for p in pend])
        seqs = sum(passes)
        out[f"rank_of_{world}"] = {"sequences_per_gpu_per_step": g1 - g0, "steps_per_engine_pass": group, "passes": passes, "rank_sequences": seqs,
                                   "rank_wall_ms": best * 1e3, "rank_sequences_per_s": seqs / best,
                                   "predicted_sequences_per_s": world * seqs / best, "ggs_iterations_per_sequence_run": float(it.min().item())}
    return out
