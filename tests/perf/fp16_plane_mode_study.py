"""GPU study (not a test): the denoiser's three GEMM modes (exact fp32 MFMA / bf16 hi+lo / fp16 hi+lo with static scales) --
per-step error and 100-step free-running deviation against the fp64 oracle over 52 sequences, and time per denoiser step at the
bench's 5 120 token rows.  Writes gpurun_out/fp16_plane_mode_study.json (copied to profiles/ by hand).
    python tests/perf/fp16_plane_mode_study.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pd_oracle as O  # noqa: E402
from posediffusion_amd import synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg  # noqa: E402
from posediffusion_amd.host import denoiser_state, draw_noise  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max()).item()


def main():
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    diff = diff.to(dev)
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    sd64 = O.cast_state_dict({k: v.detach().cpu() for k, v in diff.model.state_dict().items()}, torch.float64)
    B, N = 52, 20
    eng = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=256, max_N=N)
    g = torch.Generator().manual_seed(77)
    x, z = torch.randn(B, N, 9, generator=g), synth.make_z(B, N, seed=3)
    out = {"modes": {"0": "exact fp32 MFMA", "1": "bf16 hi + lo, three products", "2": "fp16 hi + lo, three products, static power-of-two scales"}}
    step = {}
    for t in (99, 70, 40, 10, 0):
        with torch.no_grad():
            ref = O.denoiser_forward(sd64, x.double(), torch.full((B,), t, dtype=torch.long), z.double())
        for mode in (0, 1, 2):
            eng.set_split_precision(mode)
            step[f"t{t}_mode{mode}"] = rel(eng.denoise(x.to(dev), z.to(dev), t), ref)
    out["one_step_rel_err_vs_fp64"] = step
    print(json.dumps(step, indent=1))
    noise = draw_noise((B, N, 9), 100, dev, generator=torch.Generator(device=dev).manual_seed(5))
    nz = noise.cpu().double()
    t64 = O.diffusion_tables(dtype=torch.float64)
    with torch.no_grad():
        p64, _ = O.p_sample_loop(sd64, t64, z.double(), nz[0], [None if t == 0 else nz[100 - t] for t in range(100)])
    free = {}
    for mode in (0, 1, 2):
        eng.set_split_precision(mode)
        pose, _, _ = eng.sample(z.to(dev), noise, 0, None, use_graph=True, want_process=False)
        d = np.array([rel(pose[b], p64[b]) for b in range(B)])
        free[str(mode)] = {"median": float(np.median(d)), "mean": float(d.mean()), "p90": float(np.percentile(d, 90)), "max": float(d.max())}
    out["free_running_100_steps_rel_dev_vs_fp64_over_52_sequences"] = free
    print(json.dumps(free, indent=1))
    EB = 256
    cfg = make_ggs_cfg(synth.GGS_CFG)
    tm = {}
    for mode in (0, 1, 2, 0, 2):
        eng.set_split_precision(mode)
        tm.setdefault(str(mode), []).append(eng.time_kernel(0, EB, N, cfg, reps=20) * 1e3)
    out["denoiser_step_us_5120_rows_alone"] = tm
    print(json.dumps(tm))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp16_plane_mode_study.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
