"""What would reduced-precision GEMMs cost the denoiser?  (missing item 4.)

CPU-side simulation on the oracle's arithmetic (TEST INFRASTRUCTURE: imports oracle/): every Linear of the Denoiser is
evaluated in one of three modes, everything else (LayerNorm, softmax, residuals, embeddings, the DDPM update) stays fp32:
    fp32    plain fp32 (what csrc/pd_denoiser.hip computes on v_mfma_f32_32x32x2_f32)
    split   both operands as bf16 hi + bf16 lo, products hi*hi + hi*lo + lo*hi, fp32 accumulation
            (the scheme of vit_gemm_split_kernel in csrc/pd_vit.hip: three bf16 MFMA products)
    bf16    both operands rounded to bf16 once, fp32 accumulation (one bf16 MFMA product)
    split3  both operands as bf16 hi + mid + lo (3 x 8 = 24 mantissa bits: an fp32 value is represented exactly up to its last bit),
            the six products of weight >= 2^-16: hi*hi + (hi*mid + mid*hi) + (hi*lo + mid*mid + lo*hi), fp32 accumulation --
            the "bf16 pipe, not narrower than the reference" candidate of VERDICT round 2 item 6
and compared with the fp64 oracle on the reference-generated trajectory fixture (tests/golden/trajectory.npz, B = 1, N = 20):
teacher-forced (each step fed the fp64 state) and free-running over the 100 steps.

    python tests/perf/denoiser_precision_study.py [out.json]
"""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pd_oracle as O  # noqa: E402
from posediffusion_amd import synth  # noqa: E402


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def linear(x, w, b, mode):
    if mode in ("fp32", "fp64"):
        return x @ w.T + b
    if mode == "bf16":
        return _bf16(x) @ _bf16(w).T + b
    xh, wh = _bf16(x), _bf16(w)
    xl, wl = _bf16(x - xh), _bf16(w - wh)
    if mode == "split3":
        xm, wm = xl, wl                                    # mid = bf16 of the first residual, lo = bf16 of the second
        xl, wl = _bf16(x - xh - xm), _bf16(w - wh - wm)
        return (xh @ wh.T + ((xh @ wm.T + xm @ wh.T) + ((xh @ wl.T + xl @ wh.T) + xm @ wm.T))) + b
    return (xh @ wh.T + (xh @ wl.T + xl @ wh.T)) + b


def denoiser(sd, x, t, z, mode, num_layers=8, nhead=4):
    B, N, _ = x.shape
    t_emb = O.timestep_embedding(t, sd)[:, None, :].expand(-1, N, -1)
    pivot = torch.zeros_like(z[..., :1])
    pivot[:, 0] = 1.0
    h = linear(torch.cat([O.harmonic_embedding(x), t_emb, z, pivot], dim=-1), sd["_first.weight"], sd["_first.bias"], mode)
    for l in range(num_layers):
        p = f"_trunk.layers.{l}."
        d, dh = h.shape[-1], h.shape[-1] // nhead
        a = O._layer_norm(h, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
        q, k, v = linear(a, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"], mode).split(d, dim=-1)
        q, k, v = [u.reshape(B, N, nhead, dh).transpose(1, 2) for u in (q, k, v)]
        ctx = (torch.softmax((q / math.sqrt(dh)) @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B, N, d)
        h = h + linear(ctx, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], mode)
        a = O._layer_norm(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
        h = h + linear(torch.relu(linear(a, sd[p + "linear1.weight"], sd[p + "linear1.bias"], mode)), sd[p + "linear2.weight"],
                       sd[p + "linear2.bias"], mode)
    h = linear(h, sd["_last.0.weight"], sd["_last.0.bias"], mode)
    h = torch.relu(O._layer_norm(h, sd["_last.1.weight"], sd["_last.1.bias"]))
    return linear(h, sd["_last.3.weight"], sd["_last.3.bias"], mode)


def step(sd, tables, x, t, z, noise, mode):
    eps = denoiser(sd, x, torch.full((x.shape[0],), t, dtype=torch.long), z, mode)
    x0 = tables["sqrt_recip_alphas_cumprod"][t] * x - tables["sqrt_recipm1_alphas_cumprod"][t] * eps
    mean = tables["posterior_mean_coef1"][t] * x0 + tables["posterior_mean_coef2"][t] * x
    return mean + torch.exp(0.5 * tables["posterior_log_variance_clipped"][t]) * (noise if t > 0 else 0.0), eps


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max())


def study(steps=100):
    torch.set_num_threads(min(8, torch.get_num_threads()))
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "trajectory.npz")))
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    sd = {k: v.detach() for k, v in diff.model.state_dict().items()}
    sd32, sd64 = O.cast_state_dict(sd, torch.float32), O.cast_state_dict(sd, torch.float64)
    t32, t64 = O.diffusion_tables(), O.diffusion_tables(dtype=torch.float64)
    z, noise, p64 = torch.from_numpy(g["z"]), torch.from_numpy(g["noise"]), torch.from_numpy(g["process64"])
    out = {"fixture": "tests/golden/trajectory.npz (B=1, N=20, 100 steps; process64 = fp64 oracle)", "steps": steps, "modes": {}}
    with torch.no_grad():
        for mode in ("fp32", "split3", "split", "bf16"):
            tf_eps, tf_x = [], []
            for s_ in range(0, steps, 7):                                   # teacher-forced on the fp64 trajectory
                t = 99 - s_
                x64 = p64[s_]
                ref_x, ref_eps = step(sd64, t64, x64, t, z.double(), noise[s_ + 1].double(), "fp64")
                xm, em = step(sd32, t32, x64.float(), t, z, noise[s_ + 1], mode)
                tf_eps.append(rel(em, ref_eps))
                tf_x.append(rel(xm, ref_x))
            x = noise[0].clone()
            prefix = None
            for s_ in range(steps):                                          # free-running
                x, _ = step(sd32, t32, x, 99 - s_, z, noise[s_ + 1], mode)
                if s_ + 1 == 30:
                    prefix = rel(x, p64[30])
            out["modes"][mode] = {"teacher_forced_eps_rel_max": max(tf_eps), "teacher_forced_x_next_rel_max": max(tf_x),
                                  "free_running_rel_after_30_steps": prefix, "free_running_rel_final": rel(x, p64[steps])}
    out["reference_fp32_free_running_rel_final"] = rel(torch.from_numpy(g["process"])[steps], p64[steps])
    return out


if __name__ == "__main__":
    res = study()
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(res, f, indent=1)
