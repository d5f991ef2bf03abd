"""First-contact GPU script: prints engine-vs-oracle deviations and rough timings.
Run on the GPU box:  timeout 600 python tools/gpu_shakeout.py [stage ...]
(development tool; the assertions live in tests/ -m gpu)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import pd_oracle as O  # noqa: E402
from posediffusion_amd import synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

stages = set(sys.argv[1:]) or {"den", "ggs", "sample", "time"}
dev = torch.device("cuda:0")
print("device:", torch.cuda.get_device_name(0), flush=True)

diff = synth.make_diffuser(seed=0)
synth.randomize_norm_and_bias_(diff.model)
sd_cpu = O.cast_state_dict(diff.model.state_dict(), torch.float32)
tables = O.diffusion_tables()
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=8, max_N=50)
print("engine:", eng.version, flush=True)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


if "den" in stages:
    for (B, N) in [(1, 20), (2, 20), (8, 20), (1, 50), (3, 7)]:
        g = torch.Generator().manual_seed(B * 100 + N)
        x = torch.randn(B, N, 9, generator=g)
        z = synth.make_z(B, N)
        for t in (99, 37, 0):
            tt = torch.full((B,), t, dtype=torch.long)
            ref = O.denoiser_forward(sd_cpu, x, tt, z)
            out = eng.denoise(x.to(dev), z.to(dev), t)
            mean_o, _, x0_o, _ = O.p_mean_variance(sd_cpu, tables, x, t, z)
            mean, x0 = eng.p_mean(x.to(dev), z.to(dev), t)
            print(f"den B={B} N={N} t={t}: eps rel {rel(out, ref):.2e}  mean rel {rel(mean, mean_o):.2e} x0 rel {rel(x0, x0_o):.2e}", flush=True)

if "ggs" in stages:
    for (N, per_pair) in [(8, 60), (20, 300)]:
        H = W = 224
        enc = synth.make_cameras(N, seed=2000)
        md = synth.make_matches(enc, H, W, per_pair=per_pair, seed=2000)
        pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        x0 = synth.perturb_pose(enc, seed=7)
        eng.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        for flags in [(True, True, True), (False, False, True), (True, False, False), (False, True, False)]:
            xo = x0.clone().requires_grad_(True)
            vo, pro = O.compute_sampson_distance(xo, pm, *flags)
            lo = vo.mean()
            (go,) = torch.autograd.grad(lo, xo)
            for k in (1, 0):
                loss, grad = eng.ggs_loss_grad(x0.to(dev), *flags, cfg=make_ggs_cfg(wgs_per_seq=k))
                torch.cuda.synchronize()
                eng.check_async()
                print(f"ggs N={N} M={len(md['kp1'])} flags={flags} k={'auto' if k == 0 else k}: loss {loss[0,0].item():.6f} vs {lo.item():.6f} "
                      f"nvalid {loss[0,1].item():.0f} vs {len(vo)} print {loss[0,2].item():.5f} vs {pro.item():.5f} grad rel {rel(grad, go):.2e}", flush=True)
        # iterations with trace
        for flags, it in [((True, True, True), 10), ((False, False, True), 10)]:
            trace = []
            xo, pr, steps = O.ggs_optimize(x0.clone(), pm, *flags, iter_num=it, trace=trace)
            outs = {}
            for k in (1, 0):
                xe, st, tr = eng.ggs_optimize(x0.to(dev), *flags, cfg=make_ggs_cfg(iter_num=it, wgs_per_seq=k), trace=True)
                torch.cuda.synchronize()
                eng.check_async()
                outs[k] = xe.cpu()
                tr = tr.cpu()
                d_first = (tr[0, 0, : N * 9] - trace[0]["x"].flatten()).abs().max().item()
                d_last = (tr[0, steps - 1, : N * 9] - trace[-1]["x"].flatten()).abs().max().item()
                print(f"ggs_opt N={N} flags={flags} k={'auto' if k == 0 else k}: steps {st[0,1].item():.0f} vs {steps}; x after it1 maxabs {d_first:.2e}; "
                      f"after last {d_last:.2e}; final rel {rel(xe, xo):.2e} moved {(xo - x0).abs().max().item():.3e}", flush=True)
            print("   k=1 vs k=auto bitwise equal:", torch.equal(outs[1], outs[0]), flush=True)
        # full guide, reduced iterations
        cfg = dict(synth.GGS_CFG, iter_num=10)
        xo = O.geometry_guided_sampling(x0.clone(), 3, md, cfg)
        xe, st = eng.ggs_guide(x0.to(dev), 3, cfg)
        torch.cuda.synchronize()
        eng.check_async()
        print(f"guide(iter 10) N={N}: rel {rel(xe, xo):.2e} moved {(xo - x0).abs().max().item():.3e} stats {st[0, :, 0].tolist()}", flush=True)

if "sample" in stages:
    B, N = 2, 20
    z = synth.make_z(B, N)
    g = torch.Generator().manual_seed(0)
    init, noises = O.draw_reference_noise((B, N, 9), g)
    noise = torch.zeros(101, B, N, 9)
    noise[0] = init
    for step in range(100):
        t = 99 - step
        if noises[t] is not None:
            noise[step + 1] = noises[t]
    for use_graph in (False, True):
        pose, process, _ = eng.sample(z.to(dev), noise.to(dev), 0, None, use_graph=use_graph)
        torch.cuda.synchronize()
        process = process.cpu()
        # teacher-forced per-step check against the oracle
        worst = 0.0
        for step in (0, 1, 50, 98, 99):
            t = 99 - step
            ref_next, _ = O.p_sample(sd_cpu, tables, process[step], t, z, noises[t])
            worst = max(worst, rel(process[step + 1], ref_next))
        po, pproc = O.p_sample_loop(sd_cpu, tables, z, init, noises)
        print(f"sample graph={use_graph}: teacher-forced worst rel {worst:.2e}; free-run rel {rel(pose, po):.2e} max|pose| {po.abs().max().item():.2f}", flush=True)

if "time" in stages:
    for (B, N) in [(1, 20), (8, 20), (1, 50)]:
        z = synth.make_z(B, N).to(dev)
        noise = torch.randn(101, B, N, 9, device=dev)
        for use_graph in (False, True):
            eng.sample(z, noise, 0, None, use_graph=use_graph)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                eng.sample(z, noise, 0, None, use_graph=use_graph)
            torch.cuda.synchronize()
            print(f"time sample GGS-off B={B} N={N} graph={use_graph}: {(time.time() - t0) / 3 * 1e3:.2f} ms per 100 steps", flush=True)
        print(f"   denoiser step (hipEvent): {eng.time_kernel(0, B, N, reps=20) * 1e3:.1f} us", flush=True)
    N, H, W = 20, 224, 224
    for B in (1, 8):
        for b in range(B):
            enc = synth.make_cameras(N, seed=2000 + b)
            md = synth.make_matches(enc, H, W, per_pair=300, seed=2000 + b)
            eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        x0 = torch.cat([synth.perturb_pose(synth.make_cameras(N, seed=2000 + b), seed=7 + b) for b in range(B)]).to(dev)
        for k in (1, 0):
            cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=k)
            eng.ggs_guide(x0, 0, cfg)
            torch.cuda.synchronize()
            t0 = time.time()
            xe, st = eng.ggs_guide(x0, 0, cfg)
            torch.cuda.synchronize()
            dt = time.time() - t0
            eng.check_async()
            its = st[:, :, 1].sum(1).max().item()
            print(f"time ggs_guide B={B} N={N} k={'auto' if k == 0 else k}: {dt * 1e3:.2f} ms for {its:.0f} iterations -> {dt / max(its, 1) * 1e6:.2f} us/iter; sampson {st[0, :, 0].tolist()}", flush=True)
print("done", flush=True)
