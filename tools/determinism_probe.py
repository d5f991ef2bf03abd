"""Is a full guided pd_sample bitwise reproducible run-to-run (same engine, same inputs, serial)?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state, draw_noise

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0).to(dev)
B, N = 8, 20
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
z = torch.cat([synth.make_z(1, N, seed=1000 + b) for b in range(B)]).to(dev)
noise = torch.empty(101, B, N, 9, device=dev)
for b in range(B):
    noise[:, b] = draw_noise((N, 9), 100, dev, 10, True, generator=torch.Generator(device=dev).manual_seed(b))
_, process, _ = eng.sample(z, noise, 0, None, use_graph=False)
mean, _ = eng.p_mean(process[90], z, 9)
mean_np = mean.cpu().numpy().astype(np.float64)
for b in range(B):
    md = synth.make_epipolar_matches(mean_np[b], 224, 224, 300, seed=2000 + b)
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
for kk in (0, 16, 1):
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=kk)
    # (a) one guided step from a fixed start, repeated
    outs = [eng.ggs_guide(mean, 9, cfg)[0].cpu() for _ in range(4)]
    eng.check_async()
    print(f"k={kk}: ggs_guide repeat bitwise equal: {[torch.equal(outs[0], o) for o in outs[1:]]}  maxdiff {max((outs[0]-o).abs().max().item() for o in outs[1:]):.3e}", flush=True)
    # (b) whole sampler
    ps = [eng.sample(z, noise, 10, cfg, use_graph=True)[0].cpu() for _ in range(3)]
    eng.check_async()
    print(f"k={kk}: pd_sample repeat bitwise equal: {[torch.equal(ps[0], o) for o in ps[1:]]}  maxdiff {max((ps[0]-o).abs().max().item() for o in ps[1:]):.3e}", flush=True)
