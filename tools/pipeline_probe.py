"""Experiment: two engine contexts on two streams, consecutive passes overlapped."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state, draw_noise

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0).to(dev)
B, N = 8, 20
tables = {k: v for k, v in diff.named_buffers(recurse=False)}
engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
cfg = make_ggs_cfg(synth.GGS_CFG)
data = []
for e, eng in enumerate(engs):
    z = torch.cat([synth.make_z(1, N, seed=1000 + 8 * e + b) for b in range(B)]).to(dev)
    noise = torch.empty(101, B, N, 9, device=dev)
    for b in range(B):
        noise[:, b] = draw_noise((N, 9), 100, dev, 10, True, generator=torch.Generator(device=dev).manual_seed(8 * e + b))
    _, process, _ = eng.sample(z, noise, 0, None, use_graph=False)
    mean, _ = eng.p_mean(process[90], z, 9)
    mean = mean.cpu().numpy().astype(np.float64)
    for b in range(B):
        md = synth.make_epipolar_matches(mean[b], 224, 224, 300, seed=2000 + 8 * e + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    data.append((z, noise))
torch.cuda.synchronize()


def run(K, depth):
    outs = []
    t0 = time.time()
    for i in range(K):
        j = i % depth
        with torch.cuda.stream(streams[j]):
            outs.append(engs[j].sample(data[j][0], data[j][1], 10, cfg, use_graph=True, want_process=False))
    torch.cuda.synchronize()
    dt = time.time() - t0
    for e in engs:
        e.check_async()
    its = min(float(o[2][:, :, :, 1].sum(dim=(0, 2)).min()) for o in outs)
    return dt, its, outs


for kk in (16, 12):
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=kk)
    for depth in (2,):
        run(2, depth)
        dt, its, outs = run(6, depth)
        print(f"k={kk} depth {depth}: {dt / 6 * 1e3:.2f} ms per pass -> {8 * 6 / dt:.1f} seq/s (iterations/seq {its:.0f})", flush=True)
cfg = make_ggs_cfg(synth.GGS_CFG)
# identical results regardless of overlap?
_, _, o1 = run(2, 1)
_, _, o2 = run(2, 2)
print("pose bitwise equal pipelined vs serial (first pass, same engine/data):", torch.equal(o1[0][0], o2[0][0]))
