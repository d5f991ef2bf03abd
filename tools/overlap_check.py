"""Concurrency check: each engine's output while another engine runs on a second stream must be
bitwise equal to its own serial output."""
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
tables = {k: v for k, v in diff.named_buffers(recurse=False)}
engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
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
import itertools
for (start, kk), ug in itertools.product(((10, 0), (10, 16)), (False, True)):
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=kk) if start else None
    refs = []
    for j in range(2):
        refs.append(engs[j].sample(data[j][0], data[j][1], start, cfg, use_graph=ug)[0].clone())
        torch.cuda.synchronize()
    bad = 0
    for rep in range(3):
        outs = []
        for i in range(4):
            j = i % 2
            with torch.cuda.stream(streams[j]):
                outs.append((j, engs[j].sample(data[j][0], data[j][1], start, cfg, use_graph=ug)[0]))
        torch.cuda.synchronize()
        for e in engs:
            e.check_async()
        for j, o in outs:
            if not torch.equal(o, refs[j]):
                bad += 1
                print(f"   mismatch engine {j}: maxdiff {(o - refs[j]).abs().max().item():.3e}")
    print(f"cond_start={start} k={kk} graph={ug}: {12 - bad}/12 overlapped outputs bitwise equal to the serial reference", flush=True)
