import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0, num_layers=1).to(dev)
B, N = 8, 20
tables = {k: v for k, v in diff.named_buffers(recurse=False)}
engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N, num_layers=1) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
x0s = []
for e, eng in enumerate(engs):
    xs = []
    for b in range(B):
        enc = synth.make_cameras(N, seed=2000 + 8 * e + b)
        md = synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + 8 * e + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        xs.append(synth.perturb_pose(enc, seed=7 + b))
    x0s.append(torch.cat(xs).to(dev))
torch.cuda.synchronize()
for kk in (0, 16, 8, 2, 1):
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=kk, iter_num=20)
    refs = []
    for j in range(2):
        refs.append(engs[j].ggs_guide(x0s[j], 0, cfg)[0].clone())
        torch.cuda.synchronize()
    bad = [0, 0]
    for rep in range(5):
        outs = []
        for i in range(4):
            j = i % 2
            with torch.cuda.stream(streams[j]):
                outs.append((j, engs[j].ggs_guide(x0s[j], 0, cfg)[0]))
        torch.cuda.synchronize()
        for e in engs:
            e.check_async()
        for j, o in outs:
            if not torch.equal(o, refs[j]):
                bad[j] += 1
    print(f"GGS-only overlap k={kk}: mismatches engine0 {bad[0]}/10 engine1 {bad[1]}/10", flush=True)
