"""Do the denoiser's MFMA kernels and the GGS kernel's VALU waves share compute units?  One GGS launch that fills the chip
(256 sequences, one workgroup = one CU each) on stream A, a run of 64-sequence denoiser steps of another engine context on stream B:
each alone, then together.  usage: python tools/overlap_probe.py [reserved flags, default "0,2"]  (2 = no LDS staging)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
BG, BD, N = int(os.environ.get("PD_PROBE_BG", "256")), 64, 20
diff = synth.make_diffuser(seed=0).to(dev)
sd, tabs = denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}
eg = PoseEngine(sd, tabs, device=dev, max_B=BG, max_N=N)
ed = PoseEngine(sd, tabs, device=dev, max_B=BD, max_N=N)
if os.environ.get("PD_PROBE_SPLIT"):
    ed.set_split_precision(True)
cams = [synth.make_cameras(N, seed=2000 + b) for b in range(8)]
mds = [synth.make_matches(c, 224, 224, per_pair=300, seed=2000 + b) for b, c in enumerate(cams)]
for b in range(BG):
    md = mds[b % 8]
    eg.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
x0 = torch.cat([synth.perturb_pose(cams[b % 8], seed=7 + b) for b in range(BG)]).to(dev)
g = torch.Generator(device="cpu").manual_seed(1)
xd = torch.randn(BD, N, 9, generator=g).to(dev)
zd = torch.randn(BD, N, 384, generator=g).to(dev)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(ggs, steps, cfg):
    a0, a1, b0, b1, t0 = ev(), ev(), ev(), ev(), ev()
    torch.cuda.synchronize()
    t0.record()
    sa.wait_event(t0)
    sb.wait_event(t0)
    if ggs:
        with torch.cuda.stream(sa):
            a0.record()
            eg.ggs_guide(x0, 0, cfg)
            a1.record()
    if steps:
        with torch.cuda.stream(sb):
            b0.record()
            for _ in range(steps):
                ed.denoise(xd, zd, 5)
            b1.record()
    torch.cuda.synchronize()
    return (a0.elapsed_time(a1) if ggs else 0.0, b0.elapsed_time(b1) if steps else 0.0,
            max(t0.elapsed_time(a1) if ggs else 0.0, t0.elapsed_time(b1) if steps else 0.0))


for reserved in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,2").split(",")]:
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=reserved)
    run(True, 4, cfg)
    tg = min(run(True, 0, cfg)[0] for _ in range(2))
    M = 32
    td = min(run(False, M, cfg)[1] for _ in range(2))
    M = max(4, int(round(tg / (td / M))))          # as many steps as take one GGS launch's time alone
    td = min(run(False, M, cfg)[1] for _ in range(2))
    both = [run(True, M, cfg) for _ in range(3)]
    bg, bd, bt = min(both, key=lambda r: r[2])
    print(f"reserved={reserved}: GGS({BG} seq) alone {tg:.2f} ms | {M} denoiser steps alone {td:.2f} ms ({td / M * 1e3:.0f} us/step) | "
          f"together: GGS {bg:.2f} ms, steps {bd:.2f} ms, both done after {bt:.2f} ms  (serial {tg + td:.2f}, perfect overlap {max(tg, td):.2f})")
