"""Minimal process for counter collection (tools/collect_pmc.sh): B sequences, two GGS launches at k workgroups per sequence
and a few denoiser steps, no graphs, one stream.  usage: python tools/pmc_target.py [B=64] [k=1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posediffusion_amd import _lib, synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, H, W = 20, 224, 224
dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
eng = PoseEngine(denoiser_state(diff.model), {n: v for n, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
cams = [synth.make_cameras(N, seed=2000 + b) for b in range(min(B, 16))]     # 16 distinct scenes, repeated over the slots
mds = [synth.make_matches(c, H, W, per_pair=300, seed=2000 + b) for b, c in enumerate(cams)]
for b in range(B):
    md = mds[b % len(mds)]
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
x0 = torch.cat([synth.perturb_pose(cams[b % len(cams)], seed=7 + b) for b in range(B)]).to(dev)
cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=k, reserved=_lib.PD_GGS_CFG_LANE_ITEMS if k == 1 else 0)   # as the pipeline asks (SamplingPipeline.make_cfg)
for _ in range(2):
    eng.ggs_guide(x0, 0, cfg)
torch.cuda.synchronize()
eng.time_kernel(0, B, N, cfg, reps=3)
torch.cuda.synchronize()
eng.check_async()
print("pmc_target done", B, k)
