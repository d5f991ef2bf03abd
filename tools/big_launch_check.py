"""One GGS launch of B > 256 sequences on the lane-per-item kernel (one workgroup per sequence: the dispatcher back-fills the CUs as workgroups finish)
against the same sequences run alone, bit for bit.  The B sequences cycle through 4 distinct ones, so every slot b must equal slot b % 4 and those four
the single-workgroup launch of an engine of its own.    python tools/big_launch_check.py [B=768] [repeats=3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posediffusion_amd import _lib, synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = 20
diff = synth.make_diffuser(seed=0).to(dev)
state, tables = denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}
mds, x0s = [], []
for s in range(4):
    enc = synth.make_cameras(N, seed=2000 + s)
    mds.append(synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + s))
    x0s.append(synth.perturb_pose(enc, seed=7 + s))
cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=_lib.PD_GGS_CFG_LANE_ITEMS)
solo = PoseEngine(state, tables, device=dev, max_B=1, max_N=N)
alone = []
for s in range(4):
    solo.set_matches(0, mds[s]["kp1"], mds[s]["kp2"], mds[s]["i12"], mds[s]["img_shape"])
    o, st = solo.ggs_guide(x0s[s].to(dev), 0, cfg)
    solo.check_async()
    alone.append((o[0].clone(), st[0].clone()))
solo.close()
eng = PoseEngine(state, tables, device=dev, max_B=B, max_N=N)
for b in range(B):
    md = mds[b % 4]
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
x0 = torch.cat([x0s[b % 4] for b in range(B)]).to(dev)
for rep in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    o, st = eng.ggs_guide(x0, 0, cfg)
    e1.record()
    torch.cuda.synchronize()
    eng.check_async()
    bad = [b for b in range(B) if not (torch.equal(o[b], alone[b % 4][0]) and torch.equal(st[b], alone[b % 4][1]))]
    its = st[:, :, 1].sum(dim=1)
    print(f"B={B} launch {rep}: {e0.elapsed_time(e1):.2f} ms ({e0.elapsed_time(e1) * 256 / B:.2f} per 256 sequences); slots that differ from the sequence alone: {len(bad)}"
          + (f" -> {bad[:24]}{' ...' if len(bad) > 24 else ''}; iterations stepped there: {sorted(set(int(v) for v in its[bad].tolist()))}; worst |diff| "
             f"{max(float((o[b] - alone[b % 4][0]).abs().max()) for b in bad):.3e}" if bad else ""), flush=True)
eng.close()
