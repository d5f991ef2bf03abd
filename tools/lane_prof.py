"""Phase clocks of pd_ggs_lane_kernel at the bench's launch shapes (pd_debug_ggs_prof: shader cycles of ONE wave of workgroup 0 per iteration: match pass incl. the lane's F,
wait for the other waves at the barrier, P3a + P3b, P4).  python tools/lane_prof.py [sequences = 256]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from posediffusion_amd import _lib, synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 20
dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
mds = []
for s in range(4):
    enc = synth.make_cameras(N, seed=2000 + s)
    mds.append((enc, synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + s)))
x0 = []
for b in range(B):
    enc, md = mds[b % 4]
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0.append(synth.perturb_pose(enc, seed=7 + b))
x0 = torch.cat(x0).to(dev)
cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=_lib.PD_GGS_CFG_LANE_ITEMS)
eng.ggs_guide(x0, 0, cfg)
torch.cuda.synchronize()
t0 = time.time()
eng.ggs_guide(x0, 0, cfg)
torch.cuda.synchronize()
print(f"B = {B}: {(time.time() - t0) * 1e3:.2f} ms per launch unprofiled = {(time.time() - t0) * 1e6 / 700:.2f} us per iteration")
for w in range(8):
    buf = (C.c_longlong * 16)()
    _lib.check(eng.lib.pd_debug_ggs_prof(eng._h, 1 + w, None), "on")
    eng.ggs_guide(x0, 0, cfg)
    torch.cuda.synchronize()
    _lib.check(eng.lib.pd_debug_ggs_prof(eng._h, 1 + w, buf), "read")
    v = list(buf)
    it = max(v[5], 1)
    print(f"  wave {w}: cycles per iteration: pass {v[1] / it:.0f}, wait at the barrier {v[2] / it:.0f}, P3a + P3b {v[3] / it:.0f}, P4 {v[4] / it:.0f}; total {(v[1] + v[2] + v[3] + v[4]) / it:.0f} over {it} iterations")
eng.ggs_prof(False)
