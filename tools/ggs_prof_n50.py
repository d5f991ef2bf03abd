"""GGS phase clocks at N = 50 (BASELINE configs[4]: 1 225 pairs x 300 matches, 336^2; the two-hop kernel pd_ggs2_kernel, k = 154 workgroups of one
resident item per wave) -- development tool.  pd_debug_ggs_prof: shader cycles per phase of wave 0 of workgroup 0 (owner of frame 0) and of the last
workgroup (owns no frame), summed over the 700 iterations of one guided step."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from posediffusion_amd import _lib, synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
N, H, W = 50, 336, 336
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=1, max_N=N)
enc = synth.make_cameras(N, seed=2000)
md = synth.make_matches(enc, H, W, per_pair=300, seed=2000)
eng.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
x0 = synth.perturb_pose(enc, seed=7).to(dev)
names = ["P1", "P2 match pass", "P3a pair backward", "hop-1 publish + totals", "P3b owner: gather + sum + publish", "hop-2 gathers", "frame gradients + totals", "P4 update"]
for k in (0, 64):
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=k)
    plan = (C.c_int * 8)()
    eng.lib.pd_debug_ggs_plan(eng._h, 1, N, C.byref(cfg), plan)
    eng.ggs_prof(False)
    eng.ggs_guide(x0, 0, cfg)
    torch.cuda.synchronize()
    t0 = time.time()
    eng.ggs_guide(x0, 0, cfg)
    torch.cuda.synchronize()
    dt = time.time() - t0
    buf = (C.c_longlong * 16)()
    _lib.check(eng.lib.pd_debug_ggs_prof(eng._h, 1, None), "prof on")
    eng.ggs_guide(x0, 0, cfg)
    torch.cuda.synchronize()
    _lib.check(eng.lib.pd_debug_ggs_prof(eng._h, 1, buf), "prof read")
    v = list(buf)
    print(f"N={N} wgs_per_seq={k} -> plan {list(plan)}: {dt / 700 * 1e6:.2f} us/it (unprofiled launch)", flush=True)
    for tag, o in (("workgroup 0 (owns frame 0)", 0), ("last workgroup (owns no frame)", 8)):
        tot = sum(v[o:o + 8])
        print(f"  {tag}: cycles per iteration: " + ", ".join(f"{n} {v[o + i] / 700:.0f}" for i, n in enumerate(names)) + f"; total {tot / 700:.0f}", flush=True)
eng.ggs_prof(False)
