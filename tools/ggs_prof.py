"""In-kernel phase cycle counters of the GGS kernels (pd_debug_ggs_prof): cycles per iteration of one wave of workgroup 0.
    python tools/ggs_prof.py B [wgs_per_seq=0 (engine's choice)] [reserved flags=0] [waves to read=0,7]
B = 1: the latency shape (BASELINE configs[2]: 24 workgroups for the one sequence); B = 64, wgs 1: the throughput shape (while the
counters are on, pd_ggs_plan keeps the 8-wave kernel unless the library was built with -DPD_GGS_PROF12)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reserved = int(sys.argv[3]) if len(sys.argv) > 3 else 0
waves = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,7").split(",")]
N = 20
diff = synth.make_diffuser(seed=0).to(dev)
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
x0 = []
for b in range(B):
    enc = synth.make_cameras(N, seed=2000 + (b % 4))
    md = synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + (b % 4))
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0.append(synth.perturb_pose(enc, seed=7 + b))
x0 = torch.cat(x0).to(dev)
cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs, reserved=reserved)
eng.ggs_guide(x0, 0, cfg)
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _, st = eng.ggs_guide(x0, 0, cfg)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"B={B} wgs_per_seq={wgs} reserved={reserved}: launch {ms:.3f} ms = {ms * 1e3 / 700:.2f} us / iteration (iterations run: {int(st[:, :, 1].sum(dim=1).min())})")
for w in waves:
    eng.ggs_prof(1 + w)
    eng.ggs_guide(x0, 0, cfg)
    p = eng.ggs_prof(1 + w)
    tot = sum(p[k] for k in ("P1", "P2", "xchg", "P3", "P4"))
    print(f"  wave {w}: " + " ".join(f"{k}={p[k]:.0f}" for k in ("P1", "P2", "xchg", "P3", "P4", "P3a", "P3_wait1", "P3b")) + f" total={tot:.0f} cycles / iteration ({p['iters']} iterations)")
eng.ggs_prof(0)
eng.check_async()
