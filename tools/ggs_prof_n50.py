"""GGS phase-cycle profile at N = 50 (BASELINE config 5) -- development tool."""
import sys, time
import torch
sys.path.insert(0, ".")
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
N, H, W = 50, 336, 336
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=1, max_N=N)
enc = synth.make_cameras(N, seed=2000)
md = synth.make_matches(enc, H, W, per_pair=300, seed=2000)
eng.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
x0 = synth.perturb_pose(enc, seed=7).to(dev)
for k in (0, 64):
    for pw in (1, 2):
        cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=k)
        eng.ggs_prof(pw)
        eng.ggs_guide(x0, 0, cfg)
        torch.cuda.synchronize()
        t0 = time.time()
        eng.ggs_guide(x0, 0, cfg)
        torch.cuda.synchronize()
        dt = time.time() - t0
        pr = eng.ggs_prof(pw)
        tot = sum(pr[k2] for k2 in ("P1", "P2", "xchg", "P3", "P4"))
        print(f"N={N} k={k} wave={pw-1}: {dt/700*1e6:.2f} us/it; cycles/it P1 {pr['P1']:.0f} P2 {pr['P2']:.0f} xchg {pr['xchg']:.0f} "
              f"P3 {pr['P3']:.0f} [P3a {pr['P3a']:.0f} wait {pr['P3_wait1']:.0f} P3b {pr['P3b']:.0f}] P4 {pr['P4']:.0f} total {tot:.0f}", flush=True)
eng.ggs_prof(False)
