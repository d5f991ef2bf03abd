"""GGS phase-cycle profile + denoiser timing (development tool)."""
import sys, time
import torch
sys.path.insert(0, ".")
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=8, max_N=50)
N, H, W = 20, 224, 224
for B in (8,):
    for b in range(B):
        enc = synth.make_cameras(N, seed=2000 + b)
        md = synth.make_matches(enc, H, W, per_pair=300, seed=2000 + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = torch.cat([synth.perturb_pose(synth.make_cameras(N, seed=2000 + b), seed=7 + b) for b in range(B)]).to(dev)
    for k, pw in ((0, 1), (0, 6), (8, 1), (8, 6)):
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
        print(f"B={B} k={k} wave={pw-1}: [P3a {pr['P3a']:.0f} wait {pr['P3_wait1']:.0f} P3b {pr['P3b']:.0f}] {dt*1e3:.2f} ms / 700 it = {dt/700*1e6:.2f} us/it; cycles/it P1 {pr['P1']:.0f} P2 {pr['P2']:.0f} xchg {pr['xchg']:.0f} P3 {pr['P3']:.0f} P4 {pr['P4']:.0f} total {tot:.0f} (iters {pr['iters']}) -> {tot/(dt/700*1e6):.0f} cycles/us", flush=True)
    eng.ggs_prof(False)
for (B, N2) in [(1, 20), (8, 20)]:
    z = synth.make_z(B, N2).to(dev)
    noise = torch.randn(101, B, N2, 9, device=dev)
    eng.sample(z, noise, 0, None, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3):
        eng.sample(z, noise, 0, None, use_graph=True)
    torch.cuda.synchronize()
    print(f"sample GGS-off B={B}: {(time.time()-t0)/3*1e3:.2f} ms/100 steps; step {eng.time_kernel(0, B, N2, reps=20)*1e3:.1f} us", flush=True)
