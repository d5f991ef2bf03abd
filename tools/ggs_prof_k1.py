"""In-kernel phase cycle counters of pd_ggs_kernel at the bench shape (64 sequences x 57 000 matches, one workgroup per
sequence), for waves 0..7 of workgroup 0, with and without the LDS-DMA staged match pass.  Cycles per iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

dev = torch.device("cuda:0")
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 20
diff = synth.make_diffuser(seed=0).to(dev)
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
x0 = []
for b in range(B):
    enc = synth.make_cameras(N, seed=2000 + (b % 4))
    md = synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + (b % 4))
    eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0.append(synth.perturb_pose(enc, seed=7 + b))
x0 = torch.cat(x0).to(dev)
for reserved in [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0', '2'])]:
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=reserved)
    eng.ggs_guide(x0, 0, cfg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.ggs_guide(x0, 0, cfg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"reserved={reserved}: launch {ms:.2f} ms = {ms * 1e3 / 700:.2f} us / iteration")
    for w in [int(v) for v in os.environ.get('PD_PROF_WAVES', '0,7').split(',')]:
        eng.ggs_prof(1 + w)
        eng.ggs_guide(x0, 0, cfg)
        p = eng.ggs_prof(1 + w)
        tot = sum(p[k] for k in ("P1", "P2", "xchg", "P3", "P4"))
        print(f"  wave {w}: " + " ".join(f"{k}={p[k]:.0f}" for k in ("P1", "P2", "xchg", "P3", "P4", "P3a", "P3_wait1", "P3b", "P2_claim", "P2_issue", "P2_wait", "P2_pass", "P2_reduce")) + f" total={tot:.0f} cycles")
    eng.ggs_prof(0)
