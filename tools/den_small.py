"""Small-batch denoiser: ms per 100-step GGS-off pass (hipGraph) at B sequences; run under rocprofv3 for per-kernel durations.
usage: python tools/den_small.py [B=1] [passes=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from posediffusion_amd import synth
from posediffusion_amd.engine import PoseEngine
from posediffusion_amd.host import denoiser_state, draw_noise
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
P = int(sys.argv[2]) if len(sys.argv) > 2 else 5
N = 20
dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0)
eng = PoseEngine(denoiser_state(diff.model), {n: v for n, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
z = synth.make_z(B, N).to(dev)
noise = draw_noise((B, N, 9), 100, dev)
eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
torch.cuda.synchronize()
ts = []
for _ in range(P):
    t0 = time.perf_counter()
    eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"B={B}, 43 launches per evaluation: {min(ts):.3f} ms per pass = {min(ts) * 10:.1f} us per step; step alone {eng.time_kernel(0, B, N, reps=50) * 1e3:.1f} us")
