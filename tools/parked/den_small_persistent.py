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
eng.set_persistent_denoiser(True)
eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
torch.cuda.synchronize()
ts = []
for _ in range(P):
    t0 = time.perf_counter()
    eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"B={B}, one persistent launch per evaluation (PD_OPT_DENOISER_PERSISTENT = 1): {min(ts):.3f} ms per pass = {min(ts) * 10:.1f} us per step; step alone {eng.time_kernel(0, B, N, reps=50) * 1e3:.1f} us")
eng.set_persistent_denoiser(False)
eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
torch.cuda.synchronize()
ts = []
for _ in range(P):
    t0 = time.perf_counter()
    eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"B={B}, multi-launch path (PD_OPT_DENOISER_PERSISTENT = 0): {min(ts):.3f} ms per pass = {min(ts) * 10:.1f} us per step; step alone {eng.time_kernel(0, B, N, reps=50) * 1e3:.1f} us")
import ctypes as C
eng.set_persistent_denoiser(True)
x = torch.randn(B, N, 9, device=dev)
for _ in range(3):
    eng.denoise(x, z, 50)
buf = (C.c_uint * 56)()
eng.lib.pd_debug_small_clocks(eng._h, buf)
v = list(buf)
d = [(v[i + 1] - v[i]) & 0xffffffff for i in range(43)]
names = ["emb", "first"] + [f"L{l}.{p}" for l in range(8) for p in ("qkv", "attn", "out", "ff1", "ff2")] + ["last0"]
print("persistent kernel, us per phase (start of phase -> workgroup 0 leaves its barrier):")
print("  " + "  ".join(f"{n} {t / 100:.2f}" for n, t in zip(names[:12], d[:12])))
print("  layers 1..7 mean: " + "  ".join(f"{p} {sum(d[2 + 5 * l + i] for l in range(1, 8)) / 700:.2f}" for i, p in enumerate(("qkv", "attn", "out", "ff1", "ff2"))) + f"  last0 {d[42] / 100:.2f}  total {sum(d) / 100:.1f}")
print("  layer 1 QKV phase of workgroup 0 (us): A load + LayerNorm %.2f  MFMA %.2f  reduce + store %.2f  second tile %.2f  prefetch issue %.2f  to arrive %.2f  arrive (release) %.2f  wait %.2f" %
      tuple(((v[b] - v[a]) & 0xffffffff) / 100 for a, b in ((44, 45), (45, 46), (46, 47), (47, 48), (48, 49), (49, 50), (50, 51), (51, 8))))
