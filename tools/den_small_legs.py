"""The latency legs of the small-batch denoiser chain (B = 1, N = 20: 43 dependent launches per evaluation), measured INSIDE the kernels:
a library built with -DPD_DEN_STAMPS (tools/build_variant.sh stamps "-DPD_DEN_STAMPS") records the constant 100 MHz clock (s_memrealtime,
10 ns ticks, comparable across kernels and CUs) in block 0 of every launch at: 0 entered, 1 own share of the A rows loaded + normalised +
written to LDS, 2 every wave's share staged (barrier), 3 first weight batch + bias landed, 4 MFMA chain issued, 5 reduction + epilogue issued,
6 stores drained.  Printed per launch of the LAST of three evaluations: the gap since the previous launch's last stamp (= the launch boundary
as the next kernel's first wave sees it) and the legs in ns.
    PD_ENGINE_LIB=gpurun_ab/libpd_stamps.so python tools/den_small_legs.py [B=1]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posediffusion_amd import synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = 20
dev = torch.device("cuda:0")
diff = synth.make_diffuser(seed=0).to(dev)
eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
if not hasattr(eng.lib, "pd_debug_den_stamps"):
    raise SystemExit("this library was not built with -DPD_DEN_STAMPS (tools/build_variant.sh stamps \"-DPD_DEN_STAMPS\"; PD_ENGINE_LIB=gpurun_ab/libpd_stamps.so)")
x = torch.randn(B, N, 9, device=dev)
z = synth.make_z(B, N).to(dev)
buf = (C.c_longlong * (8 * 256))()
for _ in range(3):
    eng.denoise(x, z, 50)
torch.cuda.synchronize()
eng.lib.pd_debug_den_stamps(buf, 1)          # reset
reps = 3
for _ in range(reps):
    eng.denoise(x, z, 50)
torch.cuda.synchronize()
assert eng.lib.pd_debug_den_stamps(buf, 43 * reps) == 0
names = ["_first"] + [f"L{l}.{k}" for l in range(8) for k in ("qkv", "attn", "out", "ff1", "ff2")] + ["_last.0", "tail"]
rows = []
for i in range(43 * reps):
    st = [buf[8 * i + j] for j in range(7)]
    rows.append(st)
print(f"B = {B}, N = {N}: evaluation {reps} of {reps} (ns; s_memrealtime ticks of 10 ns); GEMM legs: A loaded | staged (barrier) | weights landed | MFMA chain | reduce + epilogue | stores drained")
print(f"{'launch':10s} {'gap':>6s} {'A':>6s} {'bar':>6s} {'W':>6s} {'mfma':>6s} {'epi':>6s} {'drain':>6s} {'body':>6s}")
tot_gap = tot_body = 0
sums = {}
base = 43 * (reps - 1)
for k in range(43):
    st, prev = rows[base + k], rows[base + k - 1]
    last_prev = max(prev)
    gap = (st[0] - last_prev) * 10
    legs = []
    cur = st[0]
    for j in range(1, 7):
        if st[j] == 0:
            legs.append(None)
        else:
            legs.append((st[j] - cur) * 10)
            cur = st[j]
    body = (max(st) - st[0]) * 10
    tot_gap += gap
    tot_body += body
    kind = names[k].split(".")[-1] if names[k].startswith("L") else names[k]
    a = sums.setdefault(kind, [0, 0, 0])
    a[0] += gap; a[1] += body; a[2] += 1
    print(f"{names[k]:10s} {gap:6d} " + " ".join(f"{v:6d}" if v is not None else "     -" for v in legs) + f" {body:6d}")
print(f"sum of gaps {tot_gap / 1e3:.1f} us + sum of bodies {tot_body / 1e3:.1f} us = {(tot_gap + tot_body) / 1e3:.1f} us per evaluation (block 0's view; untimed product step: see tools/den_ab.py)")
for kind, (g_, b_, n_) in sums.items():
    print(f"  {kind:8s} x{n_:2d}: gap {g_ / n_:7.0f} ns, body {b_ / n_:7.0f} ns")
ms = eng.time_kernel(0, B, N, None, reps=50)
print(f"pd_time_kernel with this (stamped, draining) library: {ms * 1e3:.1f} us per evaluation")
