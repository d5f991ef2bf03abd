"""Development timing of one large-batch denoiser step (5 120 token rows = 256 sequences x 20 frames), alone on the chip, and the kernels it is made of.
    python tools/den_large.py [sequences=256]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from posediffusion_amd import synth  # noqa: E402
from posediffusion_amd.engine import PoseEngine  # noqa: E402
from posediffusion_amd.host import denoiser_state  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
diff = synth.make_diffuser(seed=0).to(dev)
tables = {k: v for k, v in diff.named_buffers(recurse=False)}
eng = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=20)
x = torch.randn(B, 20, 9, device=dev)
z = synth.make_z(B, 20).to(dev)
eng.denoise(x, z, 50)
torch.cuda.synchronize()
for _ in range(3):
    ms = eng.time_kernel(0, B, 20, None, reps=20)
    gf = B * 20 * 34.73e-3
    print(f"denoiser step at {B * 20} token rows: {ms * 1e3:.0f} us = {gf / ms:.1f} TFLOP/s = {gf / ms / 157.3:.3f} of exact-fp32 MFMA")
