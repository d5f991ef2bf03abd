"""Time the image feature extractor (csrc/pd_vit.hip): n images x the reference's three scales, exact-fp32 MFMA.
usage: python tools/vit_bench.py [n_images=20] [reps=20] [size=224]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from posediffusion_amd import synth
from posediffusion_amd.vit import VitEngine, vit_state


def flops(n, H, W, sf, D=384, L=12, FF=1536, P=16):
    hs, ws = (H, W) if sf == 1 else (int(H * sf), int(W * sf))
    p = (hs // P) * (ws // P)
    t = p + 1
    per_layer = 2 * t * (D * 3 * D + D * D + 2 * D * FF) + 4 * t * t * D
    return n * (2 * p * 3 * P * P * D + L * per_layer)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ext = synth._dropin().MultiScaleImageFeatureExtractor()          # random-init DINO-shaped parameters
    eng = VitEngine(vit_state(ext._net), dev)
    x = torch.rand(n, 3, size, size, device=dev)
    scales = (1, 1 / 2, 1 / 3)
    out = {}
    for name, sc in (("three_scales", scales), ("scale_1", (1,)), ("scale_1/2", (1 / 2,)), ("scale_1/3", (1 / 3,))):
        for _ in range(3):
            eng.multiscale(x, sc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.multiscale(x, sc)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = sum(flops(n, size, size, s) for s in sc)
        out[name] = {"ms": round(ms, 4), "gflop": round(fl / 1e9, 2), "tflops": round(fl / ms / 1e9, 2), "images_per_s": round(n / ms * 1e3, 1)}
    print(json.dumps({"n_images": n, "size": size, "fp32_mfma_peak_tflops": 157.3, **out}))


if __name__ == "__main__":
    main()
