"""Same-box A / B of two engine libraries on the GGS launch shapes (boxes differ by +-5 %, so two libraries are only comparable inside ONE
gpurun call).  python tools/ab_ggs.py [libA.so libB.so ...]   (default: gpurun_ab/libpd_engine_r3.so and the built library)
Per library and shape: ms per pd_ggs_guide launch (700 iterations) and us per iteration, best of 3 after a warm launch.
Shapes: B = 1 / 8 (engine's choice of workgroups per sequence: the latency shapes), B = 64 / 80 / 256 at one workgroup per sequence."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(1, 0, 0), (1, 0, 32), (8, 0, 0), (8, 0, 32), (2, 0, 0), (64, 1, 0), (80, 1, 0), (256, 1, 0)]   # (B, wgs_per_seq, cfg.reserved flags)


def child(lib):
    import ctypes as C
    import torch
    from posediffusion_amd import _lib
    _lib.LIB_PATH = lib
    probe = C.CDLL(lib)
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if hasattr(probe, k)}
    from posediffusion_amd import synth
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state
    dev = torch.device("cuda:0")
    N = int(os.environ.get("PD_AB_N", "20"))
    diff = synth.make_diffuser(seed=0).to(dev)
    mds = []
    for s in range(4):
        enc = synth.make_cameras(N, seed=2000 + s)
        mds.append((enc, synth.make_matches(enc, 224, 224, per_pair=300, seed=2000 + s)))
    shapes = [(1, 0, 0), (64, 1, 0)] if os.environ.get("PD_AB_ABLATION") else SHAPES      # ablated libraries (PD_GGS_ABLATE): no early exit on garbage sums
    if os.environ.get("PD_AB_SHAPES"):                                                   # "B,wgs,flags;B,wgs,flags;..."
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["PD_AB_SHAPES"].split(";")]
    for B, wgs, flags in shapes:
        eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
        x0 = []
        for b in range(B):
            enc, md = mds[b % 4]
            eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
            x0.append(synth.perturb_pose(enc, seed=7 + b))
        x0 = torch.cat(x0).to(dev)
        cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=wgs, reserved=flags, **({"min_matches": 0} if os.environ.get("PD_AB_ABLATION") else {}))
        eng.ggs_guide(x0, 0, cfg)
        best, iters = 1e9, 0
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out, st = eng.ggs_guide(x0, 0, cfg)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
            iters = int(st[:, :, 1].sum(dim=1).min())
        eng.check_async()
        print(f"  B={B:3d} wgs={wgs} flags={flags:2d}: {best:8.3f} ms per launch = {best * 1e3 / 700:6.2f} us / iteration   (iterations run: {iters}; checksum {float(out.double().sum()):.6f})", flush=True)
        eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "gpurun_ab", "libpd_engine_r3.so"), os.path.join(ROOT, "posediffusion_amd", "lib", "libpd_engine.so")]
        for rnd in range(2):
            for lib in libs:
                print(f"{os.path.relpath(lib, ROOT)} (round {rnd}):", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
