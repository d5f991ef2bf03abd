"""Same-box A / B of the small-batch denoiser step between engine libraries: us per step alone (pd_time_kernel) at B = 1 and B = 8, N = 20,
and a checksum of one step (parity is the tests' business).   python tools/den_ab.py [libA.so libB.so ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from posediffusion_amd import synth
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, draw_noise
    dev = torch.device("cuda:0")
    N = 20
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    diff = diff.to(dev)
    for B in (1, 8):
        eng = PoseEngine(denoiser_state(diff.model), {n: v for n, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
        z = synth.make_z(B, N).to(dev)
        x = torch.randn(B, N, 9, generator=torch.Generator().manual_seed(3))
        out = eng.denoise(x.to(dev), z, 40).cpu().double()
        err = float(out.sum())
        noise = draw_noise((B, N, 9), 100, dev)
        eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
        torch.cuda.synchronize()
        import time
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            eng.sample(z, noise, 0, None, use_graph=True, want_process=False)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"  B={B}: step alone {eng.time_kernel(0, B, N, reps=50) * 1e3:7.1f} us; 100-step pass (hipGraph) {min(ts):7.3f} ms; checksum of one step {err:.9f}", flush=True)
        eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "gpurun_ab", "libpd_engine_r3.so"), os.path.join(ROOT, "posediffusion_amd", "lib", "libpd_engine.so")]
        for rnd in range(2):
            for lib in libs:
                print(f"{os.path.relpath(lib, ROOT)} (round {rnd}):", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, PD_ENGINE_LIB=lib), check=False)
