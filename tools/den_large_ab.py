"""Same-box A / B of the LARGE-batch denoiser step between engine libraries: us per step alone (pd_time_kernel) at 5 120 and 2 060 token rows
in the default fp16-plane mode, and a sha256 of one step's output (bitwise comparison between libraries; parity is the tests' business).
python tools/den_large_ab.py [libA.so libB.so ...]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from posediffusion_amd import synth
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    dev = torch.device("cuda:0")
    N = 20
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    diff = diff.to(dev)
    for B in (256, 103):
        eng = PoseEngine(denoiser_state(diff.model), {n: v for n, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
        z = synth.make_z(B, N).to(dev)
        x = torch.randn(B, N, 9, generator=torch.Generator().manual_seed(3))
        out = eng.denoise(x.to(dev), z, 40).cpu().contiguous()
        h = hashlib.sha256(out.numpy().tobytes()).hexdigest()[:16]
        ts = [eng.time_kernel(0, B, N, reps=30) * 1e3 for _ in range(3)]
        print(f"  B={B} ({B * N} rows): step alone {min(ts):7.1f} us (of {[round(t, 1) for t in ts]}); sha256 of one step {h}; finite {bool(torch.isfinite(out).all())}", flush=True)
        if hasattr(eng.lib, "pd_engine_get_option"):         # round 5: the two-launch attention path of the same library beside the fused kernel
            eng.set_option(5, 0)
            out0 = eng.denoise(x.to(dev), z, 40).cpu().contiguous()
            ts = [eng.time_kernel(0, B, N, reps=30) * 1e3 for _ in range(3)]
            print(f"      PD_OPT_DENOISER_FUSED_ATTN = 0: step alone {min(ts):7.1f} us; bitwise equal to the fused path: {bool(torch.equal(out0, out))}", flush=True)
        eng.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "gpurun_ab", "libpd_base.so"), os.path.join(ROOT, "posediffusion_amd", "lib", "libpd_engine.so")]
        for rnd in range(2):
            for lib in libs:
                print(f"{os.path.relpath(lib, ROOT)} (round {rnd}):", flush=True)
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, PD_ENGINE_LIB=os.path.abspath(lib)), check=False)
