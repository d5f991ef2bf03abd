"""posediffusion_amd -- MI355X-native sampling engine for PoseDiffusion (gfx950 / CDNA4).

Scope: ONE hot path of facebookresearch/PoseDiffusion -- the DDPM reverse loop over camera-pose
tokens (GaussianDiffusion.p_sample_loop + the transformer Denoiser) and the Geometry-Guided
Sampling step -- as hand-written HIP kernels behind a C-ABI (include/pd_engine.h), plus the
host-side mirror of the reference's Python interface for that path (``dropin/``).

    csrc/      HIP kernels + C-ABI  -> lib/libpd_engine.so   (no CPU fallback)
    _lib.py    ctypes binding        engine.py  PoseEngine wrapper
    host.py    engine cache / cond_fn recognition for the drop-in modules
    dropin/    `models` + `util` packages with the reference's names (Hydra _target_ registry)
    synth.py   synthetic weights / features / matches for tests and bench
    shard.py   one-process-per-GPU sharding of independent sequences + final gather
"""
import os

DROPIN_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")

__all__ = ["DROPIN_PATH"]
