"""Host glue between the drop-in modules and the HIP engine.

* engine cache keyed on a fingerprint of the live parameters (data_ptr + in-place version), so
  ``load_state_dict`` / ``.to(device)`` transparently rebuild the repacked weights;
* recognition of the shipped guidance plug-in: ``functools.partial(geometry_guided_sampling,
  matches_dict=..., GGS_cfg=...)`` (demo.py:89, test.py:186) is turned into one match upload +
  the fused GGS kernel; any other callable ``cond_fn(model_mean, t)`` still works through the
  step-level API (denoiser and DDPM update stay on the HIP kernels);
* the reference's RNG call order (gaussian_diffuser.py:289, :276-278) so that the same seed draws
  the same noise the reference would draw on this device.
"""
from __future__ import annotations

import functools
from typing import Dict, Optional

import torch

from .engine import PoseEngine, make_ggs_cfg

_DENOISER_PREFIXES = ("time_embed.", "_first.", "_trunk.", "_last.")


def _fingerprint(denoiser: torch.nn.Module, diffuser: Optional[torch.nn.Module]):
    fp_den = tuple((p.data_ptr(), p._version, tuple(p.shape)) for p in denoiser.parameters())
    fp_diff = None
    if diffuser is not None:
        fp_diff = tuple((b.data_ptr(), b._version) for n, b in diffuser.named_buffers(recurse=False))
    return fp_den, fp_diff


def denoiser_state(denoiser: torch.nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in denoiser.state_dict().items() if k.startswith(_DENOISER_PREFIXES)}


def get_engine(denoiser: torch.nn.Module, diffuser: Optional[torch.nn.Module], B: int, N: int) -> PoseEngine:
    """Engine for these live modules (rebuilt when weights, device or capacity change)."""
    dev = next(denoiser.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("the PoseDiffusion sampling path of posediffusion_amd runs only on an AMD GPU "
                           f"(model is on {dev}); move the model with .to('cuda'). There is no CPU fallback.")
    fp_den, fp_diff = _fingerprint(denoiser, diffuser)
    cache = denoiser.__dict__.setdefault("_pd_engine_cache", {})
    ent = cache.get("e")
    # Denoiser.forward alone (diffuser None) never needs the schedule tables: any engine built from
    # these weights serves it.  With a diffuser the tables must match too.
    objective = getattr(diffuser, "objective", None)           # Denoiser.forward alone does not depend on it
    if ent is not None and ent[0][0] == fp_den and (fp_diff is None or ent[0][1] == fp_diff) \
            and ent[1].max_B >= B and ent[1].max_N >= N and objective in (None, ent[1].objective):
        return ent[1]
    if ent is not None:
        B, N = max(B, ent[1].max_B), max(N, ent[1].max_N)     # never shrink capacity on a rebuild
        ent[1].close()
    fp = (fp_den, fp_diff)
    if diffuser is not None:
        tables = {n: b for n, b in diffuser.named_buffers(recurse=False)}
    else:
        from .schedule import diffusion_buffers
        tables = diffusion_buffers()
    layers = len(denoiser._trunk.layers)
    nhead = denoiser._trunk.layers[0].self_attn.num_heads
    eng = PoseEngine(denoiser_state(denoiser), tables, device=dev, max_B=max(B, 1), max_N=max(N, 1),
                     num_layers=layers, nhead=nhead, objective=objective or "pred_noise")
    cache["e"] = (fp, eng)
    _ENGINES[dev.index if dev.index is not None else torch.cuda.current_device()] = eng
    return eng


def parse_ggs_cond_fn(cond_fn):
    """-> (matches_dict, GGS_cfg) if cond_fn is the shipped GGS partial, else None."""
    if not isinstance(cond_fn, functools.partial):
        return None
    if getattr(cond_fn.func, "__name__", "") != "geometry_guided_sampling":
        return None
    kw = cond_fn.keywords or {}
    if "matches_dict" not in kw or "GGS_cfg" not in kw or cond_fn.args:
        return None
    return kw["matches_dict"], dict(kw["GGS_cfg"])


def draw_noise(shape, timesteps: int, device, guided_from: int = 0, has_cond: bool = False,
               generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """[T+1, *shape] noise in the reference's draw order: randn(shape) (gaussian_diffuser.py:289), then
    one randn_like per step t = T-1..0 that is unguided and has t > 0 (:276-278); other slots stay 0."""
    out = torch.zeros((timesteps + 1, *shape), device=device, dtype=torch.float32)
    # randn straight into the slot (`out=` a contiguous slice): the same generator calls in the same order with the same shapes as
    # the reference, hence the same values, without a temporary + copy kernel per step (tests/test_host_cpu.py pins the equality)
    torch.randn(shape, out=out[0], generator=generator)
    for step in range(timesteps):
        t = timesteps - 1 - step
        guided = has_cond and t < guided_from
        if not guided and t > 0:
            torch.randn(shape, out=out[step + 1], generator=generator)
    return out


def ggs_stage_iters(iter_num: int):
    """Iterations each of the five GGS_optimize calls of one guided step is given (geometry_guided_sampling.py:48-63, :86-87) -- asked of
    the engine's own stage table (pd_ggs_stage_iters), so that the drop line below cannot drift from the rule the kernels run."""
    import ctypes as C
    from . import _lib
    out = (C.c_int * 5)()
    cfg = make_ggs_cfg(iter_num=int(iter_num))
    _lib.check(_lib.load().pd_ggs_stage_iters(C.byref(cfg), out), "pd_ggs_stage_iters")
    return tuple(int(v) for v in out)


def print_ggs_stats(stats, t: int, iter_num: int, out=None):
    """The lines the reference prints for one guided step (geometry_guided_sampling.py:104-108, :124), from the engine's per-stage
    statistics ``stats`` [B, 5, 4] = {sampson_to_print, iterations stepped, last n_valid, last loss}: a stage that stepped fewer
    iterations than it was given left through the `min_matches` break and printed the drop line first.  One line per GGS_optimize
    call as in the reference, which is defined for B = 1: its lines exactly.  A batch (this package's list-of-matches extension) prints
    every sequence's lines, sequence after sequence, each prefixed ``[b] `` -- so a break in ANY sequence is shown."""
    import sys
    out = out or sys.stdout
    st = stats.detach().cpu() if hasattr(stats, "detach") else stats
    given_all = ggs_stage_iters(int(iter_num))
    B = int(st.shape[0])
    for b in range(B):
        pre = f"[{b}] " if B > 1 else ""
        for s, given in enumerate(given_all):
            if int(st[b, s, 1]) < given:
                print(pre + "Drop this pair because of insufficient valid matches", file=out)      # :107
            print(f"{pre}t={t:02d} | sampson={float(st[b, s, 0]):05f}", file=out)                  # :124


_ENGINES = {}   # device index -> most recently built engine (used by the free functions of dropin/util)


def current_engine(device) -> PoseEngine:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("posediffusion_amd runs only on an AMD GPU; got tensors on " + str(dev))
    eng = _ENGINES.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if eng is None:
        raise RuntimeError("no PoseEngine exists on this device yet: build one with "
                           "posediffusion_amd.host.get_engine(denoiser, diffuser, B, N) or run the model once")
    return eng


def _match_fingerprint(md):
    """Cheap content fingerprint of a matches_dict (shape, three sums and an order-sensitive hash of a strided sample):
    catches in-place edits of arrays the cache still holds a reference to -- row permutations included, which keep the
    sums; identity alone is not enough (CPython recycles ids, arrays can be rewritten)."""
    import zlib
    import numpy as np
    kp1, kp2, i12 = md["kp1"], md["kp2"], md["i12"]
    step = max(1, len(kp1) // 256)
    sample = zlib.crc32(np.ascontiguousarray(kp1[::step]).tobytes()) ^ zlib.crc32(np.ascontiguousarray(kp2[::step]).tobytes()) * 3 ^ \
        zlib.crc32(np.ascontiguousarray(i12[::step]).tobytes()) * 5
    return (tuple(kp1.shape), float(kp1.sum()), float(kp2.sum()), int(i12.sum()), int(sample), tuple(int(v) for v in md["img_shape"]))


def has_matches(md) -> bool:
    """demo.py:79-92 guards `kp1 is None` (hloc found nothing): such a dict means "sample without GGS"."""
    return md is not None and md.get("kp1") is not None and len(md["kp1"]) > 0


def upload_matches(engine: PoseEngine, matches, B: int):
    """matches: one reference-style matches_dict (B == 1) or a list of B of them.  Uploads are cached per slot so
    the five calls per guided step upload once: a slot is skipped only when it still holds the SAME arrays (the cache
    keeps references, so their ids cannot be recycled) with the same content fingerprint."""
    lst = list(matches) if isinstance(matches, (list, tuple)) else [matches]
    if len(lst) != B:
        raise ValueError(f"GGS needs one matches_dict per sequence: got {len(lst)} for B={B} "
                         "(the reference defines GGS only for B = 1, geometry_guided_sampling.py:16)")
    cache = engine.__dict__.setdefault("_match_ids", {})
    for b, md in enumerate(lst):
        if not has_matches(md):
            raise ValueError(f"matches_dict of sequence {b} holds no matches (kp1 is None or empty); "
                             "call the sampler without cond_fn for such a sequence (demo.py:79-92)")
        fp = _match_fingerprint(md)
        ent = cache.get(b)
        if ent is not None and ent[0] is md["kp1"] and ent[1] is md["kp2"] and ent[2] is md["i12"] and ent[3] == fp:
            continue
        engine.set_matches(b, md["kp1"], md["kp2"], md["i12"], tuple(md["img_shape"]))
        cache[b] = (md["kp1"], md["kp2"], md["i12"], fp)


def pack_matches(matches_list, pin: bool = True):
    """CSR packing of per-sequence matches_dicts for ``PoseEngine.set_matches_async``: -> (kp1 [total,2] float64,
    kp2, i12 [total,2] int64, offsets [n+1] int64, img_shape).  ``pin``: in pinned host memory, so that the upload
    (a non_blocking copy, or the ingestion kernels reading it in place) never stages through pageable memory."""
    import numpy as np
    shapes = {tuple(int(v) for v in md["img_shape"]) for md in matches_list}
    if len(shapes) != 1:
        raise ValueError(f"all sequences of a batch must share img_shape (frame count and image size): {sorted(shapes)}")
    offsets = np.zeros(len(matches_list) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(md["kp1"]) for md in matches_list])
    total = int(offsets[-1])
    kp1 = torch.empty(total, 2, dtype=torch.float64, pin_memory=pin)
    kp2 = torch.empty(total, 2, dtype=torch.float64, pin_memory=pin)
    i12 = torch.empty(total, 2, dtype=torch.int64, pin_memory=pin)
    for b, md in enumerate(matches_list):
        a, e = int(offsets[b]), int(offsets[b + 1])
        kp1[a:e] = torch.from_numpy(np.ascontiguousarray(md["kp1"], dtype=np.float64))
        kp2[a:e] = torch.from_numpy(np.ascontiguousarray(md["kp2"], dtype=np.float64))
        i12[a:e] = torch.from_numpy(np.ascontiguousarray(md["i12"], dtype=np.int64))
    return kp1, kp2, i12, offsets, shapes.pop()
