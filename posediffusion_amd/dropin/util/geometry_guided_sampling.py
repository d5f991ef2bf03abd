"""geometry_guided_sampling (pose_diffusion/util/geometry_guided_sampling.py:14-64) on the HIP engine.

``functools.partial(geometry_guided_sampling, matches_dict=..., GGS_cfg=...)`` is the guidance
plug-in of the reference (demo.py:89).  GaussianDiffusion.sample recognises that partial and runs
the whole guided loop inside one captured graph; calling this function directly (any cond_fn
protocol: ``cond_fn(model_mean [1,N,9], t) -> [1,N,9]``) uploads the matches once per dict and
runs the fused five-stage kernel."""
import os

import torch

from posediffusion_amd.engine import make_ggs_cfg


@torch.no_grad()
def geometry_guided_sampling(model_mean: torch.Tensor, t: int, matches_dict, GGS_cfg, engine=None):
    from posediffusion_amd.host import current_engine, print_ggs_stats, upload_matches
    if model_mean.shape[0] != 1 and not isinstance(matches_dict, (list, tuple)):
        raise ValueError("GGS is defined per sequence: pass one matches_dict per batch element (list) for B > 1")
    if engine is None:
        engine = current_engine(model_mean.device)
    upload_matches(engine, matches_dict, model_mean.shape[0])
    out, stats = engine.ggs_guide(model_mean, t, make_ggs_cfg(GGS_cfg))
    engine.check_async()      # a bounded cross-workgroup spin that gave up must raise, not return garbage (and is cleared)
    if os.environ.get("PD_GGS_VERBOSE", "1") not in ("", "0"):     # the reference prints unconditionally (:124); PD_GGS_VERBOSE=0 mutes
        print_ggs_stats(stats, int(t), int(dict(GGS_cfg).get("iter_num", 100)))   # geometry_guided_sampling.py:104-108, :124
    return out
