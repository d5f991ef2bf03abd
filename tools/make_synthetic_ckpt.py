#!/usr/bin/env python
"""Write a synthetic PoseDiffusion checkpoint with the reference's exact key set (SURVEY.md section 5 "Checkpoint").

    python tools/make_synthetic_ckpt.py out.pth [--cfg /path/to/cfgs/default.yaml] [--seed 0]

The trained checkpoint (co3d_model1.pth, a Google-Drive download) is not available offline; demo.py:52-60 does
``torch.load(ckpt)`` + ``model.load_state_dict(checkpoint, strict=True)`` on the whole PoseDiffusionModel, so what it
needs is a flat state_dict whose keys are exactly the model's: ``image_feature_extractor._net.*`` (DINO ViT-S/16),
``diffuser.<13 schedule buffers>``, ``diffuser.model.{time_embed, _first, _trunk.layers.0..7, _last}.*``.  This tool
instantiates the model through the Hydra ``_target_`` registry from the yaml (the built-in copy of cfgs/default.yaml's
MODEL node when no --cfg is given), seeds torch, applies the reference's init rule (pose_diffusion_model.py:67-74, run
by the constructor) and saves ``model.state_dict()``.  Random weights: poses are meaningless, plumbing and timing are not.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEFAULT_MODEL_CFG = {  # cfgs/default.yaml:16-40
    "_target_": "models.PoseDiffusionModel",
    "pose_encoding_type": "absT_quaR_logFL",
    "IMAGE_FEATURE_EXTRACTOR": {"_target_": "models.MultiScaleImageFeatureExtractor", "freeze": False},
    "DENOISER": {"_target_": "models.Denoiser",
                 "TRANSFORMER": {"_target_": "models.TransformerEncoderWrapper", "d_model": 512, "nhead": 4, "dim_feedforward": 1024,
                                 "num_encoder_layers": 8, "dropout": 0.1, "batch_first": True, "norm_first": True}},
    "DIFFUSER": {"_target_": "models.GaussianDiffusion", "beta_schedule": "custom"},
}


def build_model(cfg_path=None, seed: int = 0):
    from posediffusion_amd import synth
    from posediffusion_amd.compat import AttrDict, instantiate
    synth._dropin()
    if cfg_path:
        import yaml
        with open(cfg_path) as f:
            model_cfg = AttrDict(yaml.safe_load(f))["MODEL"]
    else:
        model_cfg = AttrDict(DEFAULT_MODEL_CFG)
    torch.manual_seed(seed)
    return instantiate(model_cfg, _recursive_=False)


def write_checkpoint(path: str, cfg_path=None, seed: int = 0):
    model = build_model(cfg_path, seed)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.save(sd, path)
    return sd


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("out")
    ap.add_argument("--cfg", default=None, help="a reference cfg yaml (cfgs/default.yaml); default: built-in copy of its MODEL node")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    sd = write_checkpoint(a.out, a.cfg, a.seed)
    n = sum(v.numel() for v in sd.values())
    print(f"wrote {a.out}: {len(sd)} tensors, {n} values ({os.path.getsize(a.out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
