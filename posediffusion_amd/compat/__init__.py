"""Optional-dependency shims for the drop-in path.

The reference's entry point (pose_diffusion/demo.py:17-33) imports omegaconf, hydra, pytorch3d and visdom
unconditionally; none of them is installed in the build image and none is part of the hot path.  Real packages are
used whenever they are importable.  Otherwise:

* this module provides what the drop-in modules themselves need: ``AttrDict`` (an OmegaConf-DictConfig-like node),
  ``instantiate`` (Hydra's ``_target_`` protocol, the plug-in registry of cfgs/default.yaml:17-40) and a
  ``PerspectiveCameras`` container;
* ``shims/`` holds minimal stand-ins for the third-party top-level packages demo.py imports (``omegaconf``, ``hydra``,
  ``pytorch3d.{renderer.cameras, ops, implicitron.tools, vis.plotly_vis, transforms, utils}``, ``visdom``);
  ``install_shims()`` appends that directory to ``sys.path`` -- appended, so an installed package always wins;
* ``posediffusion_amd/run_reference.py`` runs an UNMODIFIED reference script (demo.py) on top of both.
"""
from __future__ import annotations

import importlib
import os
import sys

import torch

SHIMS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def _wrap(v):
    if isinstance(v, AttrDict):
        return v
    if isinstance(v, dict):
        return AttrDict(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(e) for e in v]
    return v


class AttrDict(dict):
    """dict with attribute access and in-place nested nodes (stands in for an OmegaConf DictConfig: ``cfg.GGS.enable``,
    ``cfg.GGS.pose_encoding_type = ...`` as at demo.py:86, ``dict(cfg)``, ``cfg["key"]``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            super().__setitem__(k, _wrap(v))

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v


def to_container(cfg):
    """Plain dict / list copy of a config node (OmegaConf.to_container, demo.py:87)."""
    if isinstance(cfg, dict):
        return {k: to_container(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [to_container(v) for v in cfg]
    return cfg


def _local_instantiate(cfg, *args, _recursive_=True, **kwargs):
    """hydra.utils.instantiate for the reference's usage: ``_target_`` is a dotted path resolved by import (the
    `models` package is the registry), remaining keys are keyword arguments.  Every call site of the reference passes
    ``_recursive_=False`` (demo.py:46, pose_diffusion_model.py:56-63, denoiser.py:47): nested nodes go through as
    config nodes; with ``_recursive_=True`` nested nodes that carry a ``_target_`` are instantiated first, as Hydra does."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.update(kwargs)
    mod, _, name = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    conv = {}
    for k, v in cfg.items():
        if _recursive_ and isinstance(v, dict) and "_target_" in v:
            conv[k] = _local_instantiate(v, _recursive_=True)
        else:
            conv[k] = _wrap(v)
    return fn(*args, **conv)


try:  # pragma: no cover - hydra is absent in the build image
    from hydra.utils import instantiate as _hydra_instantiate  # type: ignore
    instantiate = _local_instantiate if getattr(_hydra_instantiate, "__pd_shim__", False) else _hydra_instantiate
except Exception:  # noqa: BLE001
    instantiate = _local_instantiate


class _LocalPerspectiveCameras:
    """Container with the attributes the sampling path and demo.py read: R [n,3,3], T [n,3],
    focal_length [n,2] (PyTorch3D NDC, principal point 0); numpy inputs and ``device=`` as at demo.py:122-124."""

    def __init__(self, focal_length=None, R=None, T=None, device=None, principal_point=None):
        conv = lambda a: torch.as_tensor(a).to(device) if device is not None else torch.as_tensor(a)   # noqa: E731
        self.R, self.T, self.focal_length = conv(R), conv(T), conv(focal_length)
        self.principal_point = torch.zeros_like(self.focal_length) if principal_point is None else conv(principal_point)
        self.device = device if device is not None else self.R.device

    def __len__(self):
        return self.R.shape[0]

    def to(self, device):
        return _LocalPerspectiveCameras(self.focal_length, self.R, self.T, device, self.principal_point)


try:  # pragma: no cover - pytorch3d is absent in the build image
    from pytorch3d.renderer.cameras import PerspectiveCameras as _P3dCameras  # type: ignore
    PerspectiveCameras = _LocalPerspectiveCameras if getattr(_P3dCameras, "__pd_shim__", False) else _P3dCameras
except Exception:  # noqa: BLE001
    PerspectiveCameras = _LocalPerspectiveCameras


def install_shims() -> str:
    """Make ``import omegaconf / hydra / pytorch3d / visdom`` resolve to the stand-ins when the real packages are not
    installed (sys.path APPEND: installed packages keep priority).  Returns the shim directory."""
    if SHIMS_PATH not in sys.path:
        sys.path.append(SHIMS_PATH)
    return SHIMS_PATH
