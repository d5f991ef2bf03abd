"""Optional-dependency shims: real hydra / pytorch3d are used when importable; otherwise these
minimal stand-ins keep the reference's call sites (`instantiate(cfg, _recursive_=False)`,
`PerspectiveCameras(focal_length=, R=, T=)`) working for the sampling path."""
from __future__ import annotations

import importlib

import torch


class AttrDict(dict):
    """dict with attribute access (stands in for an OmegaConf DictConfig node)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def _local_instantiate(cfg, *args, _recursive_=True, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.update(kwargs)
    mod, _, name = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    conv = {k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    return fn(*args, **conv)


try:  # pragma: no cover - hydra is absent in the build image
    from hydra.utils import instantiate  # type: ignore
except Exception:  # noqa: BLE001
    instantiate = _local_instantiate


class _LocalPerspectiveCameras:
    """Container with the attributes the sampling path and demo.py read: R [n,3,3], T [n,3],
    focal_length [n,2] (PyTorch3D NDC, principal point 0)."""

    def __init__(self, focal_length=None, R=None, T=None, device=None, principal_point=None):
        self.R = torch.as_tensor(R)
        self.T = torch.as_tensor(T)
        self.focal_length = torch.as_tensor(focal_length)
        self.principal_point = (torch.zeros_like(self.focal_length) if principal_point is None
                                else torch.as_tensor(principal_point))
        self.device = device if device is not None else self.R.device

    def __len__(self):
        return self.R.shape[0]

    def to(self, device):
        return _LocalPerspectiveCameras(self.focal_length.to(device), self.R.to(device), self.T.to(device), device,
                                        self.principal_point.to(device))


try:  # pragma: no cover - pytorch3d is absent in the build image
    from pytorch3d.renderer.cameras import PerspectiveCameras  # type: ignore
except Exception:  # noqa: BLE001
    PerspectiveCameras = _LocalPerspectiveCameras
