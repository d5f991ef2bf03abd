"""Run the UNMODIFIED reference hot-path files on CPU in the build container.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only where /root/reference exists
(the build container); the GPU box never has it, so nothing under ``-m gpu``, ``smoke()`` or
``bench.py`` may call this module.  ``oracle/make_golden.py`` uses it to produce the committed
fixtures under ``tests/golden/`` and ``tests/test_oracle_golden.py`` uses it (when available)
to re-validate the oracle restatement against the live reference.

What is stubbed (absent third-party modules):
  * ``torchvision``           -- imported but unused by models/gaussian_diffuser.py:24
  * ``hydra.utils.instantiate`` -- denoiser.py:16,48; resolved through a tiny registry
  * ``pytorch3d``             -- five helpers restated in oracle/pd_oracle.py from the published
                                 0.7.x algorithms (HarmonicEmbedding, quaternion_to_matrix,
                                 PerspectiveCameras, opencv_from_cameras_projection, hat)
No reference source is copied: the files are executed from where they lie.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import torch

from . import pd_oracle as O

REF_ROOT = os.environ.get("PD_REFERENCE_ROOT", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "pose_diffusion")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_PKG, "models", "gaussian_diffuser.py"))


class AttrDict(dict):
    """Minimal stand-in for an OmegaConf DictConfig (attribute + item access)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e


class _HarmonicEmbedding(torch.nn.Module):
    def __init__(self, n_harmonic_functions=6, omega_0=1.0, logspace=True, append_input=True):
        super().__init__()
        assert logspace and omega_0 == 1.0 and append_input
        self.n = n_harmonic_functions

    def get_output_dim(self, input_dims=3):
        return input_dims * (2 * self.n + 1)

    def forward(self, x):
        return O.harmonic_embedding(x, self.n)


class _PerspectiveCameras:
    def __init__(self, focal_length=None, R=None, T=None, device=None, principal_point=None):
        self.R, self.T = torch.as_tensor(R), torch.as_tensor(T)
        self.focal_length = torch.as_tensor(focal_length)
        self.device = device if device is not None else self.R.device

    def __len__(self):
        return self.R.shape[0]


def _opencv_from_cameras_projection(cameras, image_size):
    h, w = int(image_size[0, 0]), int(image_size[0, 1])
    return O.opencv_from_cameras_projection(cameras.R, cameras.T, cameras.focal_length, h, w)


_REGISTRY = {}


def _instantiate(cfg, *args, _recursive_=True, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.update(kwargs)
    fn = _REGISTRY[target]
    wrapped = {k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    return fn(*args, **wrapped)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = None


def load_reference():
    """Import the reference hot-path modules; returns a namespace with
    GaussianDiffusion, Denoiser, TransformerEncoderWrapper, geometry_guided_sampling,
    GGS_optimize, compute_sampson_distance, get_fundamental_matrices, pose_encoding_to_camera."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    if "pytorch3d" in sys.modules or "hydra" in sys.modules:
        raise RuntimeError("real pytorch3d/hydra present; stubs not needed -- adapt ref_stubs")
    stub_names = ("torchvision", "hydra", "hydra.utils", "pytorch3d", "pytorch3d.renderer", "pytorch3d.renderer.cameras",
                  "pytorch3d.utils", "pytorch3d.transforms", "pytorch3d.transforms.so3",
                  "pytorch3d.transforms.rotation_conversions")

    _mod("torchvision", transforms=types.SimpleNamespace(), utils=types.SimpleNamespace())
    _mod("hydra")
    _mod("hydra.utils", instantiate=_instantiate)
    _mod("pytorch3d")
    _mod("pytorch3d.renderer", HarmonicEmbedding=_HarmonicEmbedding)
    _mod("pytorch3d.renderer.cameras", CamerasBase=_PerspectiveCameras, PerspectiveCameras=_PerspectiveCameras)
    _mod("pytorch3d.utils", opencv_from_cameras_projection=_opencv_from_cameras_projection)
    _mod("pytorch3d.transforms")
    _mod("pytorch3d.transforms.so3", hat=O.hat)
    _mod("pytorch3d.transforms.rotation_conversions", quaternion_to_matrix=O.quaternion_to_matrix,
         matrix_to_quaternion=None)

    # `util` is a plain package in the reference (empty __init__); models/denoiser.py does
    # `from util.embedding import ...`.  The product's drop-in packages use the same top-level names
    # (`util`, `models`), so the reference is imported hermetically: our entries are parked, the
    # reference modules are imported with its directory first on sys.path, and afterwards every
    # `util*` / `models*` entry it created is removed again (its functions keep direct references).
    def _ours():
        return {k: v for k, v in sys.modules.items() if k in ("util", "models") or k.startswith(("util.", "models."))}

    parked = _ours()
    for k in parked:
        del sys.modules[k]
    sys.path.insert(0, REF_PKG)
    try:
        ns = types.SimpleNamespace()

        def load(name, rel):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_PKG, rel))
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            spec.loader.exec_module(m)
            return m

        gd = load("_ref_gaussian_diffuser", "models/gaussian_diffuser.py")
        dn = load("_ref_denoiser", "models/denoiser.py")
        ggs = importlib.import_module("util.geometry_guided_sampling")
        fm = importlib.import_module("util.get_fundamental_matrix")
        ct = importlib.import_module("util.camera_transform")
        assert ggs.__file__.startswith(REF_PKG) and fm.__file__.startswith(REF_PKG), "not the reference's util package"
        _REGISTRY["models.TransformerEncoderWrapper"] = dn.TransformerEncoderWrapper
        _REGISTRY["models.Denoiser"] = dn.Denoiser
        _REGISTRY["models.GaussianDiffusion"] = gd.GaussianDiffusion
        ns.GaussianDiffusion = gd.GaussianDiffusion
        ns.Denoiser = dn.Denoiser
        ns.TransformerEncoderWrapper = dn.TransformerEncoderWrapper
        ns.geometry_guided_sampling = ggs.geometry_guided_sampling
        ns.GGS_optimize = ggs.GGS_optimize
        ns.compute_sampson_distance = ggs.compute_sampson_distance
        ns.get_fundamental_matrices = fm.get_fundamental_matrices
        ns.pose_encoding_to_camera = ct.pose_encoding_to_camera
        ns.PerspectiveCameras = _PerspectiveCameras
    finally:
        sys.path.remove(REF_PKG)
        for k in list(_ours()):
            del sys.modules[k]
        sys.modules.update(parked)
        for k in stub_names:          # the stubs are only for the reference's import time
            sys.modules.pop(k, None)
    _loaded = ns
    return ns


class _MetricCameras(_PerspectiveCameras):
    """What util/metric.py:30-31 asks of a pytorch3d camera object."""

    def get_world_to_view_transform(self):
        M = O.world_to_view_matrix(self.R, self.T)
        return types.SimpleNamespace(get_matrix=lambda: M)


_loaded_metric = None


def load_reference_metric():
    """The reference's util/metric.py executed in place (stub: pytorch3d.transforms.so3_relative_angle, restated in
    the oracle).  -> namespace with camera_to_rel_deg, calculate_auc_np, compute_ARE and a Cameras class for it."""
    global _loaded_metric
    if _loaded_metric is not None:
        return _loaded_metric
    if not available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    names = ("pytorch3d", "pytorch3d.transforms")
    saved = {k: sys.modules.get(k) for k in names}
    _mod("pytorch3d")
    _mod("pytorch3d.transforms", so3_relative_angle=lambda R1, R2, eps=1e-4, **kw: O.so3_relative_angle(R1, R2, eps=eps))
    try:
        spec = importlib.util.spec_from_file_location("_ref_metric", os.path.join(REF_PKG, "util", "metric.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _loaded_metric = types.SimpleNamespace(camera_to_rel_deg=m.camera_to_rel_deg, calculate_auc_np=m.calculate_auc_np,
                                           compute_ARE=m.compute_ARE, Cameras=_MetricCameras)
    return _loaded_metric


def load_reference_extractor(net: torch.nn.Module):
    """The reference's models/image_feature_extractor.py executed in place with `torch.hub.load` answering with `net`
    (the DINO hub code is third-party and absent) and a stub `torchvision`.  -> a reference
    MultiScaleImageFeatureExtractor("dino_vits16") wrapped around `net`."""
    if not available():
        raise RuntimeError(f"reference not found under {REF_ROOT}")
    saved_tv, saved_hub = sys.modules.get("torchvision"), torch.hub.load
    _mod("torchvision", models=types.SimpleNamespace())
    torch.hub.load = lambda repo, name, *a, **k: net
    try:
        spec = importlib.util.spec_from_file_location("_ref_extractor", os.path.join(REF_PKG, "models", "image_feature_extractor.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        ext = m.MultiScaleImageFeatureExtractor(modelname="dino_vits16")
    finally:
        torch.hub.load = saved_hub
        if saved_tv is None:
            sys.modules.pop("torchvision", None)
        else:
            sys.modules["torchvision"] = saved_tv
    return ext.eval()


TRANSFORMER_CFG = {  # cfgs/default.yaml:27-35
    "_target_": "models.TransformerEncoderWrapper",
    "d_model": 512,
    "nhead": 4,
    "dim_feedforward": 1024,
    "num_encoder_layers": 8,
    "dropout": 0.1,
    "batch_first": True,
    "norm_first": True,
}


def init_weights_reference_rule(module: torch.nn.Module):
    """models/pose_diffusion_model.py:67-74 (applied with ``module.apply``)."""
    def f(m):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)
        elif isinstance(m, torch.nn.LayerNorm):
            torch.nn.init.constant_(m.bias, 0)
            torch.nn.init.constant_(m.weight, 1.0)
    module.apply(f)


def build_reference_diffuser(seed: int = 0):
    """Reference GaussianDiffusion (cfgs/default.yaml:38-40) with a seeded reference Denoiser
    attached as ``.model`` (pose_diffusion_model.py:57-63, :67-74), eval mode."""
    ref = load_reference()
    torch.manual_seed(seed)
    den = ref.Denoiser(TRANSFORMER=AttrDict(TRANSFORMER_CFG))
    init_weights_reference_rule(den)
    diff = ref.GaussianDiffusion(beta_schedule="custom")
    diff.model = den
    diff.eval()
    return diff
