"""Synthetic inputs for tests and bench: seeded reference-rule weights,
image features z, and pairwise-consistent 2-D matches in the hloc/COLMAP-derived format the
GGS plug-in consumes (demo.py:82-84).  No datasets or checkpoints are available offline."""
from __future__ import annotations

import sys
from typing import Dict, Tuple

import numpy as np
import torch

from . import DROPIN_PATH
from .compat import AttrDict

TRANSFORMER_CFG = {  # cfgs/default.yaml:27-35
    "_target_": "models.TransformerEncoderWrapper", "d_model": 512, "nhead": 4, "dim_feedforward": 1024,
    "num_encoder_layers": 8, "dropout": 0.1, "batch_first": True, "norm_first": True,
}
GGS_CFG = {  # cfgs/default.yaml:6-13 (+ pose_encoding_type injected at demo.py:86)
    "enable": True, "start_step": 10, "learning_rate": 0.01, "iter_num": 100, "sampson_max": 10, "min_matches": 10,
    "alpha": 0.0001, "pose_encoding_type": "absT_quaR_logFL",
}


def _dropin():
    if DROPIN_PATH not in sys.path:
        sys.path.insert(0, DROPIN_PATH)
    import models  # noqa: F401  (the drop-in package)
    return models


def reference_init_(module: torch.nn.Module):
    """Weight-init rule of pose_diffusion_model.py:67-74 (trunc-normal 0.02 Linear, unit LayerNorm)."""
    def f(m):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)
        elif isinstance(m, torch.nn.LayerNorm):
            torch.nn.init.constant_(m.bias, 0)
            torch.nn.init.constant_(m.weight, 1.0)
    module.apply(f)


def make_diffuser(seed: int = 0, num_layers: int = 8):
    """Drop-in GaussianDiffusion with a seeded Denoiser attached (torch.manual_seed(seed), module
    construction, then the reference init rule -- the protocol of SURVEY.md section 8d)."""
    models = _dropin()
    torch.manual_seed(seed)
    cfg = dict(TRANSFORMER_CFG, num_encoder_layers=num_layers)
    den = models.Denoiser(TRANSFORMER=AttrDict(cfg))
    reference_init_(den)
    diff = models.GaussianDiffusion(beta_schedule="custom")
    diff.model = den
    return diff.eval()


def randomize_norm_and_bias_(denoiser: torch.nn.Module, seed: int = 1234, scale: float = 0.05):
    """The reference init leaves every bias at 0 and LayerNorm at (1, 0); tests perturb them so a
    kernel that dropped a bias / gamma / beta cannot pass."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in denoiser.named_parameters():
            if n.endswith("bias") or "norm" in n or n.startswith("_last.1"):
                p.add_(scale * torch.randn(p.shape, generator=g))


def make_z(B: int, N: int, seed: int = 1000, z_dim: int = 384) -> torch.Tensor:
    out = torch.empty(B, N, z_dim)
    for b in range(B):
        out[b] = torch.randn(N, z_dim, generator=torch.Generator().manual_seed(seed + b))
    return out


# ------------------------------------------------------------------------------------------------
# cameras + matches
# ------------------------------------------------------------------------------------------------
def _quat_to_R(q: np.ndarray) -> np.ndarray:
    r, i, j, k = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s = 2.0 / (q * q).sum(-1)
    return np.stack([1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                     s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                     s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)], -1).reshape(q.shape[:-1] + (3, 3))


def make_cameras(N: int, seed: int = 2000, f_ndc: float = 3.0) -> np.ndarray:
    """Pose encodings [N,9] = [T | quat wxyz | logFL] of cameras looking at a scene around the origin:
    T ~ (0,0,6) +- 0.3, small random rotations, focal f_ndc (log-encoded with the 1.8 bias)."""
    rng = np.random.default_rng(seed)
    enc = np.zeros((N, 9))
    enc[:, 0:3] = np.array([0.0, 0.0, 6.0]) + rng.uniform(-0.3, 0.3, (N, 3))
    ang = rng.normal(0, 0.25, (N, 3))
    q = np.concatenate([np.ones((N, 1)), 0.5 * ang], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.8, 1.25, (N, 1))          # the encoding does not normalise quaternions
    enc[:, 3:7] = q
    enc[:, 7:9] = np.log(f_ndc) - 1.8 + rng.normal(0, 0.02, (N, 2))
    return enc


def project(enc: np.ndarray, X: np.ndarray, H: int, W: int, f=None) -> Tuple[np.ndarray, np.ndarray]:
    """PyTorch3D NDC projection of world points X [P,3] into every camera: X_cam = X R + T,
    x_ndc = f X/Z, u = W/2 - x_ndc s, v = H/2 - y_ndc s (s = min(H,W)/2).  -> (uv [N,P,2], z [N,P])."""
    R = _quat_to_R(enc[:, 3:7])
    T = enc[:, 0:3]
    if f is None:   # GGS uses the mean focal of ALL cameras of the sequence (geometry_guided_sampling.py:142)
        f = np.clip(np.exp(enc[:, 7:9] + 1.8), 0.1, 20.0).mean(0)
    f = np.broadcast_to(np.asarray(f, dtype=np.float64), (len(enc), 2))
    Xc = np.einsum("pk,nkc->npc", X, R) + T[:, None, :]
    s = min(H, W) / 2.0
    u = W / 2.0 - f[:, None, 0] * Xc[..., 0] / Xc[..., 2] * s
    v = H / 2.0 - f[:, None, 1] * Xc[..., 1] / Xc[..., 2] * s
    return np.stack([u, v], -1), Xc[..., 2]


def make_matches(enc: np.ndarray, H: int = 224, W: int = 224, per_pair: int = 300, noise_px: float = 0.5,
                 outlier_frac: float = 0.1, seed: int = 2000, ordered_pairs: bool = False) -> Dict:
    """Matches for all C(N,2) pairs i<j (hloc exhaustive pairs), grouped by pair, `per_pair` each:
    3-D points in front of the cameras projected into both frames, N(0, noise_px) pixel noise,
    `outlier_frac` uniform outliers.  Returns the reference's matches_dict (kp float64, i12 int64)."""
    rng = np.random.default_rng(seed)
    N = len(enc)
    kp1, kp2, i12 = [], [], []
    pairs = [(i, j) for i in range(N) for j in range(N) if (i < j or (ordered_pairs and i != j))]
    fbar = np.clip(np.exp(enc[:, 7:9] + 1.8), 0.1, 20.0).mean(0)
    for (i, j) in pairs:
        got1, got2 = [], []
        need = per_pair
        while need > 0:
            X = rng.uniform(-1.0, 1.0, (4 * need + 16, 3))
            uv, z = project(enc[[i, j]], X, H, W, fbar)
            ok = (z > 0.5).all(0) & (uv[..., 0] >= 0).all(0) & (uv[..., 0] < W).all(0) & (uv[..., 1] >= 0).all(0) & \
                 (uv[..., 1] < H).all(0)
            sel = np.nonzero(ok)[0][:need]
            got1.append(uv[0, sel])
            got2.append(uv[1, sel])
            need -= len(sel)
        a, b = np.concatenate(got1), np.concatenate(got2)
        a = a + rng.normal(0, noise_px, a.shape)
        b = b + rng.normal(0, noise_px, b.shape)
        n_out = int(round(outlier_frac * per_pair))
        if n_out:
            idx = rng.choice(per_pair, n_out, replace=False)
            b[idx] = rng.uniform(0, [W, H], (n_out, 2))
        kp1.append(a)
        kp2.append(b)
        i12.append(np.repeat(np.array([[i, j]], dtype=np.int64), per_pair, 0))
    return {"kp1": np.concatenate(kp1).astype(np.float64), "kp2": np.concatenate(kp2).astype(np.float64),
            "i12": np.concatenate(i12).astype(np.int64), "img_shape": torch.Size((N, 3, H, W))}


def perturb_pose(enc: np.ndarray, seed: int = 7, sigma_T: float = 0.05, sigma_q: float = 0.02, sigma_f: float = 0.05):
    """A start point in the GGS basin: the true cameras plus small noise (float32 [1,N,9])."""
    rng = np.random.default_rng(seed)
    e = enc.copy()
    e[:, 0:3] += rng.normal(0, sigma_T, (len(e), 3))
    e[:, 3:7] += rng.normal(0, sigma_q, (len(e), 4))
    e[:, 7:9] += rng.normal(0, sigma_f, (len(e), 2))
    return torch.from_numpy(e[None].astype(np.float32))


# ------------------------------------------------------------------------------------------------
# pairwise-consistent matches for ARBITRARY cameras (bench: cameras = the engine's own model mean)
# ------------------------------------------------------------------------------------------------
def fundamental_matrices_np(enc: np.ndarray, H: int, W: int) -> np.ndarray:
    """Fo[i, j] (x_j^T Fo x_i = 0) for every ordered frame pair, in fp64, with the conventions of
    get_fundamental_matrix.py:38-51 + opencv_from_cameras_projection and the mean focal of
    geometry_guided_sampling.py:142."""
    N = len(enc)
    R = _quat_to_R(enc[:, 3:7])
    D = np.array([-1.0, -1.0, 1.0])
    Rc = D[None, :, None] * R.transpose(0, 2, 1)
    tc = D[None] * enc[:, 0:3]
    f = np.clip(np.exp(enc[:, 7:9] + 1.8), 0.1, 20.0).mean(0)
    s = min(H, W) / 2.0
    K = np.array([[f[0] * s, 0, W / 2.0], [0, f[1] * s, H / 2.0], [0, 0, 1.0]])
    A = np.linalg.inv(K)
    F = np.zeros((N, N, 3, 3))
    for i in range(N):
        for j in range(N):
            R12 = Rc[j] @ Rc[i].T
            t12 = tc[j] - R12 @ tc[i]
            Et = -R12.T @ t12
            Hm = np.array([[0, -Et[2], Et[1]], [Et[2], 0, -Et[0]], [-Et[1], Et[0], 0]])
            F[i, j] = A.T @ (R12 @ Hm) @ A
    return F


def make_epipolar_matches(enc: np.ndarray, H: int = 224, W: int = 224, per_pair: int = 300, noise_px: float = 0.5,
                          outlier_frac: float = 0.1, seed: int = 2000) -> Dict:
    """Matches for all C(N,2) pairs that satisfy the epipolar constraint of the given cameras pair by
    pair: x1 uniform in image i; x2 on its epipolar line in image j, within +-W/2 of the point of the
    line closest to the image centre (so x2 is inside the image whenever the line crosses it, and
    may fall outside for the wild cameras a random-weight denoiser produces -- the constraint, and
    therefore the GGS workload, is the same).  Plus N(0, noise_px) pixel noise and uniform outliers;
    grouped by pair like hloc's output."""
    rng = np.random.default_rng(seed)
    N = len(enc)
    F = fundamental_matrices_np(np.asarray(enc, dtype=np.float64), H, W)
    kp1, kp2, i12 = [], [], []
    ctr = np.array([W / 2.0, H / 2.0])
    for i in range(N):
        for j in range(i + 1, N):
            n = per_pair
            x1 = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n), np.ones(n)], 1)
            l = x1 @ F[i, j].T                       # line in image j: l . x2 = 0
            nrm = np.sqrt(l[:, 0] ** 2 + l[:, 1] ** 2)
            nrm = np.where(nrm > 0, nrm, 1.0)
            a, b, c = l[:, 0] / nrm, l[:, 1] / nrm, l[:, 2] / nrm
            dist = a * ctr[0] + b * ctr[1] + c
            p0 = ctr[None] - dist[:, None] * np.stack([a, b], 1)
            sft = rng.uniform(-W / 2.0, W / 2.0, n)
            a2 = p0 + sft[:, None] * np.stack([-b, a], 1)
            a1 = x1[:, :2] + rng.normal(0, noise_px, (n, 2))
            a2 = a2 + rng.normal(0, noise_px, (n, 2))
            n_out = int(round(outlier_frac * per_pair))
            if n_out:
                idx = rng.choice(per_pair, n_out, replace=False)
                a2[idx] = rng.uniform(0, [W, H], (n_out, 2))
            kp1.append(a1)
            kp2.append(a2)
            i12.append(np.repeat(np.array([[i, j]], dtype=np.int64), per_pair, 0))
    return {"kp1": np.concatenate(kp1).astype(np.float64), "kp2": np.concatenate(kp2).astype(np.float64),
            "i12": np.concatenate(i12).astype(np.int64), "img_shape": torch.Size((N, 3, H, W))}
