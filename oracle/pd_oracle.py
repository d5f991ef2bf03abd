"""CPU oracle: restatement of the PoseDiffusion sampling hot path (torch-CPU, fp32 or fp64).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Never imported by the product.

Parity status: the reference ships NO tests / golden vectors, so this
oracle is pinned against *the reference's own Python files executed in the build container*
(``oracle/ref_stubs.py`` + ``oracle/make_golden.py`` -> ``tests/golden/*.npz``) and those
fixtures are re-checked on every CPU test run (``tests/test_oracle_golden.py``).  The five
pytorch3d helpers the reference calls (HarmonicEmbedding, quaternion_to_matrix,
PerspectiveCameras, opencv_from_cameras_projection, hat) are NOT in /root/reference and not
installed; they are restated here from the published pytorch3d 0.7.x algorithm (see each
docstring), so for those five "parity unpinned" applies and is recorded in DESIGN.md.

Every function cites the reference file:line it follows (paths relative to
/root/reference/pose_diffusion/).  All functions are dtype-generic: pass fp64 tensors to get
the fp64 oracle used for the end-to-end deviation metric.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# S1  diffusion schedule tables            models/gaussian_diffuser.py:120-187
# --------------------------------------------------------------------------------------

TABLE_NAMES = (
    "betas",
    "alphas_cumprod",
    "alphas_cumprod_prev",
    "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod",
    "posterior_variance",
    "posterior_log_variance_clipped",
    "posterior_mean_coef1",
    "posterior_mean_coef2",
    "p2_loss_weight",
)


def diffusion_tables(timesteps: int = 100, beta_1: float = 1e-4, beta_T: float = 0.1,
                     dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """The 13 schedule buffers for beta_schedule="custom" (gaussian_diffuser.py:136-187).

    Computed in fp64 and cast (gaussian_diffuser.py:157); ``dtype=float64`` keeps fp64.
    """
    betas = torch.linspace(beta_1, beta_T, timesteps, dtype=torch.float64)          # :137
    alphas = 1.0 - betas                                                              # :141
    ac = torch.cumprod(alphas, dim=0)                                                 # :142
    ac_prev = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])                # :143
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)                                   # :171
    t = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),                                        # :164
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),                        # :165
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),                          # :166
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),                            # :167
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),                      # :168
        "posterior_variance": post_var,                                               # :174
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),       # :178
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),             # :179
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),    # :180-182
        "p2_loss_weight": (1 + ac / (1 - ac)) ** -0.0,                                # :185-187
    }
    return {k: v.to(dtype) for k, v in t.items()}


# --------------------------------------------------------------------------------------
# D2  time-step embedding                  util/embedding.py:13-37
# --------------------------------------------------------------------------------------

def timestep_embedding(t: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "time_embed.",
                       dim: int = 256, max_period: int = 10000) -> torch.Tensor:
    """[B] integer steps -> [B,128].  freqs are built in fp32 exactly as embedding.py:24-26."""
    w0, b0 = sd[prefix + "linear.0.weight"], sd[prefix + "linear.0.bias"]
    w2, b2 = sd[prefix + "linear.2.weight"], sd[prefix + "linear.2.bias"]
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]                                           # :31 (fp32)
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1).to(w0.dtype)          # :32
    h = emb @ w0.T + b0
    h = h * torch.sigmoid(h)                                                          # SiLU :19
    return h @ w2.T + b2


# --------------------------------------------------------------------------------------
# D3  harmonic pose embedding              util/embedding.py:40-50 -> pytorch3d HarmonicEmbedding
# --------------------------------------------------------------------------------------

def harmonic_embedding(x: torch.Tensor, n_harmonic_functions: int = 10) -> torch.Tensor:
    """pytorch3d 0.7.x ``HarmonicEmbedding(n, omega_0=1, logspace=True, append_input=True)``.

    Restated from the published algorithm (pytorch3d/renderer/implicit/harmonic_embedding.py,
    0.7.x): ``embed = x[..., None] * 2**k``; ``sin`` of ``[embed, embed + pi/2]`` (cos computed as
    sin(x + pi/2) with the constant held in x's dtype); layout
    ``[sin (dim-major, k-minor) | cos (same) | x]`` -> 9*21 = 189 dims for 9-d poses.
    """
    freqs = (2.0 ** torch.arange(n_harmonic_functions, dtype=torch.float32)).to(x.dtype)
    half_pi = torch.tensor([0.0, 0.5 * torch.pi], dtype=torch.float32).to(x.dtype)
    embed = x[..., None] * freqs                                   # [..., dim, n]
    embed = embed[..., None, :, :] + half_pi[..., None, None]      # [..., 2, dim, n]
    embed = embed.sin().reshape(*x.shape[:-1], -1)
    return torch.cat([embed, x], dim=-1)


# --------------------------------------------------------------------------------------
# D4/D5/D1  transformer trunk + heads      models/denoiser.py:53-76, :79-98, :101-163
# --------------------------------------------------------------------------------------

def _layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)               # biased, as nn.LayerNorm
    return (x - mu) / torch.sqrt(var + eps) * w + b


def encoder_layer(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, nhead: int = 4) -> torch.Tensor:
    """One pre-norm nn.TransformerEncoderLayer in eval mode (denoiser.py:88-97; ReLU, eps 1e-5).

    x: [B, N, d].  ``x += MHA(LN1(x)); x += W2 relu(W1 LN2(x))``; attention over the N frames of
    one sequence, no mask, softmax(q k^T / sqrt(dh)) v.
    """
    B, N, d = x.shape
    dh = d // nhead
    h = _layer_norm(x, sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"])
    qkv = h @ sd[prefix + "self_attn.in_proj_weight"].T + sd[prefix + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(B, N, nhead, dh).transpose(1, 2)
    k = k.reshape(B, N, nhead, dh).transpose(1, 2)
    v = v.reshape(B, N, nhead, dh).transpose(1, 2)
    att = torch.softmax((q / math.sqrt(dh)) @ k.transpose(-1, -2), dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(B, N, d)
    x = x + (ctx @ sd[prefix + "self_attn.out_proj.weight"].T + sd[prefix + "self_attn.out_proj.bias"])
    h = _layer_norm(x, sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"])
    h = torch.relu(h @ sd[prefix + "linear1.weight"].T + sd[prefix + "linear1.bias"])
    return x + (h @ sd[prefix + "linear2.weight"].T + sd[prefix + "linear2.bias"])


def denoiser_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, z: torch.Tensor,
                     num_layers: int = 8, nhead: int = 4) -> torch.Tensor:
    """``Denoiser.forward(x [B,N,9], t [B] int64, z [B,N,384]) -> [B,N,9]`` (denoiser.py:53-76).

    ``sd`` holds the Denoiser's own state_dict keys (``time_embed.linear.0.weight`` ...,
    ``_first``, ``_trunk.layers.{l}...``, ``_last.{0,1,3}``) already in the working dtype.
    """
    B, N, _ = x.shape
    t_emb = timestep_embedding(t, sd)[:, None, :].expand(-1, N, -1)                   # :56-58
    x_emb = harmonic_embedding(x)                                                     # :60
    pivot = torch.zeros_like(z[..., :1])
    pivot[:, 0] = 1.0                                                                 # :62-66
    feed = torch.cat([x_emb, t_emb, z, pivot], dim=-1)                                # :68 (702)
    h = feed @ sd["_first.weight"].T + sd["_first.bias"]                              # :70
    for l in range(num_layers):                                                       # :72
        h = encoder_layer(h, sd, f"_trunk.layers.{l}.", nhead)
    h = h @ sd["_last.0.weight"].T + sd["_last.0.bias"]                               # :74  MLP :140-159
    h = torch.relu(_layer_norm(h, sd["_last.1.weight"], sd["_last.1.bias"]))
    return h @ sd["_last.3.weight"].T + sd["_last.3.bias"]


def cast_state_dict(sd: Dict[str, torch.Tensor], dtype, strip_prefix: str = "") -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if strip_prefix:
            if not k.startswith(strip_prefix):
                continue
            k = k[len(strip_prefix):]
        out[k] = v.detach().to(dtype) if v.is_floating_point() else v.detach()
    return out


# --------------------------------------------------------------------------------------
# S2-S5  one reverse-diffusion step        models/gaussian_diffuser.py:190-282
# --------------------------------------------------------------------------------------

def p_mean_variance(sd, tables, x: torch.Tensor, t: int, z: torch.Tensor, objective: str = "pred_noise"):
    """-> (model_mean, posterior_log_variance_clipped[t], x_start, model output)  (:231-246, :218-229)."""
    tt = torch.full((x.shape[0],), t, dtype=torch.long)                               # :265
    eps = denoiser_forward(sd, x, tt, z)
    if objective == "pred_x0":                                                        # :225-227 (pred_noise is not used by sampling)
        x0 = eps
    else:
        x0 = tables["sqrt_recip_alphas_cumprod"][t] * x - tables["sqrt_recipm1_alphas_cumprod"][t] * eps   # :190-194, :221-223
    mean = tables["posterior_mean_coef1"][t] * x0 + tables["posterior_mean_coef2"][t] * x              # :201-205
    return mean, tables["posterior_log_variance_clipped"][t], x0, eps


def p_sample(sd, tables, x, t: int, z, noise: Optional[torch.Tensor], cond_fn=None, cond_start_step: int = 0,
             objective: str = "pred_noise"):
    """One ``p_sample`` (:248-282).  ``noise`` is the tensor the reference would draw with
    ``randn_like`` (ignored on guided steps and at t == 0, exactly as :270-278)."""
    mean, logvar, x0, _ = p_mean_variance(sd, tables, x, t, z, objective)
    if cond_fn is not None and t < cond_start_step:                                   # :270
        mean = cond_fn(mean, t)
        nz = 0.0                                                                      # :276
    else:
        nz = noise if t > 0 else 0.0                                                  # :278
    return mean + torch.exp(0.5 * logvar) * nz, x0                                    # :280


def p_sample_loop(sd, tables, z: torch.Tensor, init: torch.Tensor, noises: Sequence[Optional[torch.Tensor]],
                  cond_fn=None, cond_start_step: int = 0, num_timesteps: int = 100, objective: str = "pred_noise"):
    """``p_sample_loop`` (:284-300) with the RNG factored out: ``init`` is the ``randn(shape)`` of
    :289 and ``noises[t]`` the ``randn_like`` the reference draws at step t (None where it draws
    nothing).  Returns (pose [B,N,9], process [T+1,B,N,9])."""
    pose = init
    process = [pose]
    for t in reversed(range(num_timesteps)):
        pose, _ = p_sample(sd, tables, pose, t, z, noises[t], cond_fn, cond_start_step, objective)
        process.append(pose)
    return pose, torch.stack(process)


def draw_reference_noise(shape, generator: Optional[torch.Generator], num_timesteps: int = 100,
                         cond_start_step: int = 0, has_cond: bool = False, device="cpu", dtype=torch.float32):
    """Replay the reference's RNG call sequence (:289 then one randn_like per unguided step with
    t > 0, :276-278): returns (init, noises) with noises[t] = None where nothing is drawn."""
    init = torch.randn(shape, generator=generator, device=device, dtype=dtype)
    noises: List[Optional[torch.Tensor]] = [None] * num_timesteps
    for t in reversed(range(num_timesteps)):
        guided = has_cond and t < cond_start_step
        if not guided and t > 0:
            noises[t] = torch.randn(shape, generator=generator, device=device, dtype=dtype)
    return init, noises


# --------------------------------------------------------------------------------------
# P1, X1-X3  pose decode + camera conventions
# --------------------------------------------------------------------------------------

def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.quaternion_to_matrix (real-first, no normalisation, two_s = 2/|q|^2).
    Restated from the published pytorch3d 0.7.x algorithm (transforms/rotation_conversions.py)."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack(
        (
            1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
            two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
            two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j),
        ),
        -1,
    )
    return o.reshape(q.shape[:-1] + (3, 3))


def pose_encoding_to_camera(enc: torch.Tensor, log_focal_length_bias: float = 1.8,
                            min_focal_length: float = 0.1, max_focal_length: float = 20.0):
    """util/camera_transform.py:64-105 for "absT_quaR_logFL": [.., 9] -> dict(R [BN,3,3], T [BN,3],
    focal_length [BN,2]) in PyTorch3D NDC (principal point 0)."""
    e = enc.reshape(-1, enc.shape[-1])                                                # :80
    T = e[:, :3]                                                                      # :85
    R = quaternion_to_matrix(e[:, 3:7])                                               # :86-87
    f = (e[:, 7:9] + log_focal_length_bias).exp()                                     # :89-94
    f = torch.clamp(f, min=min_focal_length, max=max_focal_length)                    # :97
    return {"R": R, "T": T, "focal_length": f}


def opencv_from_cameras_projection(R: torch.Tensor, T: torch.Tensor, focal: torch.Tensor, height: int, width: int):
    """pytorch3d.utils.opencv_from_cameras_projection for PerspectiveCameras in NDC with principal
    point 0 (published 0.7.x algorithm, renderer/camera_conversions.py): flip the first two
    columns of R and entries of T, transpose R, scale = min(h, w)/2, c0 = (w/2, h/2)."""
    Rp = R.clone()
    Tp = T.clone()
    Tp[:, :2] *= -1
    Rp[:, :, :2] *= -1
    Rcv = Rp.permute(0, 2, 1)
    scale = min(height, width) / 2.0
    K = torch.zeros_like(Rcv)
    K[:, 0, 0] = focal[:, 0] * scale
    K[:, 1, 1] = focal[:, 1] * scale
    K[:, 0, 2] = width / 2.0
    K[:, 1, 2] = height / 2.0
    K[:, 2, 2] = 1.0
    return Rcv, Tp, K


def hat(v: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.so3.hat: [[0,-z,y],[z,0,-x],[-y,x,0]]."""
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack((o, -z, y, z, o, -x, -y, x, o), -1).reshape(v.shape[:-1] + (3, 3))


# --------------------------------------------------------------------------------------
# F1  fundamental matrices for all N^2 ordered pairs   util/get_fundamental_matrix.py:14-51
# --------------------------------------------------------------------------------------

def get_fundamental_matrices(R, T, focal, height: int, width: int, i1: torch.Tensor, i2: torch.Tensor):
    Rcv, tcv, K = opencv_from_cameras_projection(R, T, focal, height, width)          # :26-27
    K1, R1, t1, K2, R2, t2 = K[i1], Rcv[i1], tcv[i1], K[i2], Rcv[i2], tcv[i2]          # :29
    R12 = R2 @ R1.permute(0, 2, 1)                                                    # :46
    t12 = t2 - (R12 @ t1[..., None])[..., 0]                                          # :47
    E_t = -(R12.permute(0, 2, 1) @ t12[..., None])[..., 0]                            # :49
    E = R12 @ hat(E_t)                                                                # :50
    return K2.inverse().permute(0, 2, 1) @ E @ K1.inverse()                           # :41  p2^T F p1 = 0


# --------------------------------------------------------------------------------------
# G3  Sampson distance                      util/geometry_guided_sampling.py:129-172
# --------------------------------------------------------------------------------------

def compute_sampson_distance(x: torch.Tensor, pm: Dict, update_R=True, update_T=True, update_FL=True,
                             sampson_max: float = 10):
    cam = pose_encoding_to_camera(x)                                                  # :139
    R, T, f = cam["R"], cam["T"], cam["focal_length"]
    f = f.mean(dim=0).repeat(len(f), 1)                                               # :142
    if not update_R:
        R = R.detach()                                                                # :144-145
    if not update_T:
        T = T.detach()                                                                # :147-148
    if not update_FL:
        f = f.detach()                                                                # :150-151
    F21 = get_fundamental_matrices(R, T, f, pm["h"], pm["w"], pm["i1"], pm["i2"])      # :154
    F = F21.permute(0, 2, 1)                                                          # :155
    k1, k2, pidx = pm["kp1_homo"].to(x.dtype), pm["kp2_homo"].to(x.dtype), pm["pair_idx"]   # :167 (.float())
    Fm = F[pidx]
    left = torch.bmm(k1[:, None], Fm)                                                 # :158
    right = torch.bmm(Fm, k2[..., None])                                              # :159
    bottom = left[:, :, 0].square() + left[:, :, 1].square() + right[:, 0, :].square() + right[:, 1, :].square()
    top = torch.bmm(left, k2[..., None]).square()                                     # :162
    s = (top[:, 0] / bottom)[:, 0]                                                    # :164 (kept 1-d)
    to_print = s.detach().clone().clamp(max=sampson_max).mean()                       # :169
    return s[s < sampson_max], to_print                                               # :170


def prepare_matches(kp1: np.ndarray, kp2: np.ndarray, i12: np.ndarray, img_shape) -> Dict:
    """Host prep of geometry_guided_sampling.py:16-45 (b = img_shape[0] is the frame count)."""
    b, c, h, w = img_shape                                                            # :16
    kp1 = torch.from_numpy(np.asarray(kp1))
    kp2 = torch.from_numpy(np.asarray(kp2))
    i12 = torch.from_numpy(np.asarray(i12))
    pair_idx = (i12[:, 0] * b + i12[:, 1]).long()                                     # :26-27
    pad = lambda a: torch.nn.functional.pad(a, [0, 1], value=1)                       # :29-30
    i1, i2 = [i.reshape(-1) for i in torch.meshgrid(torch.arange(b), torch.arange(b), indexing="ij")]  # :35
    return {"kp1_homo": pad(kp1), "kp2_homo": pad(kp2), "i1": i1, "i2": i2, "h": int(h), "w": int(w),
            "pair_idx": pair_idx}


# --------------------------------------------------------------------------------------
# G2  clipped momentum-SGD on the Sampson loss      util/geometry_guided_sampling.py:67-126
# --------------------------------------------------------------------------------------

def ggs_optimize(x: torch.Tensor, pm: Dict, update_R=True, update_T=True, update_FL=True, alpha=1e-4,
                 learning_rate=1e-2, iter_num=100, sampson_max=10, min_matches=10, trace: Optional[list] = None,
                 **_):
    """Returns (x_new detached, last sampson_to_print, iterations actually stepped).

    Autograd-based exactly like the reference; the SGD(momentum=0.9, dampening=0) step and
    clip_grad_norm_ (coef = max_norm / (||g|| + 1e-6), clamped to 1) are written out explicitly.
    ``trace`` (optional list) receives per-iteration dicts (loss, n_valid, grad, x_after).
    """
    x = x.detach().clone().requires_grad_(True)
    if update_R and update_T and update_FL:
        iter_num = iter_num * 2                                                       # :86-87
    buf = None
    n_frames = x.shape[1]                                                             # :90
    to_print = torch.tensor(float("nan"), dtype=x.dtype)
    steps = 0
    for _ in range(iter_num):                                                         # :92
        valid, to_print = compute_sampson_distance(x, pm, update_R, update_T, update_FL, sampson_max)
        if min_matches > 0 and len(valid) / n_frames < min_matches:                   # :104-108
            break
        loss = valid.mean()                                                           # :110
        (g,) = torch.autograd.grad(loss, x)                                           # :111-112
        mask = (g.abs() > 0)                                                          # :116
        x_norm = (x.detach() * mask).norm()                                           # :117
        max_norm = alpha * x_norm / learning_rate                                     # :119
        coef = torch.clamp(max_norm / (g.norm() + 1e-6), max=1.0)                     # :121 clip_grad_norm_
        g = g * coef
        buf = g.clone() if buf is None else 0.9 * buf + g                             # torch.optim.SGD momentum
        with torch.no_grad():
            x -= learning_rate * buf                                                  # :122
        steps += 1
        if trace is not None:
            trace.append({"loss": loss.detach().clone(), "n_valid": len(valid), "grad": g.detach().clone(),
                          "x": x.detach().clone()})
    return x.detach(), to_print, steps


def geometry_guided_sampling(model_mean: torch.Tensor, t: int, matches_dict: Dict, GGS_cfg: Dict,
                             stats: Optional[list] = None) -> torch.Tensor:
    """The five sequential optimisations of geometry_guided_sampling.py:48-63."""
    pm = prepare_matches(matches_dict["kp1"], matches_dict["kp2"], matches_dict["i12"], matches_dict["img_shape"])
    cfg = {k: v for k, v in GGS_cfg.items() if k in ("alpha", "learning_rate", "iter_num", "sampson_max", "min_matches")}
    flags = [(True, True, True), (False, False, True), (True, False, False), (False, True, False), (True, True, True)]
    x = model_mean
    for (uR, uT, uF) in flags:                                                        # (R, T, FL)
        x, pr, _ = ggs_optimize(x, pm, update_R=uR, update_T=uT, update_FL=uF, **cfg)
        if stats is not None:
            stats.append(float(pr))
    return x


# --------------------------------------------------------------------------------------
# analytic Sampson loss + gradient (fp64 numpy) -- development cross-check for the HIP kernel's
# hand-derived backward; validated against the autograd path above in tests/test_oracle_golden.py
# --------------------------------------------------------------------------------------

def sampson_loss_grad_analytic(x: np.ndarray, kp1: np.ndarray, kp2: np.ndarray, pair_i: np.ndarray,
                               pair_j: np.ndarray, height: int, width: int, update_R=True, update_T=True,
                               update_FL=True, sampson_max: float = 10.0):
    """x [N,9] -> (loss, n_valid, grad [N,9], mean(clamp(s))) with the same math as
    compute_sampson_distance + autograd, derived by hand (this is the derivation the HIP kernel
    implements; DESIGN.md section "GGS backward")."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    T = x[:, 0:3]
    q = x[:, 3:7]
    fl_raw = np.exp(x[:, 7:9] + 1.8)
    fl = np.clip(fl_raw, 0.1, 20.0)
    fl_pass = ((fl_raw >= 0.1) & (fl_raw <= 20.0)).astype(np.float64)
    fbar = fl.mean(axis=0)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    n2 = (q * q).sum(-1)
    ts = 2.0 / n2
    Pm = np.stack([-(j * j + k * k), i * j - k * r, i * k + j * r, i * j + k * r, -(i * i + k * k), j * k - i * r,
                   i * k - j * r, j * k + i * r, -(i * i + j * j)], -1).reshape(N, 3, 3)
    R = np.eye(3)[None] + ts[:, None, None] * Pm
    D = np.array([-1.0, -1.0, 1.0])
    Rc = D[None, :, None] * R.transpose(0, 2, 1)          # Rc[a][b] = D[a] R[b][a]
    tc = D[None] * T
    sc = min(height, width) / 2.0
    cx, cy = width / 2.0, height / 2.0
    A = np.array([[1.0 / (fbar[0] * sc), 0, -cx / (fbar[0] * sc)], [0, 1.0 / (fbar[1] * sc), -cy / (fbar[1] * sc)],
                  [0, 0, 1.0]])
    pairs = sorted(set(zip(pair_i.tolist(), pair_j.tolist())))
    gRc = np.zeros((N, 3, 3))
    gtc = np.zeros((N, 3))
    gA = np.zeros((3, 3))
    x1 = np.concatenate([kp1, np.ones((len(kp1), 1))], 1).astype(np.float64)
    x2 = np.concatenate([kp2, np.ones((len(kp2), 1))], 1).astype(np.float64)
    tot, cnt, tot_clamp = 0.0, 0, 0.0
    per_pair = {}
    for (a, b) in pairs:
        sel = (pair_i == a) & (pair_j == b)
        R12 = Rc[b] @ Rc[a].T
        t12 = tc[b] - R12 @ tc[a]
        Et = -R12.T @ t12
        H = np.array([[0, -Et[2], Et[1]], [Et[2], 0, -Et[0]], [-Et[1], Et[0], 0]])
        E = R12 @ H
        Fo = A.T @ E @ A
        F = Fo.T
        u1, u2 = x1[sel], x2[sel]
        left = u1 @ F                       # [m,3]
        right = u2 @ F.T                    # [m,3]  (F x2)
        e = (left * u2).sum(-1)
        bottom = left[:, 0] ** 2 + left[:, 1] ** 2 + right[:, 0] ** 2 + right[:, 1] ** 2
        s = e * e / bottom
        tot_clamp += np.minimum(s, sampson_max).sum()
        v = s < sampson_max
        tot += s[v].sum()
        cnt += int(v.sum())
        ca = (2 * e / bottom)[v]
        cb = (2 * s / bottom)[v]
        u1v, u2v, lv, rv = u1[v], u2[v], left[v], right[v]
        G = np.einsum("m,mr,mc->rc", ca, u1v, u2v)
        lm = lv.copy(); lm[:, 2] = 0
        rm = rv.copy(); rm[:, 2] = 0
        G -= np.einsum("m,mr,mc->rc", cb, u1v, lm)
        G -= np.einsum("m,mr,mc->rc", cb, rm, u2v)
        per_pair[(a, b)] = (G, R12, t12, Et, H, E)
    for (a, b), (G, R12, t12, Et, H, E) in per_pair.items():
        G = G / max(cnt, 1) if cnt > 0 else G * np.nan
        Gf = G.T                                                   # dL/dFo
        gE = A @ Gf @ A.T
        gA += E @ A @ Gf.T + E.T @ A @ Gf
        gR12 = gE @ H.T
        gH = R12.T @ gE
        gEt = np.array([gH[2, 1] - gH[1, 2], gH[0, 2] - gH[2, 0], gH[1, 0] - gH[0, 1]])
        gt12 = -R12 @ gEt
        gR12 += -np.outer(t12, gEt)
        gtc[b] += gt12
        gtc[a] += -R12.T @ gt12
        gR12 += -np.outer(gt12, tc[a])
        gRc[b] += gR12 @ Rc[a]
        gRc[a] += gR12.T @ Rc[b]
    grad = np.zeros((N, 9))
    if update_T:
        grad[:, 0:3] = D[None] * gtc
    if update_R:
        gR = D[None, None, :] * gRc.transpose(0, 2, 1)              # gR[b][a] = D[a] gRc[a][b]
        gts = (gR * Pm).sum((1, 2))
        g = ts[:, None, None] * gR
        gq = np.zeros((N, 4))
        gq[:, 0] = -k * g[:, 0, 1] + j * g[:, 0, 2] + k * g[:, 1, 0] - i * g[:, 1, 2] - j * g[:, 2, 0] + i * g[:, 2, 1]
        gq[:, 1] = (j * g[:, 0, 1] + k * g[:, 0, 2] + j * g[:, 1, 0] - 2 * i * g[:, 1, 1] - r * g[:, 1, 2]
                    + k * g[:, 2, 0] + r * g[:, 2, 1] - 2 * i * g[:, 2, 2])
        gq[:, 2] = (-2 * j * g[:, 0, 0] + i * g[:, 0, 1] + r * g[:, 0, 2] + i * g[:, 1, 0] + k * g[:, 1, 2]
                    - r * g[:, 2, 0] + k * g[:, 2, 1] - 2 * j * g[:, 2, 2])
        gq[:, 3] = (-2 * k * g[:, 0, 0] - r * g[:, 0, 1] + i * g[:, 0, 2] + r * g[:, 1, 0] - 2 * k * g[:, 1, 1]
                    + j * g[:, 1, 2] + i * g[:, 2, 0] + j * g[:, 2, 1])
        gq += (gts * (-4.0 / (n2 * n2)))[:, None] * q
        grad[:, 3:7] = gq
    if update_FL:
        gfx = gA[0, 0] * (-1.0 / (fbar[0] ** 2 * sc)) + gA[0, 2] * (cx / (fbar[0] ** 2 * sc))
        gfy = gA[1, 1] * (-1.0 / (fbar[1] ** 2 * sc)) + gA[1, 2] * (cy / (fbar[1] ** 2 * sc))
        grad[:, 7] = gfx / N * fl[:, 0] * fl_pass[:, 0]
        grad[:, 8] = gfy / N * fl[:, 1] * fl_pass[:, 1]
    loss = tot / cnt if cnt > 0 else float("nan")
    return loss, cnt, grad, tot_clamp / len(kp1)


# --------------------------------------------------------------------------------------
# N3  evaluation metrics              util/metric.py, demo.py:120-133, test.py:113-121
# --------------------------------------------------------------------------------------

def world_to_view_matrix(R: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """pytorch3d ``cameras.get_world_to_view_transform().get_matrix()``: 4x4, row-vector convention
    X_view = X_world R + T  ->  [[R, 0], [T, 1]]   (restated; pytorch3d absent)."""
    n = R.shape[0]
    M = torch.zeros(n, 4, 4, dtype=R.dtype)
    M[:, :3, :3] = R
    M[:, 3, :3] = T
    M[:, 3, 3] = 1.0
    return M


def _acos_linear_extrapolation(x: torch.Tensor, bound: float) -> torch.Tensor:
    """pytorch3d.transforms.math.acos_linear_extrapolation with bounds (-bound, bound): acos inside, first-order
    Taylor expansion of acos around the bound outside (restated from the published 0.7.x algorithm)."""
    import math
    out = torch.empty_like(x)
    up, lo = x >= bound, x <= -bound
    mid = ~up & ~lo
    out[mid] = torch.acos(x[mid])
    for mask, x0 in ((up, bound), (lo, -bound)):
        out[mask] = (x[mask] - x0) * (-1.0 / math.sqrt(1.0 - x0 * x0)) + math.acos(x0)
    return out


def so3_relative_angle(R1: torch.Tensor, R2: torch.Tensor, cos_bound: float = 1e-4, eps: float = 1e-4) -> torch.Tensor:
    """pytorch3d.transforms.so3_relative_angle(R1, R2, eps=1e-4) as metric.py:147 calls it: angle of R1 R2^T from
    its trace, acos linearly extrapolated beyond |cos| > 1 - cos_bound (restated; raises like pytorch3d when the
    trace is outside [-1 - eps, 3 + eps])."""
    R12 = R1 @ R2.transpose(1, 2)
    tr = R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]
    if ((tr < -1.0 - eps) | (tr > 3.0 + eps)).any():
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    return _acos_linear_extrapolation((tr - 1.0) * 0.5, 1.0 - cos_bound)


def camera_to_rel_deg(R_pred, T_pred, R_gt, T_gt, batch_size: int):
    """util/metric.py:14-47 on (R, T) arrays: pairwise relative poses of all i < j within each sequence, rotation angle
    (:143-151) and translation direction angle (:154-172) between ground truth and prediction, in degrees."""
    gt, pr = world_to_view_matrix(R_gt, T_gt), world_to_view_matrix(R_pred, T_pred)
    n = gt.shape[0] // batch_size
    i1_, i2_ = torch.combinations(torch.arange(n), 2, with_replacement=False).unbind(-1)               # :106-111
    i1, i2 = [(i[None] + torch.arange(batch_size)[:, None] * n).reshape(-1) for i in (i1_, i2_)]

    def inv(se3):                                                                                       # :114-140
        Rt = se3[:, :3, :3].transpose(1, 2)
        out = se3.clone()
        out[:, :3, :3] = Rt
        out[:, 3:, :3] = -se3[:, 3:, :3] @ Rt
        return out

    rel_gt, rel_pr = inv(gt[i1]) @ gt[i2], inv(pr[i1]) @ pr[i2]
    r = so3_relative_angle(rel_gt[:, :3, :3], rel_pr[:, :3, :3]) * 180.0 / np.pi
    t_gt, t = rel_gt[:, 3, :3], rel_pr[:, 3, :3]
    eps = 1e-15
    t = t / (t.norm(dim=1, keepdim=True) + eps)
    t_gt = t_gt / (t_gt.norm(dim=1, keepdim=True) + eps)
    loss_t = torch.clamp_min(1.0 - (t * t_gt).sum(1) ** 2, eps)
    err_t = torch.acos(torch.sqrt(1 - loss_t))
    err_t[torch.isnan(err_t) | torch.isinf(err_t)] = 1e6
    return r, err_t * 180.0 / np.pi


def calculate_auc_np(r_error: np.ndarray, t_error: np.ndarray, max_threshold: int = 30) -> float:
    """util/metric.py:50-78."""
    max_errors = np.max(np.concatenate((r_error[:, None], t_error[:, None]), axis=1), axis=1)
    histogram, _ = np.histogram(max_errors, bins=np.arange(max_threshold + 1))
    return float(np.mean(np.cumsum(histogram.astype(float) / float(len(max_errors)))))


def compute_ARE(R1: np.ndarray, R2: np.ndarray) -> np.ndarray:
    """util/metric.py:174-185: absolute rotation error in degrees."""
    R_rel = np.einsum("Bij,Bjk ->Bik", R1.transpose(0, 2, 1), R2)
    t = (np.trace(R_rel, axis1=1, axis2=2) - 1) / 2
    return np.arccos(np.clip(t, -1, 1)) * 180 / np.pi


def corresponding_cameras_alignment(R_src, T_src, R_tgt, T_tgt, estimate_scale: bool = True, eps: float = 1e-9):
    """pytorch3d.ops.corresponding_cameras_alignment(mode="extrinsics") as demo.py:127-129 calls it (restated from the
    published 0.7.x algorithm; it is the least-squares similarity of the derivation in DESIGN section 3.4):
    R_A = V U^T from the SVD of mean_i R_src_i R_tgt_i^T; with A_i = T_src_i R_src_i^T and B_i = T_tgt_i R_src_i^T,
    s = <Ac, Bc> / <Ac, Ac>, T_A = mean B - s mean A;  aligned R_i = R_A R_src_i, T_i = T_A R_src_i + s T_src_i."""
    RRcov = (R_src @ R_tgt.transpose(1, 2)).mean(0)
    U, _, Vh = torch.linalg.svd(RRcov)
    RA = Vh.transpose(0, 1) @ U.transpose(0, 1)
    A = (R_src @ T_src[:, :, None])[:, :, 0]
    B = (R_src @ T_tgt[:, :, None])[:, :, 0]
    Amu, Bmu = A.mean(0, keepdim=True), B.mean(0, keepdim=True)
    if estimate_scale and A.shape[0] > 1:
        Ac, Bc = A - Amu, B - Bmu
        s = (Ac * Bc).mean() / (Ac ** 2).mean().clamp(eps)
    else:
        s = torch.tensor(1.0, dtype=R_src.dtype)
    TA = Bmu - s * Amu
    R_al = RA[None] @ R_src
    T_al = (TA[None].expand(R_src.shape[0], 1, 3) @ R_src)[:, 0] + T_src * s
    return R_al, T_al, (RA, TA[0], s)
