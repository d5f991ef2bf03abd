"""DDPM schedule buffers (S1: models/gaussian_diffuser.py:120-187), host side.

Computed in fp64 and stored fp32 exactly like the reference's ``register_buffer`` helper
(gaussian_diffuser.py:157).  Init-time only; not part of the per-step hot loop.
"""
import math

import torch

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2", "p2_loss_weight",
)


def make_betas(beta_schedule: str, timesteps: int, beta_1: float, beta_T: float) -> torch.Tensor:
    if beta_schedule == "custom":                       # gaussian_diffuser.py:136-137
        return torch.linspace(beta_1, beta_T, timesteps, dtype=torch.float64)
    if beta_schedule == "linear":                       # :55-59
        scale = 1000 / timesteps
        return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)
    if beta_schedule == "cosine":                       # :62-72
        s = 0.008
        x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
        ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    raise ValueError(f"unknown beta schedule {beta_schedule}")   # :138-139


def diffusion_buffers(beta_schedule="custom", timesteps=100, beta_1=1e-4, beta_T=0.1, p2_gamma=0.0, p2_k=1):
    b = make_betas(beta_schedule, timesteps, beta_1, beta_T)
    a = 1.0 - b
    ac = torch.cumprod(a, dim=0)
    acp = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
    pv = b * (1.0 - acp) / (1.0 - ac)
    out = {
        "betas": b, "alphas_cumprod": ac, "alphas_cumprod_prev": acp,
        "sqrt_alphas_cumprod": ac.sqrt(), "sqrt_one_minus_alphas_cumprod": (1.0 - ac).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - ac).log(), "sqrt_recip_alphas_cumprod": (1.0 / ac).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / ac - 1).sqrt(), "posterior_variance": pv,
        "posterior_log_variance_clipped": pv.clamp(min=1e-20).log(),
        "posterior_mean_coef1": b * acp.sqrt() / (1.0 - ac), "posterior_mean_coef2": (1.0 - acp) * a.sqrt() / (1.0 - ac),
        "p2_loss_weight": (p2_k + ac / (1 - ac)) ** -p2_gamma,
    }
    return {k: v.to(torch.float32) for k, v in out.items()}
