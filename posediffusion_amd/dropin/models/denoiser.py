"""Drop-in `Denoiser` / `TransformerEncoderWrapper` (pose_diffusion/models/denoiser.py:22-98).

Same constructor arguments, same parameter names (checkpoints load with strict=True), same
``forward(x [B,N,9], t [B], z [B,N,384]) -> [B,N,9]`` -- but forward runs the hand-written HIP
kernels (posediffusion_amd/csrc/pd_denoiser.hip) instead of ~970 ATen launches."""
from typing import Dict

import torch
import torch.nn as nn

from posediffusion_amd.compat import instantiate
from util.embedding import PoseEmbedding, TimeStepEmbedding


def TransformerEncoderWrapper(d_model: int, nhead: int, num_encoder_layers: int, dim_feedforward: int = 2048,
                              dropout: float = 0.1, norm_first: bool = True, batch_first: bool = True):
    """Weight container with nn.TransformerEncoder's parameter names (denoiser.py:79-98)."""
    if not (norm_first and batch_first):
        raise ValueError("the HIP engine implements the pre-norm, batch-first encoder of cfgs/default.yaml")
    layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                                       batch_first=batch_first, norm_first=norm_first)
    return nn.TransformerEncoder(layer, num_encoder_layers)


class Denoiser(nn.Module):
    def __init__(self, TRANSFORMER: Dict, target_dim: int = 9, pivot_cam_onehot: bool = True, z_dim: int = 384,
                 mlp_hidden_dim: int = 128):
        super().__init__()
        if target_dim != 9 or not pivot_cam_onehot:
            raise ValueError("the HIP engine is built for target_dim=9 with the pivot one-hot")
        self.pivot_cam_onehot, self.target_dim = pivot_cam_onehot, target_dim
        self.time_embed = TimeStepEmbedding()
        self.pose_embed = PoseEmbedding(target_dim=target_dim)
        first_dim = self.time_embed.out_dim + self.pose_embed.out_dim + z_dim + int(pivot_cam_onehot)
        d_model = TRANSFORMER["d_model"]
        self._first = nn.Linear(first_dim, d_model)
        self._trunk = instantiate(TRANSFORMER, _recursive_=False)
        self._last = nn.Sequential(nn.Linear(d_model, mlp_hidden_dim), nn.LayerNorm(mlp_hidden_dim), nn.ReLU(inplace=True),
                                   nn.Linear(mlp_hidden_dim, target_dim))
        # older pytorch3d checkpoints carry the (non-learned) harmonic frequencies as a buffer
        self._register_load_state_dict_pre_hook(self._drop_harmonic_buffers)

    @staticmethod
    def _drop_harmonic_buffers(state_dict, prefix, *args):
        for k in [k for k in state_dict if k.startswith(prefix + "pose_embed._emb_pose.")]:
            state_dict.pop(k)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, t: torch.Tensor, z: torch.Tensor):
        from posediffusion_amd.host import get_engine
        B, N, _ = x.shape
        eng = get_engine(self, None, B, N)
        t = torch.as_tensor(t).reshape(-1)
        steps = t.unique().tolist()
        if len(steps) == 1:
            return eng.denoise(x, z, int(steps[0]))
        out = torch.empty_like(x, dtype=torch.float32)
        for s in steps:                      # per-sequence timesteps: one launch group per distinct t
            sel = (t == s).nonzero().flatten()
            out[sel] = eng.denoise(x[sel], z[sel], int(s))
        return out
