"""pose_diffusion/util/embedding.py of the reference on the HIP engine (same classes, constructor arguments, state_dict keys
and `forward` results).

Inside `Denoiser.forward` / `GaussianDiffusion.sample` both embeddings are fused into the denoiser kernels (time embedding: a
[T,128] table built at engine creation; harmonic pose embedding: formed while `_first`'s rows are staged), so the sampling path
never calls these `forward`s.  Called piecewise -- as a user of the reference's modules may -- they run the same device code
through two stateless C-ABI entry points (`pd_time_embedding`, `pd_pose_embedding`, include/pd_engine.h).  GPU only, like
everything else here: there is no CPU fallback."""
import torch
import torch.nn as nn

from posediffusion_amd import _lib


def _on_gpu(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.device.type != "cuda":
        raise RuntimeError(f"{what} runs only on an AMD GPU (posediffusion_amd has no CPU path); got a tensor on {t.device}")
    return t


class TimeStepEmbedding(nn.Module):
    """embedding.py:13-37: sinusoidal features of the timestep -> Linear(256,128) -> SiLU -> Linear(128,128)."""

    def __init__(self, dim=256, max_period=10000):
        super().__init__()
        self.dim, self.max_period = dim, max_period
        self.linear = nn.Sequential(nn.Linear(dim, dim // 2), nn.SiLU(), nn.Linear(dim // 2, dim // 2))
        self.out_dim = dim // 2

    @torch.no_grad()
    def forward(self, timesteps):
        if self.dim != 256 or self.max_period != 10000:
            raise ValueError("the HIP engine is built for TimeStepEmbedding(dim=256, max_period=10000) (models/denoiser.py:44)")
        t = _on_gpu(timesteps, "TimeStepEmbedding.forward").reshape(-1).to(torch.float32).contiguous()     # embedding.py:31 `.float()`
        w0, b0, w2, b2 = (_on_gpu(p, "TimeStepEmbedding.forward").detach().to(torch.float32).contiguous()
                          for p in (self.linear[0].weight, self.linear[0].bias, self.linear[2].weight, self.linear[2].bias))
        out = torch.empty(t.shape[0], self.out_dim, device=t.device, dtype=torch.float32)
        _lib.check(_lib.load().pd_time_embedding(w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(), t.data_ptr(), t.shape[0],
                                                 out.data_ptr(), torch.cuda.current_stream(t.device).cuda_stream), "pd_time_embedding")
        return out


class PoseEmbedding(nn.Module):
    """embedding.py:40-54: pytorch3d HarmonicEmbedding(n_harmonic_functions, append_input) of the pose encoding."""

    def __init__(self, target_dim, n_harmonic_functions=10, append_input=True):
        super().__init__()
        if n_harmonic_functions != 10 or not append_input:
            raise ValueError("the HIP engine is built for HarmonicEmbedding(n=10, append_input=True)")
        self.target_dim = target_dim
        self.out_dim = target_dim * (2 * n_harmonic_functions + 1)

    @torch.no_grad()
    def forward(self, pose_encoding):
        x = _on_gpu(pose_encoding, "PoseEmbedding.forward").to(torch.float32).contiguous()
        dim = x.shape[-1]
        rows = x.numel() // max(dim, 1)
        out = torch.empty(*x.shape[:-1], 21 * dim, device=x.device, dtype=torch.float32)
        _lib.check(_lib.load().pd_pose_embedding(x.data_ptr(), rows, dim, out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                   "pd_pose_embedding")
        return out
