"""Parameter containers mirroring pose_diffusion/util/embedding.py (same state_dict keys).

The arithmetic of both embeddings is fused into the HIP engine (time embedding: a [T,128] table
built at engine creation; harmonic pose embedding: computed while staging the first GEMM's rows),
so these modules only own weights and report dimensions."""
import torch.nn as nn


class TimeStepEmbedding(nn.Module):
    def __init__(self, dim=256, max_period=10000):
        super().__init__()
        self.dim, self.max_period = dim, max_period
        self.linear = nn.Sequential(nn.Linear(dim, dim // 2), nn.SiLU(), nn.Linear(dim // 2, dim // 2))
        self.out_dim = dim // 2

    def forward(self, timesteps):
        raise NotImplementedError("fused into the HIP engine (pd_denoiser.hip: pd_time_table_kernel); "
                                  "call Denoiser.forward / GaussianDiffusion.sample")


class PoseEmbedding(nn.Module):
    def __init__(self, target_dim, n_harmonic_functions=10, append_input=True):
        super().__init__()
        if n_harmonic_functions != 10 or not append_input:
            raise ValueError("the HIP engine is built for HarmonicEmbedding(n=10, append_input=True)")
        self.out_dim = target_dim * (2 * n_harmonic_functions + 1)

    def forward(self, pose_encoding):
        raise NotImplementedError("fused into the HIP engine (pd_denoiser.hip: pd_gemm_kernel<704,2,0>)")
