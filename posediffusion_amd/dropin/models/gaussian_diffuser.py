"""Drop-in `GaussianDiffusion` (pose_diffusion/models/gaussian_diffuser.py:75-306, sampling half).

Same constructor, same 13 persistent buffers (checkpoint keys `diffuser.betas` ...), same
``sample(shape, z, cond_fn=None, cond_start_step=0) -> (pose [B,N,9], process [T+1,B,N,9])``.
The loop itself -- 100 denoiser evaluations, posterior updates and (when cond_fn is the shipped
GGS partial) the 7000 guided iterations -- runs as one hipGraph replay of hand-written kernels.
Training (`forward`/`p_losses`, :308-332) is out of scope and raises."""
import os
from collections import namedtuple

import torch
from torch import nn

from posediffusion_amd import host
from posediffusion_amd.schedule import BUFFER_NAMES, diffusion_buffers


ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])      # gaussian_diffuser.py:34


def _at(table, t, like):
    """table[t] broadcast over the trailing axes of ``like`` (the reference's ``extract``, :49-52); t: int or LongTensor [B]."""
    t = torch.as_tensor(t, device=table.device, dtype=torch.long).reshape(-1)
    return table[t].reshape((-1,) + (1,) * (like.dim() - 1))


class GaussianDiffusion(nn.Module):
    def __init__(self, timesteps=100, sampling_timesteps=None, beta_1=0.0001, beta_T=0.1, loss_type="l1",
                 objective="pred_noise", beta_schedule="custom", p2_loss_weight_gamma=0.0, p2_loss_weight_k=1):
        super().__init__()
        if objective not in {"pred_noise", "pred_x0"}:
            raise AssertionError("objective must be either pred_noise (predict noise) or pred_x0 (predict image start)")
        self.objective, self.loss_type, self.beta_schedule = objective, loss_type, beta_schedule
        self.timesteps, self.beta_1, self.beta_T = timesteps, beta_1, beta_T
        bufs = diffusion_buffers(beta_schedule, timesteps, beta_1, beta_T, p2_loss_weight_gamma, p2_loss_weight_k)
        for name in BUFFER_NAMES:
            self.register_buffer(name, bufs[name])
        self.num_timesteps = int(timesteps)
        self.sampling_timesteps = timesteps if sampling_timesteps is None else sampling_timesteps
        assert self.sampling_timesteps <= timesteps
        self.model = None          # the Denoiser, assigned after construction (pose_diffusion_model.py:61)
        self.use_graph = os.environ.get("PD_USE_GRAPH", "1") != "0"
        self.last_ggs_stats = None

    # ---- schedule helpers (:190-216): elementwise on the buffers, same names and argument order; the sampler itself has these
    # fused into pd_tail_kernel and does not call them
    def predict_start_from_noise(self, x_t, t, noise):
        return _at(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - _at(self.sqrt_recipm1_alphas_cumprod, t, x_t) * noise

    def predict_noise_from_start(self, x_t, t, x0):
        return (_at(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t - x0) / _at(self.sqrt_recipm1_alphas_cumprod, t, x_t)

    def q_posterior(self, x_start, x_t, t):
        mean = _at(self.posterior_mean_coef1, t, x_t) * x_start + _at(self.posterior_mean_coef2, t, x_t) * x_t
        return mean, _at(self.posterior_variance, t, x_t), _at(self.posterior_log_variance_clipped, t, x_t)

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return _at(self.sqrt_alphas_cumprod, t, x_start) * x_start + _at(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise

    def model_predictions(self, x, t, z, x_self_cond=None):
        """(:218-229) one denoiser evaluation on the engine; the pair (pred_noise, pred_x_start) by the objective."""
        out = self.model(x, t, z)
        if self.objective == "pred_noise":
            return ModelPrediction(out, self.predict_start_from_noise(x, t, out))
        return ModelPrediction(self.predict_noise_from_start(x, t, out), out)

    @property
    def loss_fn(self):                                              # :334-341
        if self.loss_type == "l1":
            return nn.functional.l1_loss
        if self.loss_type == "l2":
            return nn.functional.mse_loss
        raise ValueError(f"invalid loss type {self.loss_type}")

    # ---- step-level pieces (same names as the reference) ------------------------------------
    def p_mean_variance(self, x, t, z, x_self_cond=None, clip_denoised=False):
        if clip_denoised:
            raise NotImplementedError("We don't clip the output because pose does not have a clear bound.")
        B, N, _ = x.shape
        eng = host.get_engine(self.model, self, B, N)
        tt = int(torch.as_tensor(t).reshape(-1)[0])
        mean, x0 = eng.p_mean(x, z, tt)
        shape = (B,) + (1,) * (x.dim() - 1)
        return (mean, self.posterior_variance[tt].expand(shape), self.posterior_log_variance_clipped[tt].expand(shape), x0)

    @torch.no_grad()
    def p_sample(self, x, t: int, z, x_self_cond=None, clip_denoised=False, cond_fn=None, cond_start_step=0):
        B, N, _ = x.shape
        eng = host.get_engine(self.model, self, B, N)
        mean, x0 = eng.p_mean(x, z, int(t))
        if cond_fn is not None and t < cond_start_step:           # gaussian_diffuser.py:270-276
            mean = cond_fn(mean, t)
            noise = None
        else:
            noise = torch.randn_like(x) if t > 0 else None        # :278
        return eng.p_finish(mean, noise, int(t)), x0

    @torch.no_grad()
    def p_sample_loop(self, shape, z, cond_fn=None, cond_start_step=0):
        B, N, _ = shape
        device = self.betas.device
        parsed = host.parse_ggs_cond_fn(cond_fn) if cond_fn is not None else None
        # demo.py:79-92: hloc returning no matches (kp1 is None) means sampling without GGS
        has_ggs = False
        if parsed is not None and cond_start_step > 0:
            mds = parsed[0] if isinstance(parsed[0], (list, tuple)) else [parsed[0]]
            have = [host.has_matches(m) for m in mds]
            if any(have) and not all(have):
                # guidance (and with it the noise schedule, gaussian_diffuser.py:270-278) is a property of the whole call: silently
                # dropping it for every sequence because one has no matches would change the results of the others
                raise ValueError(f"sequences {[b for b, h in enumerate(have) if not h]} of the batch have no matches: sample them in a call "
                                 "without cond_fn (demo.py:79-92) and the others with it")
            has_ggs = all(have)
        eng = host.get_engine(self.model, self, B, N)
        if cond_fn is not None and parsed is None:
            # unknown guidance callable: reference control flow in Python, arithmetic on the HIP kernels
            pose = torch.randn(shape, device=device)
            process = [pose.unsqueeze(0)]
            for t in reversed(range(self.num_timesteps)):
                pose, _ = self.p_sample(pose, t, z, cond_fn=cond_fn, cond_start_step=cond_start_step)
                process.append(pose.unsqueeze(0))
            return pose, torch.cat(process)
        noise = host.draw_noise(tuple(shape), self.num_timesteps, device, cond_start_step, has_ggs)
        cfg = None
        if has_ggs:
            matches, cfg = parsed
            host.upload_matches(eng, matches, B)
        pose, process, stats = eng.sample(z, noise, cond_start_step if has_ggs else 0, cfg, use_graph=self.use_graph)
        self.last_ggs_stats = stats
        if has_ggs:
            # the GGS workgroups of a sequence exchange sums through bounded spins; one that gave up (co-residency lost
            # to another process) raises here instead of returning garbage poses, and the flag is cleared for the next call
            eng.check_async()
        # (the reference prints these lines unconditionally, geometry_guided_sampling.py:124; PD_GGS_VERBOSE=0 mutes them)
        if stats is not None and os.environ.get("PD_GGS_VERBOSE", "1") not in ("", "0"):
            st = stats.cpu()
            for k in range(st.shape[0]):
                host.print_ggs_stats(st[k], cond_start_step - 1 - k, int(dict(cfg).get("iter_num", 100)))   # incl. the drop line, :104-108
        return pose, process

    @torch.no_grad()
    def sample(self, shape, z, cond_fn=None, cond_start_step=0):
        return self.p_sample_loop(shape, z=z, cond_fn=cond_fn, cond_start_step=cond_start_step)

    # ---- training half: out of scope ---------------------------------------------------------
    def p_losses(self, *a, **k):
        raise NotImplementedError("training is out of scope of the MI355X sampling engine (SURVEY.md section 2.1)")

    def forward(self, *a, **k):
        raise NotImplementedError("training is out of scope of the MI355X sampling engine (SURVEY.md section 2.1)")
