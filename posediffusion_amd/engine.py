"""PoseEngine: thin Python owner of a ``pd_engine`` (include/pd_engine.h).

PyTorch is plumbing only (device memory, the current HIP stream, RNG): every arithmetic step
of the sampling path runs in the hand-written HIP kernels behind the C-ABI.  There is no CPU
path -- constructing an engine without a GPU or without the built library raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

GGS_DEFAULTS = dict(alpha=1e-4, learning_rate=1e-2, iter_num=100, sampson_max=10.0, min_matches=10,
                    momentum=0.9, wgs_per_seq=0, reserved=0)   # cfgs/default.yaml:6-13 (+ SGD momentum of :89; engine knobs)

_TABLES = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
           "posterior_mean_coef2", "posterior_log_variance_clipped")


def make_ggs_cfg(cfg: Optional[Dict] = None, **over) -> _lib.pd_ggs_cfg:
    d = dict(GGS_DEFAULTS)
    for src in (cfg or {}), over:
        for k, v in src.items():
            if k in d:
                d[k] = v
    return _lib.pd_ggs_cfg(float(d["alpha"]), float(d["learning_rate"]), int(d["iter_num"]), float(d["sampson_max"]),
                           int(d["min_matches"]), float(d["momentum"]), int(d["wgs_per_seq"]), int(d.get("reserved", 0)))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class PoseEngine:
    """One engine per device.  ``denoiser_sd`` uses the reference Denoiser's state_dict keys
    (``time_embed.linear.0.weight`` ... ``_last.3.bias``); ``tables`` the GaussianDiffusion buffers."""

    def __init__(self, denoiser_sd: Dict[str, torch.Tensor], tables: Dict[str, torch.Tensor], device=None,
                 max_B: int = 8, max_N: int = 20, num_layers: int = 8, nhead: int = 4, objective: str = "pred_noise"):
        if objective not in ("pred_noise", "pred_x0"):                     # models/gaussian_diffuser.py:105-108
            raise AssertionError("objective must be either pred_noise (predict noise) or pred_x0 (predict image start)")
        if not torch.cuda.is_available():
            raise RuntimeError("posediffusion_amd.PoseEngine needs an AMD GPU (torch.cuda unavailable); "
                               "there is no CPU fallback for the sampling path")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise RuntimeError(f"PoseEngine device must be a GPU, got {self.device}")
        self.max_B, self.max_N = int(max_B), int(max_N)
        self._h = C.c_void_p(None)
        keep = []

        def dev(t: torch.Tensor) -> int:
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        sd = denoiser_sd
        w = _lib.pd_weights()
        d_model = sd["_first.weight"].shape[0]
        w.d_model, w.nhead, w.num_layers = d_model, nhead, num_layers
        w.dim_ff = sd["_trunk.layers.0.linear1.weight"].shape[0]
        w.mlp_hidden = sd["_last.0.weight"].shape[0]
        w.t_emb_dim = sd["time_embed.linear.0.weight"].shape[1]
        w.n_harmonic = 10
        w.z_dim = sd["_first.weight"].shape[1] - (9 * 21 + w.t_emb_dim // 2 + 1)
        w.timesteps = int(tables[_TABLES[0]].shape[0])
        w.reserved = _lib.PD_WEIGHTS_PRED_X0 if objective == "pred_x0" else 0
        self.objective = objective
        w.time_w0, w.time_b0 = dev(sd["time_embed.linear.0.weight"]), dev(sd["time_embed.linear.0.bias"])
        w.time_w2, w.time_b2 = dev(sd["time_embed.linear.2.weight"]), dev(sd["time_embed.linear.2.bias"])
        w.first_w, w.first_b = dev(sd["_first.weight"]), dev(sd["_first.bias"])
        for l in range(num_layers):
            p = f"_trunk.layers.{l}."
            L = w.layers[l]
            L.norm1_w, L.norm1_b = dev(sd[p + "norm1.weight"]), dev(sd[p + "norm1.bias"])
            L.in_proj_w, L.in_proj_b = dev(sd[p + "self_attn.in_proj_weight"]), dev(sd[p + "self_attn.in_proj_bias"])
            L.out_proj_w, L.out_proj_b = dev(sd[p + "self_attn.out_proj.weight"]), dev(sd[p + "self_attn.out_proj.bias"])
            L.norm2_w, L.norm2_b = dev(sd[p + "norm2.weight"]), dev(sd[p + "norm2.bias"])
            L.linear1_w, L.linear1_b = dev(sd[p + "linear1.weight"]), dev(sd[p + "linear1.bias"])
            L.linear2_w, L.linear2_b = dev(sd[p + "linear2.weight"]), dev(sd[p + "linear2.bias"])
        w.last0_w, w.last0_b = dev(sd["_last.0.weight"]), dev(sd["_last.0.bias"])
        w.last_ln_w, w.last_ln_b = dev(sd["_last.1.weight"]), dev(sd["_last.1.bias"])
        w.last3_w, w.last3_b = dev(sd["_last.3.weight"]), dev(sd["_last.3.bias"])
        for name in _TABLES:
            setattr(w, name, dev(tables[name]))
        self.timesteps = int(w.timesteps)
        self.z_dim = int(w.z_dim)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _lib.check(self.lib.pd_engine_create(C.byref(w), self.max_B, self.max_N, C.byref(self._h)), "pd_engine_create")
        del keep

    # ---------------------------------------------------------------- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.pd_engine_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def version(self) -> str:
        return self.lib.pd_version().decode()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _f32(self, t: torch.Tensor, shape=None) -> torch.Tensor:
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def check_async(self):
        _lib.check(self.lib.pd_check_async_error(self._h), "pd_check_async_error")

    # ---------------------------------------------------------------- denoiser / DDPM
    def denoise(self, x: torch.Tensor, z: torch.Tensor, t: int) -> torch.Tensor:
        """Denoiser.forward (models/denoiser.py:53-76) for one shared timestep t."""
        B, N, _ = x.shape
        x, z = self._f32(x, (B, N, 9)), self._f32(z, (B, N, self.z_dim))
        out = torch.empty_like(x)
        _lib.check(self.lib.pd_denoise_step(self._h, x.data_ptr(), z.data_ptr(), int(t), B, N, out.data_ptr(),
                                            self._stream()), "pd_denoise_step")
        return out

    def p_mean(self, x: torch.Tensor, z: torch.Tensor, t: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(model_mean, x_start) of p_mean_variance (models/gaussian_diffuser.py:231-246)."""
        B, N, _ = x.shape
        x, z = self._f32(x, (B, N, 9)), self._f32(z, (B, N, self.z_dim))
        mean, x0 = torch.empty_like(x), torch.empty_like(x)
        _lib.check(self.lib.pd_p_mean(self._h, x.data_ptr(), z.data_ptr(), int(t), B, N, mean.data_ptr(), x0.data_ptr(),
                                      self._stream()), "pd_p_mean")
        return mean, x0

    def p_finish(self, mean: torch.Tensor, noise: Optional[torch.Tensor], t: int) -> torch.Tensor:
        B, N, _ = mean.shape
        mean = self._f32(mean)
        noise = None if noise is None else self._f32(noise, mean.shape)
        out = torch.empty_like(mean)
        _lib.check(self.lib.pd_p_finish(self._h, mean.data_ptr(), _ptr(noise), int(t), B, N, out.data_ptr(),
                                        self._stream()), "pd_p_finish")
        return out

    # ---------------------------------------------------------------- GGS
    def set_matches(self, seq: int, kp1: np.ndarray, kp2: np.ndarray, i12: np.ndarray, img_shape: Sequence[int]):
        """Upload matches exactly as demo.py:82-84 holds them (kp float64 [M,2], i12 int64 [M,2],
        img_shape = (N, 3, H, W))."""
        kp1 = np.ascontiguousarray(kp1, dtype=np.float64)
        kp2 = np.ascontiguousarray(kp2, dtype=np.float64)
        i12 = np.ascontiguousarray(i12, dtype=np.int64)
        if kp1.ndim != 2 or kp1.shape[1] != 2 or kp1.shape != kp2.shape or i12.shape != kp1.shape:
            raise ValueError("kp1/kp2/i12 must all be [M, 2]")
        n, _, h, w = (int(v) for v in img_shape)
        self.__dict__.setdefault("_match_ids", {}).pop(int(seq), None)   # host.upload_matches' identity cache
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pd_ggs_set_matches(self._h, int(seq), kp1.ctypes.data, kp2.ctypes.data, i12.ctypes.data,
                                                   kp1.shape[0], n, h, w), "pd_ggs_set_matches")

    def set_matches_async(self, seq_first: int, kp1: torch.Tensor, kp2: torch.Tensor, i12: torch.Tensor, offsets,
                          img_shape: Sequence[int], max_pairs: int = 0, max_matches_per_pair: int = 0, one_order: bool = False):
        """Asynchronous, device-resident upload of the matches of consecutive slots (pd_ggs_set_matches_csr_async).

        kp1 / kp2: float64 [total, 2], i12: int64 [total, 2] -- CUDA tensors on this device or PINNED host tensors (the
        kernels then read them over PCIe); ``offsets`` [n + 1]: CSR offsets, sequence b = rows offsets[b]:offsets[b+1].
        Runs on torch's current stream; returns at once.  The engine keeps the tensors alive until the upload has
        executed.  ``max_pairs`` / ``max_matches_per_pair``: capacity hints (include/pd_engine.h pd_match_hints); ``one_order``: every
        frame pair occurs in one order only (hloc's exhaustive i < j pairs: PD_MATCH_HINT_ONE_ORDER) -- with it the engine plans the same
        launch shape as for host-uploaded tables."""
        for name, t, dt in (("kp1", kp1, torch.float64), ("kp2", kp2, torch.float64), ("i12", i12, torch.int64)):
            if t.dtype != dt or t.dim() != 2 or t.shape[1] != 2 or not t.is_contiguous():
                raise ValueError(f"{name} must be a contiguous {dt} tensor of shape [total, 2]")
            if not (t.is_cuda and t.device == self.device) and not (t.device.type == "cpu" and t.is_pinned()):
                raise ValueError(f"{name} must live on {self.device} or in pinned host memory (got {t.device}, "
                                 f"pinned={t.device.type == 'cpu' and t.is_pinned()})")
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        if off.ndim != 1 or len(off) < 2 or off[0] < 0 or off[-1] > kp1.shape[0] or kp1.shape != kp2.shape or kp1.shape != i12.shape:
            raise ValueError("offsets must be [n_seqs + 1] within the rows of kp1 / kp2 / i12 (equal shapes)")
        n, _, h, w = (int(v) for v in img_shape)
        hints = _lib.pd_match_hints(int(max_pairs) | (_lib.PD_MATCH_HINT_ONE_ORDER if (one_order and max_pairs > 0) else 0), int(max_matches_per_pair))
        cache = self.__dict__.setdefault("_match_ids", {})
        for b in range(len(off) - 1):
            cache.pop(int(seq_first) + b, None)                    # host.upload_matches' identity cache
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pd_ggs_set_matches_csr_async(
                self._h, int(seq_first), len(off) - 1, off.ctypes.data_as(C.POINTER(C.c_int64)), kp1.data_ptr(), kp2.data_ptr(),
                i12.data_ptr(), n, h, w, C.byref(hints), self._stream()), "pd_ggs_set_matches_csr_async")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        keep = self.__dict__.setdefault("_upload_keep", [])
        keep[:] = [(e, ts) for e, ts in keep if not e.query()]
        keep.append((ev, (kp1, kp2, i12)))

    def ggs_guide(self, model_mean: torch.Tensor, t: int, cfg=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """geometry_guided_sampling(model_mean, t, ...) for every sequence b (match slot b)."""
        B, N, _ = model_mean.shape
        x = self._f32(model_mean).clone()
        stats = torch.zeros(B, 5, 4, device=self.device)
        c = cfg if isinstance(cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(cfg)
        _lib.check(self.lib.pd_ggs_guide(self._h, x.data_ptr(), B, N, int(t), C.byref(c), stats.data_ptr(), self._stream()),
                   "pd_ggs_guide")
        return x, stats

    def ggs_optimize(self, model_mean: torch.Tensor, update_R=True, update_T=True, update_FL=True, cfg=None,
                     trace: bool = False):
        B, N, _ = model_mean.shape
        x = self._f32(model_mean).clone()
        c = cfg if isinstance(cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(cfg)
        iters = c.iter_num * (2 if (update_R and update_T and update_FL) else 1)
        stats = torch.zeros(B, 1, 4, device=self.device)
        tr = torch.zeros(B, max(iters, 1), N * 9 + 3, device=self.device) if trace else None
        _lib.check(self.lib.pd_ggs_optimize(self._h, x.data_ptr(), B, N, int(update_R), int(update_T), int(update_FL),
                                            C.byref(c), stats.data_ptr(), _ptr(tr), self._stream()), "pd_ggs_optimize")
        return x, stats[:, 0], tr

    def ggs_loss_grad(self, x: torch.Tensor, update_R=True, update_T=True, update_FL=True, cfg=None):
        B, N, _ = x.shape
        x = self._f32(x)
        c = cfg if isinstance(cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(cfg)
        loss = torch.zeros(B, 4, device=self.device)
        grad = torch.zeros_like(x)
        _lib.check(self.lib.pd_ggs_loss_grad(self._h, x.data_ptr(), B, N, int(update_R), int(update_T), int(update_FL),
                                             C.byref(c), loss.data_ptr(), grad.data_ptr(), self._stream()), "pd_ggs_loss_grad")
        return loss, grad

    # ---------------------------------------------------------------- sampler
    def sample(self, z: torch.Tensor, noise: torch.Tensor, cond_start_step: int = 0, ggs_cfg=None,
               use_graph: bool = True, want_process: bool = True, phase: int = 0, out=None):
        """GaussianDiffusion.sample (models/gaussian_diffuser.py:284-306).  ``noise`` is
        [T+1,B,N,9]: noise[0] the initial randn, noise[1+k] the randn_like of step t = T-1-k.

        ``phase`` (pd_engine.h PD_PHASE_*): 0 = the whole loop; 1 = inputs + unguided steps only,
        2 = guided steps + results.  A phase-2 call takes the ``(pose, process, stats)`` tuple the
        phase-1 call returned as ``out`` (the buffers are written by phase 2 only)."""
        B, N, _ = z.shape
        T = self.timesteps
        z = self._f32(z, (B, N, self.z_dim))
        noise = self._f32(noise, (T + 1, B, N, 9))
        has_ggs = ggs_cfg is not None and cond_start_step > 0
        c = None
        if has_ggs:
            c = ggs_cfg if isinstance(ggs_cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(ggs_cfg)
        if out is None:
            pose = torch.empty(B, N, 9, device=self.device)
            process = torch.empty(T + 1, B, N, 9, device=self.device) if want_process else None
            stats = torch.zeros(max(cond_start_step, 1), B, 5, 4, device=self.device) if has_ggs else None
        else:
            pose, process, stats = out
        _lib.check(self.lib.pd_sample_phase(self._h, z.data_ptr(), noise.data_ptr(), B, N, int(cond_start_step),
                                            C.byref(c) if c is not None else None, int(phase), pose.data_ptr(),
                                            _ptr(process), _ptr(stats), int(bool(use_graph)), self._stream()),
                   "pd_sample_phase")
        return pose, process, stats

    def set_split_precision(self, mode):
        """Encoder GEMMs of the denoiser at >= 1024 token rows (include/pd_engine.h PD_OPT_DENOISER_SPLIT): 0 / False exact fp32,
        1 / True bf16 hi + lo (fast mode, ~16 bits: narrower than the reference's fp32), 2 fp16 hi + lo with static power-of-two
        scales (22 bits, fp32 accumulation: fp32-grade)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pd_engine_set_option(self._h, _lib.PD_OPT_DENOISER_SPLIT, int(mode)), "pd_engine_set_option")

    def set_option(self, option: int, value: int):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.pd_engine_set_option(self._h, int(option), int(value)), "pd_engine_set_option")

    def get_option(self, option: int) -> int:
        """pd_engine_get_option: e.g. ``_lib.PD_OPT_DENOISER_SPLIT`` -> the encoder GEMM mode in force (0 after the non-finite-weights downgrade)."""
        v = C.c_int(0)
        _lib.check(self.lib.pd_engine_get_option(self._h, int(option), C.byref(v)), "pd_engine_get_option")
        return int(v.value)

    def pose_to_camera(self, enc: torch.Tensor, log_focal_length_bias: float = 1.8, min_focal_length: float = 0.1,
                       max_focal_length: float = 20.0):
        """pose_encoding_to_camera (camera_transform.py:64-105) with its three focal-length parameters."""
        enc = self._f32(enc).reshape(-1, 9)
        n = enc.shape[0]
        R = torch.empty(n, 3, 3, device=self.device)
        T = torch.empty(n, 3, device=self.device)
        F = torch.empty(n, 2, device=self.device)
        _lib.check(self.lib.pd_pose_to_camera_ex(self._h, enc.data_ptr(), n, R.data_ptr(), T.data_ptr(), F.data_ptr(),
                                                 float(log_focal_length_bias), float(min_focal_length), float(max_focal_length),
                                                 self._stream()), "pd_pose_to_camera_ex")
        return R, T, F

    def lane_tables(self, seq: int = 0):
        """(lane items, waves, base item length, [steps per wave]) of match slot ``seq`` as the device holds them (pd_debug_lane_tables)."""
        buf = (C.c_int * 20)()
        _lib.check(self.lib.pd_debug_lane_tables(self._h, int(seq), buf, 20), "pd_debug_lane_tables")
        return int(buf[0]), int(buf[1]), int(buf[2]), [int(buf[4 + w]) for w in range(int(buf[1]))]

    def ggs_launch_stamps(self, n: int, out: Optional[torch.Tensor] = None):
        """(int64 tensor [n, 2] on the device, ticks per millisecond): {start, end} of the GGS launches of guided steps 0 .. n-1 of the last
        sampling pass, recorded by the kernel itself (pd_ggs_launch_stamps); copied on the current stream behind the pass.  ``out``: a
        preallocated contiguous int64 [n, 2] device tensor (no allocation between the passes of a pipe)."""
        if out is None:
            out = torch.zeros(int(n), 2, dtype=torch.int64, device=self.device)
        elif out.dtype != torch.int64 or tuple(out.shape) != (int(n), 2) or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous int64 [n, 2] tensor on the engine's device")
        khz = C.c_int(0)
        _lib.check(self.lib.pd_ggs_launch_stamps(self._h, out.data_ptr(), int(n), C.byref(khz), self._stream()), "pd_ggs_launch_stamps")
        return out, float(khz.value)

    def time_kernel(self, what: int, B: int, N: int, cfg=None, reps: int = 10) -> float:
        c = cfg if isinstance(cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(cfg)
        ms = C.c_float(0.0)
        _lib.check(self.lib.pd_time_kernel(self._h, int(what), B, N, C.byref(c), int(reps), C.byref(ms), self._stream()),
                   "pd_time_kernel")
        return float(ms.value)

    def ggs_prof(self, enable=True):
        """Debug: enable the GGS phase counters / read them -> dict of cycles per iteration."""
        buf = (C.c_longlong * 16)()
        _lib.check(self.lib.pd_debug_ggs_prof(self._h, int(enable), buf), "pd_debug_ggs_prof")
        v = list(buf)
        it = max(v[5], 1)
        return {"P1": v[0] / it, "P2": v[1] / it, "xchg": v[2] / it, "P3": v[3] / it, "P4": v[4] / it, "iters": v[5],
                "P3a": v[6] / it, "P3_wait1": v[7] / it, "P3b": v[8] / it,
                # inside the match pass, per iteration: slot claim, LDS-DMA issue, wait for the staged item, the item pass, reduction + store
                "P2_claim": v[10] / it, "P2_issue": v[11] / it, "P2_wait": v[12] / it, "P2_pass": v[13] / it, "P2_reduce": v[14] / it}
