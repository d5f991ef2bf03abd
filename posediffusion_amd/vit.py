"""Host mirror of the image feature extractor: MultiScaleImageFeatureExtractor
(pose_diffusion/models/image_feature_extractor.py:28-87) around a DINO ViT-S/16, on the kernels of csrc/pd_vit.hip.

`VitEngine` owns the repacked weights; `multiscale` runs _compute_multiscale_features (:65-84): one C-ABI call per scale.
DINO's `interpolate_pos_encoding` (bicubic resampling of the 14 x 14 position grid with its `+ 0.1` scale-factor rule) is
weight preparation: it is evaluated with torch once per image size and cached."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _lib

PATCH = 16


def vit_state(net: torch.nn.Module) -> Dict[str, torch.Tensor]:
    """The DINO state_dict entries the engine reads, as contiguous fp32 device tensors."""
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    need = ["patch_embed.proj.weight", "patch_embed.proj.bias", "cls_token", "pos_embed", "norm.weight", "norm.bias"]
    blocks = [int(k.split(".")[1]) for k in sd if k.startswith("blocks.")]
    if not blocks:
        raise KeyError("not a DINO ViT state_dict: no `blocks.N.*` entries")
    depth = 1 + max(blocks)
    for l in range(depth):
        need += [f"blocks.{l}.{n}" for n in ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
                                              "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                                              "mlp.fc2.weight", "mlp.fc2.bias")]
    missing = [k for k in need if k not in sd]
    if missing:
        raise KeyError(f"not a DINO ViT state_dict: missing {missing[:4]}...")
    return {k: sd[k] for k in need}


class VitEngine:
    def __init__(self, state: Dict[str, torch.Tensor], device: torch.device):
        device = torch.device(device)
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("posediffusion_amd image features run only on an AMD GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.device = device
        keep = {k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in state.items()}
        self.depth = 1 + max(int(k.split(".")[1]) for k in keep if k.startswith("blocks."))
        n_pos = keep["pos_embed"].reshape(-1, keep["pos_embed"].shape[-1]).shape[0]
        self.grid0 = int(round(math.sqrt(n_pos - 1)))
        self.pos_embed = keep["pos_embed"].reshape(1, n_pos, -1)
        w = _lib.pd_vit_weights()
        w.dim, w.depth, w.num_heads, w.patch_size, w.pos_grid = keep["cls_token"].shape[-1], self.depth, 6, PATCH, self.grid0
        w.mlp_hidden = keep["blocks.0.mlp.fc1.weight"].shape[0]
        p = lambda k: keep[k].data_ptr()   # noqa: E731
        w.patch_w, w.patch_b, w.cls_token, w.pos_embed = p("patch_embed.proj.weight"), p("patch_embed.proj.bias"), p("cls_token"), p("pos_embed")
        w.norm_w, w.norm_b = p("norm.weight"), p("norm.bias")
        for l in range(self.depth):
            L, b = w.layers[l], f"blocks.{l}."
            L.norm1_w, L.norm1_b, L.qkv_w, L.qkv_b = p(b + "norm1.weight"), p(b + "norm1.bias"), p(b + "attn.qkv.weight"), p(b + "attn.qkv.bias")
            L.proj_w, L.proj_b, L.norm2_w, L.norm2_b = p(b + "attn.proj.weight"), p(b + "attn.proj.bias"), p(b + "norm2.weight"), p(b + "norm2.bias")
            L.fc1_w, L.fc1_b, L.fc2_w, L.fc2_b = p(b + "mlp.fc1.weight"), p(b + "mlp.fc1.bias"), p(b + "mlp.fc2.weight"), p(b + "mlp.fc2.bias")
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.pd_vit_create(C.byref(w), C.byref(h)), "pd_vit_create")
        self._h = h
        self._pos_cache: Dict[Tuple[int, int], torch.Tensor] = {}

    def set_exact_fp32(self, on):
        """Arithmetic of batches of >= 1024 token rows (PD_VIT_OPT_EXACT_FP32): False / 0 (default) fp16 hi + lo planes for the four Linear
        layers (22 bits, static scales: fp32-grade); True / 1 the exact-fp32 matrix instruction everywhere; 2 the bf16 planes of rounds 1-5."""
        _lib.check(self.lib.pd_vit_set_option(self._h, 1, int(on)), "pd_vit_set_option")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pd_vit_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _pos_for(self, hs: int, ws: int):
        """DINO interpolate_pos_encoding for an hs x ws pixel input -> [1 + gh*gw, dim] or None for the trained grid."""
        gh, gw = hs // PATCH, ws // PATCH
        if gh == self.grid0 and gw == self.grid0 and hs == ws:
            return None
        key = (hs, ws)
        if key not in self._pos_cache:
            dim, g = self.pos_embed.shape[-1], self.grid0
            # (DINO names the first spatial size w; it passes (w0 / g, h0 / g) for the (rows, cols) of the grid)
            w0, h0 = gh + 0.1, gw + 0.1
            pp = F.interpolate(self.pos_embed[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2), scale_factor=(w0 / g, h0 / g), mode="bicubic")
            if (int(w0), int(h0)) != tuple(pp.shape[-2:]):
                raise RuntimeError("position-grid resampling produced an unexpected size")
            pp = pp.permute(0, 2, 3, 1).reshape(-1, dim)
            self._pos_cache[key] = torch.cat((self.pos_embed[0, :1], pp), dim=0).contiguous()
        return self._pos_cache[key]

    @torch.no_grad()
    def multiscale(self, image_rgb: torch.Tensor, scale_factors: Sequence[float] = (1, 1 / 2, 1 / 3)) -> torch.Tensor:
        """image_rgb [n,3,H,W] in [0,1] -> [n,384] averaged CLS features (image_feature_extractor.py:57-87)."""
        if len(scale_factors) <= 0:
            raise ValueError(f"Wrong format of self.scale_factors: {scale_factors}")                    # :68-69
        x = image_rgb.to(device=self.device, dtype=torch.float32).contiguous()
        n, _, H, W = x.shape
        z = torch.empty(n, self.pos_embed.shape[-1], device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for i, sf in enumerate(scale_factors):
            hs, ws = (H, W) if sf == 1 else (int(math.floor(H * sf)), int(math.floor(W * sf)))
            pos = self._pos_for(hs, ws)
            _lib.check(self.lib.pd_vit_forward_scale(self._h, x.data_ptr(), n, H, W, C.c_double(float(sf)),
                                                     None if pos is None else pos.data_ptr(), C.c_float(1.0 / len(scale_factors)),
                                                     int(i > 0), z.data_ptr(), stream), "pd_vit_forward_scale")
        return z
