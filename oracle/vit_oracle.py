"""CPU oracle for the image feature extractor.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's ``MultiScaleImageFeatureExtractor`` (models/image_feature_extractor.py:28-87) wraps a DINO ViT-S/16 it
fetches with ``torch.hub.load("facebookresearch/dino:main", "dino_vits16")`` (:40-42).  That model's code
(facebookresearch/dino, vision_transformer.py) is a third-party dependency that is NOT under /root/reference and cannot
be fetched here, so ``DinoViT`` below RESTATES its published algorithm (parity against the DINO SOURCE is unpinned;
weights are random-init, the trained checkpoint is not available offline).  What CAN be pinned offline is pinned
(round 4): ``to_hf_vit`` loads the same weights into HuggingFace ``transformers.ViTModel`` -- an independent
implementation of the same published architecture, the class the Hub's ``facebook/dino-vits16`` conversion of this very
checkpoint instantiates -- and tests/test_oracle_golden.py compares the two in fp64 (2e-15 at 224 x 224; at other input
sizes with the position grid resampled by DINO's rule, which HF resamples by another one).  The multi-scale wrapper itself -- ImageNet
normalisation, bilinear rescaling by 1, 1/2, 1/3, averaging of the CLS features (:57-87) -- is the reference's own code:
``multiscale_features`` restates it and ``ref_stubs.load_reference_extractor`` runs the reference file in place around
this ViT to pin that restatement.

State-dict names follow DINO's (patch_embed.proj, cls_token, pos_embed, blocks.N.{norm1,attn.qkv,attn.proj,norm2,
mlp.fc1,mlp.fc2}, norm), so a real ``dino_vits16`` checkpoint loads with ``strict=True``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

RESNET_MEAN = (0.485, 0.456, 0.406)          # models/image_feature_extractor.py:24-25
RESNET_STD = (0.229, 0.224, 0.225)


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))          # nn.GELU(): exact erf form


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, dim):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoViT(nn.Module):
    """DINO ``vit_small(patch_size=16)``: 12 blocks, dim 384, 6 heads, MLP 1536, LayerNorm eps 1e-6, CLS token output."""

    def __init__(self, img_size=224, patch_size=16, dim=384, depth=12, num_heads=6):
        super().__init__()
        self.patch_embed = _PatchEmbed(patch_size, 3, dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.blocks = nn.ModuleList([_Block(dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)

    def interpolate_pos_encoding(self, x, w, h):
        """Bicubic resampling of the 14 x 14 position grid for other input sizes, with DINO's `+ 0.1` on the target grid
        size before it forms the scale factor (so that floor() lands on the intended size)."""
        npatch, N = x.shape[1] - 1, self.pos_embed.shape[1] - 1
        if npatch == N and w == h:
            return self.pos_embed
        class_pos, patch_pos = self.pos_embed[:, 0], self.pos_embed[:, 1:]
        dim = x.shape[-1]
        w0, h0 = w // self.patch_embed.patch_size + 0.1, h // self.patch_embed.patch_size + 0.1
        g = int(math.sqrt(N))
        patch_pos = F.interpolate(patch_pos.reshape(1, g, g, dim).permute(0, 3, 1, 2), scale_factor=(w0 / g, h0 / g), mode="bicubic")
        assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
        patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)

    def forward(self, x):
        B, _, w, h = x.shape
        t = self.patch_embed(x)
        t = torch.cat((self.cls_token.expand(B, -1, -1), t), dim=1)
        t = t + self.interpolate_pos_encoding(t, w, h)
        for blk in self.blocks:
            t = blk(t)
        return self.norm(t)[:, 0]


def make_vit(seed: int = 0, dtype=torch.float32) -> DinoViT:
    """Random-init ViT-S/16 with non-trivial LayerNorm affines and biases (so that folding mistakes show)."""
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    m = DinoViT()
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name == "norm.weight":
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2 and "pos_embed" not in name and "cls_token" not in name:
                p.copy_(torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(p[0].numel())))
    return m.to(dtype).eval()


@torch.no_grad()
def multiscale_features(net: nn.Module, image_rgb: torch.Tensor, scale_factors=(1, 1 / 2, 1 / 3)) -> torch.Tensor:
    """models/image_feature_extractor.py:57-87: normalise, run the backbone at every scale, average."""
    mean = torch.tensor(RESNET_MEAN, dtype=image_rgb.dtype).view(1, 3, 1, 1)
    std = torch.tensor(RESNET_STD, dtype=image_rgb.dtype).view(1, 3, 1, 1)
    img = (image_rgb - mean) / std                                                      # :62-63
    feats = None
    for sf in scale_factors:                                                            # :71-80
        inp = img if sf == 1 else F.interpolate(img, scale_factor=sf, mode="bilinear", align_corners=False)   # :86-87
        f = net(inp)
        feats = f if feats is None else feats + f
    return feats / len(scale_factors)                                                   # :82-83


def to_hf_vit(net: "DinoViT", img_size: int = 224):
    """The same weights in HuggingFace ``transformers.ViTModel`` (ViT-S/16 configuration, LayerNorm eps 1e-6 as in DINO's
    ``partial(nn.LayerNorm, eps=1e-6)``, no pooler): DINO's fused ``attn.qkv`` rows split into q / k / v projections, every
    other tensor renamed.  ``img_size`` other than 224: the position table is resampled by ``DinoViT.interpolate_pos_encoding``
    first (HF's own resampling passes the target SIZE to ``F.interpolate``; DINO passes a scale factor of (size + 0.1) / 14,
    which samples the bicubic kernel at slightly different points -- the reference inherits DINO's).  Test infrastructure."""
    from transformers import ViTConfig, ViTModel
    sd = net.state_dict()
    dim, depth = sd["cls_token"].shape[-1], len(net.blocks)
    heads = net.blocks[0].attn.num_heads
    cfg = ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=sd["blocks.0.mlp.fc1.weight"].shape[0],
                    hidden_act="gelu", layer_norm_eps=1e-6, image_size=img_size, patch_size=net.patch_embed.patch_size, num_channels=3,
                    qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = ViTModel(cfg, add_pooling_layer=False).to(sd["cls_token"].dtype).eval()
    pos = sd["pos_embed"]
    if img_size != 224:
        g = img_size // net.patch_embed.patch_size
        with torch.no_grad():
            pos = net.interpolate_pos_encoding(torch.zeros(1, g * g + 1, dim, dtype=pos.dtype), img_size, img_size)
    out = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": pos,
           "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
           "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
           "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    for i in range(depth):
        w, b = sd[f"blocks.{i}.attn.qkv.weight"], sd[f"blocks.{i}.attn.qkv.bias"]
        for j, n in enumerate("qkv"):
            out[f"layers.{i}.attention.{n}_proj.weight"] = w[dim * j:dim * (j + 1)]
            out[f"layers.{i}.attention.{n}_proj.bias"] = b[dim * j:dim * (j + 1)]
        for hf_name, dino_name in (("attention.o_proj", "attn.proj"), ("layernorm_before", "norm1"), ("layernorm_after", "norm2"),
                                   ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
            for t in ("weight", "bias"):
                out[f"layers.{i}.{hf_name}.{t}"] = sd[f"blocks.{i}.{dino_name}.{t}"]
    hf.load_state_dict(out, strict=True)
    return hf
