"""Drop-in `MultiScaleImageFeatureExtractor` (pose_diffusion/models/image_feature_extractor.py:28-87) on the HIP kernels
of csrc/pd_vit.hip.

The reference fetches its backbone with ``torch.hub.load("facebookresearch/dino:main", "dino_vits16")`` (:40-42); there is
no network here, and the hub code is third-party.  This module therefore owns a parameter tree with DINO's state_dict
names (``_net.patch_embed.proj``, ``_net.cls_token``, ``_net.pos_embed``, ``_net.blocks.N.{norm1, attn.qkv, attn.proj,
norm2, mlp.fc1, mlp.fc2}``, ``_net.norm``), so a PoseDiffusion checkpoint (or a plain ``dino_vits16`` checkpoint loaded
into ``._net``) loads with ``strict=True``; the arithmetic runs in the engine (`posediffusion_amd.vit.VitEngine`).
Other backbones of the reference (``resnet*``, ``dinov2*``) are not implemented: inject a callable as ``.backbone``."""
import torch
import torch.nn as nn

_RESNET_MEAN = [0.485, 0.456, 0.406]
_RESNET_STD = [0.229, 0.224, 0.225]


class _Holder(nn.Module):
    """Parameter container; the compute lives in the HIP engine."""


def _dino_vits16_parameters(img_size=224, patch=16, dim=384, depth=12, mlp=1536) -> nn.Module:
    net = _Holder()
    net.patch_embed = _Holder()
    net.patch_embed.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)
    net.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
    net.pos_embed = nn.Parameter(torch.zeros(1, (img_size // patch) ** 2 + 1, dim))
    nn.init.trunc_normal_(net.pos_embed, std=0.02)
    nn.init.trunc_normal_(net.cls_token, std=0.02)
    blocks = []
    for _ in range(depth):
        b = _Holder()
        b.norm1 = nn.LayerNorm(dim, eps=1e-6)
        b.attn = _Holder()
        b.attn.qkv = nn.Linear(dim, dim * 3, bias=True)
        b.attn.proj = nn.Linear(dim, dim)
        b.norm2 = nn.LayerNorm(dim, eps=1e-6)
        b.mlp = _Holder()
        b.mlp.fc1 = nn.Linear(dim, mlp)
        b.mlp.fc2 = nn.Linear(mlp, dim)
        blocks.append(b)
    net.blocks = nn.ModuleList(blocks)
    net.norm = nn.LayerNorm(dim, eps=1e-6)
    return net


class MultiScaleImageFeatureExtractor(nn.Module):
    def __init__(self, modelname: str = "dino_vits16", freeze: bool = False, scale_factors: list = [1, 1 / 2, 1 / 3]):
        super().__init__()
        self.freeze, self.scale_factors, self.modelname = freeze, scale_factors, modelname
        self.backbone = None                       # optional injected callable images[BN,3,H,W] -> [BN,C]
        if modelname == "dino_vits16":
            self._net = _dino_vits16_parameters()
            self._output_dim = self._net.norm.weight.shape[0]                   # :42
        elif "res" in modelname or "dino" in modelname:
            self._net = None                       # resnet* / dinov2* / other DINO sizes: give `.backbone`
            self._output_dim = None
        else:
            raise ValueError(f"Unknown model name {modelname}")                 # :43-44
        for name, value in (("_resnet_mean", _RESNET_MEAN), ("_resnet_std", _RESNET_STD)):
            self.register_buffer(name, torch.FloatTensor(value).view(1, 3, 1, 1), persistent=False)       # :46-47
        if self.freeze:
            for p in self.parameters():
                p.requires_grad = False
        self.feature_dim = self._output_dim

    def get_output_dim(self):
        return self._output_dim

    def _engine(self):
        from posediffusion_amd.vit import VitEngine, vit_state
        dev = self._net.norm.weight.device
        fp = tuple((p.data_ptr(), p._version) for p in self._net.parameters())
        ent = self.__dict__.get("_pd_vit_cache")
        if ent is None or ent[0] != fp:
            if ent is not None:
                ent[1].close()
            ent = (fp, VitEngine(vit_state(self._net), dev))
            self.__dict__["_pd_vit_cache"] = ent
        return ent[1]

    @torch.no_grad()
    def forward(self, image_rgb: torch.Tensor) -> torch.Tensor:
        if self.backbone is not None:
            return self.backbone(image_rgb)
        if self._net is None:
            raise RuntimeError(f"backbone {self.modelname!r} is not implemented by posediffusion_amd (dino_vits16 is): set "
                               "`.backbone` or pass `z=` to PoseDiffusionModel.forward")
        if self._net.norm.weight.device.type != "cuda":
            raise RuntimeError("posediffusion_amd image features run only on an AMD GPU; move the model with .to('cuda')")
        return self._engine().multiscale(image_rgb, self.scale_factors)
