"""Placeholder for MultiScaleImageFeatureExtractor (pose_diffusion/models/image_feature_extractor.py).

OUT OF SCOPE for this engine: the DINO ViT-S/16 backbone comes from
torch.hub (no network here) and is not on the sampling hot path.  The class keeps the `_target_`
name resolvable; give it a backbone (`extractor.backbone = callable(images[BN,3,H,W]) -> [BN,384]`)
or pass precomputed features with `PoseDiffusionModel.forward(..., z=features)`."""
import torch.nn as nn


class MultiScaleImageFeatureExtractor(nn.Module):
    def __init__(self, modelname: str = "dino_vits16", freeze: bool = False, scale_factors: list = [1, 1 / 2, 1 / 3]):
        super().__init__()
        self.freeze, self.scale_factors, self.modelname = freeze, scale_factors, modelname
        self.backbone = None
        self.feature_dim = 384

    def forward(self, image_rgb):
        if self.backbone is None:
            raise RuntimeError("image features are out of scope of posediffusion_amd (no DINO weights offline): set "
                               "`.backbone` or pass `z=` to PoseDiffusionModel.forward")
        return self.backbone(image_rgb)
