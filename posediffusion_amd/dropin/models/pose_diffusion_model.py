"""Drop-in `PoseDiffusionModel` (pose_diffusion/models/pose_diffusion_model.py:35-142, inference branch).

``forward(image, gt_cameras=None, sequence_name=None, cond_fn=None, cond_start_step=0, training=True,
batch_repeat=-1)`` keeps the reference signature; `training=False` returns
``{"pred_cameras": PerspectiveCameras(R, T, focal_length) in PyTorch3D NDC, "z": features}``.
Extension: ``z=`` accepts precomputed image features instead of ``image`` (skips the feature extractor)."""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from posediffusion_amd import host
from posediffusion_amd.compat import instantiate
from util.camera_transform import pose_encoding_to_camera


class PoseDiffusionModel(nn.Module):
    def __init__(self, pose_encoding_type: str, IMAGE_FEATURE_EXTRACTOR: Dict, DIFFUSER: Dict, DENOISER: Dict):
        super().__init__()
        self.pose_encoding_type = pose_encoding_type
        self.image_feature_extractor = instantiate(IMAGE_FEATURE_EXTRACTOR, _recursive_=False)
        self.diffuser = instantiate(DIFFUSER, _recursive_=False)
        denoiser = instantiate(DENOISER, _recursive_=False)
        self.diffuser.model = denoiser
        self.target_dim = denoiser.target_dim
        self.apply(self._init_weights)

    def _init_weights(self, m):
        # same rule as pose_diffusion_model.py:67-74
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, image: torch.Tensor = None, gt_cameras=None, sequence_name: Optional[List[str]] = None, cond_fn=None,
                cond_start_step=0, training=True, batch_repeat=-1, z: Optional[torch.Tensor] = None):
        if training:
            raise NotImplementedError("training is out of scope of the MI355X sampling engine; call with training=False")
        if z is None:
            B, N = image.shape[0], image.shape[1]
            z = self.image_feature_extractor(image.reshape(B * N, *image.shape[2:])).reshape(B, N, -1)
        B, N, _ = z.shape
        pose_encoding, _ = self.diffuser.sample(shape=[B, N, self.target_dim], z=z, cond_fn=cond_fn,
                                                cond_start_step=cond_start_step)
        eng = host.get_engine(self.diffuser.model, self.diffuser, B, N)
        pred_cameras = pose_encoding_to_camera(pose_encoding, pose_encoding_type=self.pose_encoding_type, engine=eng)
        return {"pred_cameras": pred_cameras, "z": z, "pose_encoding": pose_encoding}
