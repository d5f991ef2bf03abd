"""pose_encoding_to_camera (pose_diffusion/util/camera_transform.py:64-105) on the HIP engine."""
import torch

from posediffusion_amd import _lib
from posediffusion_amd.compat import PerspectiveCameras


@torch.no_grad()
def pose_encoding_to_camera(pose_encoding, pose_encoding_type="absT_quaR_logFL", log_focal_length_bias=1.8,
                            min_focal_length=0.1, max_focal_length=20, return_dict=False, engine=None):
    if pose_encoding_type != "absT_quaR_logFL":
        raise ValueError(f"Unknown pose encoding {pose_encoding_type}")           # camera_transform.py:98-99
    if engine is None:
        from posediffusion_amd.host import current_engine
        engine = current_engine(pose_encoding.device)
    R, T, f = engine.pose_to_camera(pose_encoding, log_focal_length_bias, min_focal_length, max_focal_length)
    if return_dict:
        return {"focal_length": f, "R": R, "T": T}
    return PerspectiveCameras(focal_length=f, R=R, T=T, device=R.device)
