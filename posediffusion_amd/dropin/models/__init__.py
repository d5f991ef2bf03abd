"""Drop-in `models` package: the five Hydra `_target_` names of cfgs/default.yaml:17,22,26,28,39."""
from .pose_diffusion_model import PoseDiffusionModel
from .denoiser import Denoiser, TransformerEncoderWrapper
from .gaussian_diffuser import GaussianDiffusion
from .image_feature_extractor import MultiScaleImageFeatureExtractor

__all__ = ["PoseDiffusionModel", "Denoiser", "TransformerEncoderWrapper", "GaussianDiffusion",
           "MultiScaleImageFeatureExtractor"]
