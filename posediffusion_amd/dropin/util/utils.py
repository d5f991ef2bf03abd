"""Mirror of pose_diffusion/util/utils.py:14-17 (RNG protocol of the demo)."""
import random

import numpy as np
import torch


def seed_all_random_engines(seed: int) -> None:
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)
