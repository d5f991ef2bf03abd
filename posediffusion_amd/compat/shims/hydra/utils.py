"""hydra.utils.{instantiate, get_original_cwd} as pose_diffusion/demo.py:19,46,49 uses them."""
import os

from posediffusion_amd.compat import _local_instantiate

_ORIGINAL_CWD = None


def get_original_cwd():
    return _ORIGINAL_CWD if _ORIGINAL_CWD is not None else os.getcwd()


def instantiate(cfg, *args, **kwargs):
    kwargs.setdefault("_recursive_", True)
    return _local_instantiate(cfg, *args, **kwargs)


instantiate.__pd_shim__ = True


def to_absolute_path(path):
    return path if os.path.isabs(path) else os.path.join(get_original_cwd(), path)
