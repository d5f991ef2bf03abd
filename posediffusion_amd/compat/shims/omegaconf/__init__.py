"""Stand-in for the part of omegaconf that pose_diffusion/demo.py uses (:17, :38-40, :86-87) -- PyYAML underneath.
Only reachable when the real omegaconf is not installed (posediffusion_amd.compat.install_shims appends this directory)."""
import yaml

from posediffusion_amd.compat import AttrDict, to_container

__pd_shim__ = True
DictConfig = AttrDict
ListConfig = list


class OmegaConf:
    @staticmethod
    def create(obj=None):
        if isinstance(obj, str):
            obj = yaml.safe_load(obj)
        return AttrDict(obj or {})

    @staticmethod
    def load(path):
        with open(path) as f:
            return AttrDict(yaml.safe_load(f) or {})

    @staticmethod
    def set_struct(cfg, value):          # struct mode only guards against typos; nothing to do (demo.py:38 switches it off)
        return None

    @staticmethod
    def to_yaml(cfg, resolve=False, sort_keys=False):
        return yaml.safe_dump(to_container(cfg), default_flow_style=False, sort_keys=sort_keys)

    @staticmethod
    def to_container(cfg, resolve=False, **_):
        return to_container(cfg)

    @staticmethod
    def merge(*cfgs):
        def rec(dst, src):
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    rec(dst[k], v)
                else:
                    dst[k] = v
        out = AttrDict()
        for c in cfgs:
            rec(out, AttrDict(to_container(c)))
        return out
