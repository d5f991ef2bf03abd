"""Stand-in for hydra.main as pose_diffusion/demo.py:36 uses it: load ``<dir of the decorated function's file>/
<config_path>/<config_name>.yaml``, apply ``key=value`` / ``a.b=value`` / ``+key=value`` command-line overrides (values
parsed as YAML scalars), call the function with the config.  The working directory is left alone
(``hydra.job.chdir=False`` semantics; ``get_original_cwd()`` returns the launch directory either way), no ``outputs/``
tree is written.  Only reachable when the real hydra is not installed."""
import functools
import os
import sys

import yaml

from posediffusion_amd.compat import AttrDict

from . import utils  # noqa: F401

__pd_shim__ = True
__version__ = "0.0-pd-shim"


def _parse_value(val):
    """Hydra's override grammar for scalars: int, float (incl. 1e-3, which YAML 1.1 reads as a string), bool, null,
    quoted / bare strings; [..] and {..} through YAML."""
    low = val.strip().lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none", "~"):
        return None
    for conv in (int, float):
        try:
            return conv(val)
        except ValueError:
            pass
    if val[:1] in "[{":
        return yaml.safe_load(val)
    if len(val) >= 2 and val[0] == val[-1] and val[0] in "'\"":
        return val[1:-1]
    return val


def _apply_override(cfg, text):
    if "=" not in text:
        raise ValueError(f"hydra override {text!r}: expected key=value")
    key, val = text.split("=", 1)
    key = key.lstrip("+")
    node = cfg
    parts = key.split(".")
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            node[p] = AttrDict()
        node = node[p]
    node[parts[-1]] = _parse_value(val)


def main(config_path=None, config_name=None, version_base=None):
    def decorator(fn):
        @functools.wraps(fn)
        def wrapper(cfg_passthrough=None):
            if cfg_passthrough is not None:
                return fn(cfg_passthrough)
            here = os.path.dirname(os.path.abspath(fn.__code__.co_filename))
            path = os.path.normpath(os.path.join(here, config_path or ".", (config_name or "config") + ".yaml"))
            with open(path) as f:
                cfg = AttrDict(yaml.safe_load(f) or {})
            for arg in sys.argv[1:]:
                if arg.startswith("-"):
                    continue                      # hydra's own flags (--cfg, -m ...) are not emulated
                _apply_override(cfg, arg)
            utils._ORIGINAL_CWD = os.getcwd()
            return fn(cfg)
        return wrapper
    return decorator
