"""Run an UNMODIFIED reference entry script (pose_diffusion/demo.py) on the MI355X engine.

    cd /path/to/PoseDiffusion/pose_diffusion
    python -m posediffusion_amd.run_reference demo.py image_folder=samples/apple ckpt=/path/to/ckpt.pth

The reference resolves `models` / `util` from its own directory (the script directory is sys.path[0] under
`python demo.py`), so "switching" to this engine means putting the drop-in packages first.  This launcher does exactly
that and nothing else: sys.path = [posediffusion_amd/dropin, ...], stand-ins for omegaconf / hydra / pytorch3d / visdom
appended for environments that lack them (installed packages win), then the script file is executed from where it lies
with `runpy` as `__main__` -- same file, same Hydra cfgs (`../cfgs/default.yaml` relative to the script), same
command-line overrides.  INTEGRATION.md section 3.
"""
from __future__ import annotations

import os
import runpy
import sys


def run(script: str, overrides=()):
    from posediffusion_amd import DROPIN_PATH
    from posediffusion_amd.compat import install_shims
    script = os.path.abspath(script)
    if not os.path.isfile(script):
        raise FileNotFoundError(script)
    ref_dir = os.path.dirname(script)
    # the reference's own packages must not shadow the drop-in ones
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != ref_dir]
    if DROPIN_PATH in sys.path:
        sys.path.remove(DROPIN_PATH)
    sys.path.insert(0, DROPIN_PATH)
    for name in [m for m in sys.modules if m in ("models", "util") or m.startswith(("models.", "util."))]:
        if not os.path.abspath(getattr(sys.modules[name], "__file__", "") or "").startswith(DROPIN_PATH):
            del sys.modules[name]
    install_shims()
    # (the drop-in prints the reference's `t=.. | sampson=..` lines, geometry_guided_sampling.py:124, by itself; PD_GGS_VERBOSE=0 mutes them)
    old_argv = sys.argv
    sys.argv = [script] + list(overrides)
    try:
        return runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old_argv


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    run(sys.argv[1], sys.argv[2:])


if __name__ == "__main__":
    main()
