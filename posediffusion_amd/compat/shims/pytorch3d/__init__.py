"""Stand-in for the pytorch3d names pose_diffusion/demo.py:23-26 imports.  Everything numerical is served by the HIP
engine through the drop-in `util` package (cameras container, Umeyama-style alignment); the visualisation / io helpers
demo.py imports but the sampling path never calls are placeholders.  Only reachable when pytorch3d is not installed."""
__pd_shim__ = True
__version__ = "0.0-pd-shim"
