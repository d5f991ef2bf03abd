"""pytorch3d.implicitron.tools.model_io stand-in: imported by demo.py:25, never called on the sampling path (checkpoint io of Implicitron is out of scope)."""
__pd_shim__ = True
