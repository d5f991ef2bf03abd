"""pytorch3d.renderer.cameras.{PerspectiveCameras, CamerasBase} (demo.py:23, :122-124)."""
from posediffusion_amd.compat import _LocalPerspectiveCameras


class CamerasBase:
    pass


class PerspectiveCameras(_LocalPerspectiveCameras, CamerasBase):
    __pd_shim__ = True

    def to(self, device):
        return PerspectiveCameras(self.focal_length, self.R, self.T, device, self.principal_point)
