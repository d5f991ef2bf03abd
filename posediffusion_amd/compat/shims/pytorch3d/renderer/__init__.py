from .cameras import CamerasBase, PerspectiveCameras  # noqa: F401
