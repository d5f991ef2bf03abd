"""pytorch3d.implicitron.tools.vis_utils stand-in (demo.py:25, util/train_util.py:20): no visualisation server."""
__pd_shim__ = True


def get_visdom_connection(*args, **kwargs):
    raise ConnectionError("visdom is not installed (posediffusion_amd stand-in): no visualisation server")
