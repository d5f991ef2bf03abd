"""placeholder: imported by demo.py:25, never called on the sampling path"""


def get_visdom_connection(*args, **kwargs):
    raise ConnectionError("visdom is not installed (posediffusion_amd stand-in)")
