"""demo.py:25 imports `model_io` and `vis_utils` and never uses them."""
from . import model_io, vis_utils  # noqa: F401
