"""pytorch3d.implicitron.tools as pose_diffusion/demo.py:25 needs it: `from pytorch3d.implicitron.tools import model_io, vis_utils`
imports the two names and never uses them on the sampling path; util/train_util.py:20 imports `vis_utils.get_visdom_connection` by its
dotted module path.  Both are real (one-line) submodules, so either import form resolves; visualisation and checkpoint io of
Implicitron are out of scope."""
from . import model_io, vis_utils  # noqa: F401
