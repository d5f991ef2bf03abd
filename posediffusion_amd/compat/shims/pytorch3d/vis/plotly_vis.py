"""pytorch3d.vis.plotly_vis.plot_scene (demo.py:26, :144): visualisation only, inside demo.py's try/except."""


def plot_scene(*args, **kwargs):
    raise RuntimeError("plotly visualisation is not available (posediffusion_amd stand-in for pytorch3d.vis)")
