"""pytorch3d.ops.corresponding_cameras_alignment (demo.py:24, :127-129) on the engine's alignment kernel
(csrc/pd_metrics.hip, through the drop-in util/metric.py)."""


def corresponding_cameras_alignment(cameras_src, cameras_tgt, estimate_scale=True, mode="extrinsics", eps=1e-9):
    import sys

    from posediffusion_amd import DROPIN_PATH
    if DROPIN_PATH not in sys.path:
        sys.path.insert(0, DROPIN_PATH)
    from util.metric import corresponding_cameras_alignment as impl
    return impl(cameras_src, cameras_tgt, estimate_scale=estimate_scale, mode=mode, eps=eps)
