"""util/metric.py of the reference (pose_diffusion/util/metric.py) on the HIP kernels of pd_metrics.hip, plus the
Umeyama-style camera alignment demo.py:127-129 takes from pytorch3d -- so that demo.py / test.py style evaluation needs
neither pytorch3d nor a host round trip.  Same function names, argument meaning and return types as the reference."""
import ctypes as C

import numpy as np
import torch

from posediffusion_amd import _lib
from posediffusion_amd.compat import PerspectiveCameras


def _dev_f32(t, device=None):
    t = torch.as_tensor(t)
    if device is not None:
        t = t.to(device)
    if t.device.type != "cuda":
        raise RuntimeError("posediffusion_amd metrics run only on an AMD GPU; got tensors on " + str(t.device))
    return t.to(torch.float32).contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def batched_all_pairs(B, N):
    """metric.py:106-111 (index helper, kept for callers that use it directly)."""
    i1_, i2_ = torch.combinations(torch.arange(N), 2, with_replacement=False).unbind(-1)
    i1, i2 = [(i[None] + torch.arange(B)[:, None] * N).reshape(-1) for i in [i1_, i2_]]
    return i1, i2


@torch.no_grad()
def camera_to_rel_deg(pred_cameras, gt_cameras, device, batch_size):
    """metric.py:14-47 -> (rel_rangle_deg, rel_tangle_deg), each [batch_size * C(N, 2)] on `device`."""
    Rp, Tp = _dev_f32(pred_cameras.R, device), _dev_f32(pred_cameras.T, device)
    Rg, Tg = _dev_f32(gt_cameras.R, device), _dev_f32(gt_cameras.T, device)
    n = Rg.shape[0] // batch_size
    if Rp.shape != Rg.shape or Rg.shape[0] != batch_size * n or n < 2:
        raise ValueError(f"camera_to_rel_deg: {tuple(Rp.shape)} / {tuple(Rg.shape)} cameras for batch_size {batch_size}")
    total = batch_size * (n * (n - 1) // 2)
    r = torch.empty(total, device=Rg.device, dtype=torch.float32)
    t = torch.empty_like(r)
    _lib.check(_lib.load().pd_metrics_rel_pose_errors(Rp.data_ptr(), Tp.data_ptr(), Rg.data_ptr(), Tg.data_ptr(), batch_size, n,
                                                      r.data_ptr(), t.data_ptr(), _stream(Rg.device)), "pd_metrics_rel_pose_errors")
    return r, t


def rotation_angle(rot_gt, rot_pred, batch_size=None):
    """metric.py:143-151 for explicit relative rotations [n,3,3]: so3_relative_angle in degrees.  (Identity second
    cameras turn the pair kernel into exactly this: R1 = I, T = 0 -> relative rotation = R2.)"""
    rot_gt, rot_pred = _dev_f32(rot_gt), _dev_f32(rot_pred)
    n = rot_gt.shape[0]
    eye = torch.eye(3, device=rot_gt.device).expand(n, 3, 3)
    zero = torch.zeros(n, 3, device=rot_gt.device)
    Rg = torch.stack([eye, rot_gt], dim=1).reshape(2 * n, 3, 3).contiguous()      # n "sequences" of 2 cameras
    Rp = torch.stack([eye, rot_pred], dim=1).reshape(2 * n, 3, 3).contiguous()
    Tz = torch.stack([zero, zero], dim=1).reshape(2 * n, 3).contiguous()
    r = torch.empty(n, device=rot_gt.device)
    t = torch.empty_like(r)
    _lib.check(_lib.load().pd_metrics_rel_pose_errors(Rp.data_ptr(), Tz.data_ptr(), Rg.data_ptr(), Tz.data_ptr(), n, 2,
                                                      r.data_ptr(), t.data_ptr(), _stream(rot_gt.device)), "pd_metrics_rel_pose_errors")
    return r.reshape(batch_size, -1) if batch_size is not None else r


def metrics_summary(rel_rangle_deg, rel_tangle_deg, max_threshold=30):
    """{Auc_<max_threshold>, Racc_5/15/30, Tacc_5/15/30} as test.py:113-121 computes them, in one kernel."""
    r, t = _dev_f32(rel_rangle_deg), _dev_f32(rel_tangle_deg)
    out = torch.empty(7, device=r.device)
    _lib.check(_lib.load().pd_metrics_summary(r.data_ptr(), t.data_ptr(), r.numel(), int(max_threshold), out.data_ptr(),
                                              _stream(r.device)), "pd_metrics_summary")
    v = out.cpu().tolist()
    keys = [f"Auc_{max_threshold}", "Racc_5", "Racc_15", "Racc_30", "Tacc_5", "Tacc_15", "Tacc_30"]
    return dict(zip(keys, v))


def calculate_auc_np(r_error, t_error, max_threshold=30):
    """metric.py:50-78 (numpy in, float out), evaluated on the device."""
    dev = torch.device("cuda", torch.cuda.current_device())
    return metrics_summary(torch.as_tensor(np.asarray(r_error)).to(dev), torch.as_tensor(np.asarray(t_error)).to(dev),
                           max_threshold)[f"Auc_{max_threshold}"]


def calculate_auc(r_error, t_error, max_threshold=30):
    """metric.py:81-103 uses torch.histc with max_threshold + 1 bins over [0, max_threshold]; kept as the reference has it."""
    max_errors, _ = torch.max(torch.stack((r_error, t_error), dim=1), dim=1)
    histogram = torch.histc(max_errors, bins=max_threshold + 1, min=0, max=max_threshold)
    return torch.cumsum(histogram / float(max_errors.size(0)), dim=0).mean()


def compute_ARE(rotation1, rotation2):
    """metric.py:174-185 -> numpy array of degrees."""
    dev = rotation1.device if isinstance(rotation1, torch.Tensor) and rotation1.is_cuda else torch.device("cuda", torch.cuda.current_device())
    Ra, Rb = _dev_f32(rotation1, dev), _dev_f32(rotation2, dev)
    err = torch.empty(Ra.shape[0], device=dev)
    _lib.check(_lib.load().pd_metrics_are(Ra.data_ptr(), Rb.data_ptr(), Ra.shape[0], err.data_ptr(), _stream(dev)), "pd_metrics_are")
    return err.cpu().numpy()


@torch.no_grad()
def corresponding_cameras_alignment(cameras_src, cameras_tgt, estimate_scale=True, mode="extrinsics", eps=1e-9):
    """pytorch3d.ops.corresponding_cameras_alignment as demo.py:127-129 calls it -> aligned copy of cameras_src."""
    if mode != "extrinsics":
        raise ValueError(f"corresponding_cameras_alignment: mode {mode!r} is not implemented (demo.py uses 'extrinsics')")
    Rs, Ts = _dev_f32(cameras_src.R), _dev_f32(cameras_src.T)
    Rt, Tt = _dev_f32(cameras_tgt.R, Rs.device), _dev_f32(cameras_tgt.T, Rs.device)
    if Rs.shape != Rt.shape:
        raise ValueError("cameras_src and cameras_tgt need to contain the same number of cameras!")
    Ro, To = torch.empty_like(Rs), torch.empty_like(Ts)
    _lib.check(_lib.load().pd_align_cameras(Rs.data_ptr(), Ts.data_ptr(), Rt.data_ptr(), Tt.data_ptr(), Rs.shape[0],
                                            int(bool(estimate_scale)), C.c_float(eps), Ro.data_ptr(), To.data_ptr(), None,
                                            _stream(Rs.device)), "pd_align_cameras")
    return PerspectiveCameras(focal_length=cameras_src.focal_length, R=Ro, T=To, device=Rs.device)
