"""ctypes binding of libpd_engine.so (the C-ABI declared in include/pd_engine.h).

The product path has NO fallback: if the shared library is missing or cannot be loaded this
module raises, and every engine entry point raises with it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# PD_ENGINE_LIB: another build of the same C-ABI (same-box A / B of two libraries: tools/ab_ggs.py, bench.py); default = the in-tree build
LIB_PATH = os.environ.get("PD_ENGINE_LIB") or os.path.join(_HERE, "lib", "libpd_engine.so")
CSRC_DIR = os.path.join(_HERE, "csrc")
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "pd_engine.h"))

PD_MAX_LAYERS = 16

c_float_p = C.POINTER(C.c_float)


class pd_layer_weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
        "norm2_w", "norm2_b", "linear1_w", "linear1_b", "linear2_w", "linear2_b")]


class pd_weights(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("d_model", "nhead", "dim_ff", "num_layers", "z_dim", "n_harmonic",
                                  "t_emb_dim", "mlp_hidden", "timesteps", "reserved")]
        + [(n, C.c_void_p) for n in ("time_w0", "time_b0", "time_w2", "time_b2", "first_w", "first_b")]
        + [("layers", pd_layer_weights * PD_MAX_LAYERS)]
        + [(n, C.c_void_p) for n in ("last0_w", "last0_b", "last_ln_w", "last_ln_b", "last3_w", "last3_b",
                                     "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                                     "posterior_mean_coef1", "posterior_mean_coef2",
                                     "posterior_log_variance_clipped")]
    )


class pd_vit_layer_weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b",
                                          "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class pd_vit_weights(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("dim", "depth", "num_heads", "mlp_hidden", "patch_size", "pos_grid", "reserved0", "reserved1")]
                + [(n, C.c_void_p) for n in ("patch_w", "patch_b", "cls_token", "pos_embed", "norm_w", "norm_b")]
                + [("layers", pd_vit_layer_weights * 16)])


# pd_ggs_cfg.reserved flags and pd_engine_set_option ids (include/pd_engine.h; tests/test_host_cpu.py checks them against the header)
PD_GGS_CFG_FORCE_ONE_HOP = 1
PD_GGS_CFG_NO_LDS_STAGING = 2
PD_GGS_CFG_WAVES8 = 4
PD_WEIGHTS_PRED_X0 = 1
PD_GGS_CFG_LANE_ITEMS = 8       # lane-per-item kernel (the throughput shape) whatever the batch size
PD_GGS_CFG_NO_LANE_ITEMS = 16   # never the lane-per-item kernel
PD_GGS_CFG_XCHG_SPREAD = 32     # k > 1: no XCD-local placement of a sequence's workgroups (comparison)
PD_OPT_DENOISER_SPLIT = 2
PD_OPT_WEIGHTS_NON_FINITE = 4   # pd_engine_get_option only
PD_OPT_DENOISER_FUSED_ATTN = 5  # in_proj + attention as one kernel, Q / K / V in LDS (default 1)
PD_MATCH_HINT_ONE_ORDER = 1 << 30   # pd_match_hints.max_pairs flag: every frame pair in one order only (hloc's i < j pairs)


class pd_ggs_cfg(C.Structure):
    _fields_ = [("alpha", C.c_float), ("learning_rate", C.c_float), ("iter_num", C.c_int32),
                ("sampson_max", C.c_float), ("min_matches", C.c_int32), ("momentum", C.c_float),
                ("wgs_per_seq", C.c_int32), ("reserved", C.c_int32)]


class pd_match_hints(C.Structure):
    _fields_ = [("max_pairs", C.c_int32), ("max_matches_per_pair", C.c_int32)]


# name -> (restype, argtypes); kept in one table so the "exports every declared symbol" test and
# the binding cannot drift apart.
_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
SIGNATURES = {
    "pd_engine_create": (_i, [C.POINTER(pd_weights), _i, _i, C.POINTER(_vp)]),
    "pd_engine_destroy": (None, [_vp]),
    "pd_last_error": (C.c_char_p, []),
    "pd_version": (C.c_char_p, []),
    "pd_denoise_step": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "pd_p_mean": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "pd_p_finish": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "pd_ggs_set_matches": (_i, [_vp, _i, _vp, _vp, _vp, _i64, _i, _i, _i]),
    "pd_ggs_set_matches_csr_async": (_i, [_vp, _i, _i, C.POINTER(_i64), _vp, _vp, _vp, _i, _i, _i, C.POINTER(pd_match_hints), _vp]),
    "pd_ggs_guide": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(pd_ggs_cfg), _vp, _vp]),
    "pd_ggs_optimize": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.POINTER(pd_ggs_cfg), _vp, _vp, _vp]),
    "pd_ggs_loss_grad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.POINTER(pd_ggs_cfg), _vp, _vp, _vp]),
    "pd_engine_set_option": (_i, [_vp, _i, _i]),
    "pd_engine_get_option": (_i, [_vp, _i, C.POINTER(_i)]),
    "pd_time_embedding": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "pd_pose_embedding": (_i, [_vp, C.c_longlong, _i, _vp, _vp]),
    "pd_metrics_rel_pose_errors": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "pd_metrics_summary": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "pd_metrics_are": (_i, [_vp, _vp, _i, _vp, _vp]),
    "pd_vit_create": (_i, [C.POINTER(pd_vit_weights), C.POINTER(_vp)]),
    "pd_vit_destroy": (None, [_vp]),
    "pd_vit_set_option": (_i, [_vp, _i, _i]),
    "pd_vit_forward_scale": (_i, [_vp, _vp, _i, _i, _i, C.c_double, _vp, C.c_float, _i, _vp, _vp]),
    "pd_preprocess_image": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "pd_align_cameras": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_float, _vp, _vp, _vp, _vp]),
    "pd_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(pd_ggs_cfg), _vp, _vp, _vp, _i, _vp]),
    "pd_sample_phase": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(pd_ggs_cfg), _i, _vp, _vp, _vp, _i, _vp]),
    "pd_pose_to_camera": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pd_debug_lane_tables": (_i, [_vp, _i, C.POINTER(C.c_int), _i]),
    "pd_ggs_launch_stamps": (_i, [_vp, _vp, _i, C.POINTER(C.c_int), _vp]),
    "pd_ggs_stage_iters": (_i, [C.POINTER(pd_ggs_cfg), C.POINTER(C.c_int)]),
    "pd_pose_to_camera_ex": (_i, [_vp, _vp, _i, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, _vp]),
    "pd_time_kernel": (_i, [_vp, _i, _i, _i, C.POINTER(pd_ggs_cfg), _i, C.POINTER(C.c_float), _vp]),
    "pd_check_async_error": (_i, [_vp]),
    "pd_debug_ggs_prof": (_i, [_vp, _i, C.POINTER(C.c_longlong)]),
    "pd_debug_ggs_plan": (_i, [_vp, _i, _i, C.POINTER(pd_ggs_cfg), C.POINTER(C.c_int)]),
    "pd_debug_mfma_f16_subnormal": (_i, [C.POINTER(C.c_float), _vp]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libpd_engine.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j4"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0 or not os.path.isfile(LIB_PATH):
        raise RuntimeError("building libpd_engine.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def load():
    """dlopen the engine.  torch is imported first so that libamdhip64.so.7 resolves to the copy
    PyTorch-ROCm already loaded (one HIP runtime per process: device pointers and streams are
    shared with torch)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"posediffusion_amd: HIP engine library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C posediffusion_amd/csrc`). "
            "There is no CPU fallback.")
    import torch  # noqa: F401  (loads the HIP runtime first)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("PD_ENGINE_LIB") and not hasattr(lib, name):
            continue              # an older build under A / B lacks the newest debug exports
        fn = getattr(lib, name)   # AttributeError here = library does not match the header
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return (load().pd_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "pd_engine"):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
