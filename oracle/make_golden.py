"""Generate tests/golden/*.npz by executing the UNMODIFIED reference files (build container only).

    python -m oracle.make_golden            # needs /root/reference

TEST INFRASTRUCTURE ONLY.  The reference ships no golden vectors, so these
fixtures ARE the pin: every output below comes from the reference's own Python
(models/gaussian_diffuser.py, models/denoiser.py, util/geometry_guided_sampling.py,
util/get_fundamental_matrix.py, util/camera_transform.py) run on CPU through oracle/ref_stubs.py.
Weights are not stored (69 MB): they are regenerated from the seed protocol
(posediffusion_amd.synth.make_diffuser + randomize_norm_and_bias_) and guarded by a checksum.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pd_oracle as O  # noqa: E402
from oracle import ref_stubs as RS  # noqa: E402
from posediffusion_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
GGS_CFG = dict(synth.GGS_CFG)


def weight_checksum(sd) -> np.ndarray:
    keys = ["_first.weight", "_trunk.layers.0.self_attn.in_proj_weight", "_trunk.layers.7.linear2.weight", "_last.3.weight",
            "_trunk.layers.3.norm1.bias", "time_embed.linear.2.bias"]
    return np.array([float(sd[k].double().abs().sum()) for k in keys])


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def proc_matches(pm, H, W):
    return {"kp1_homo": pm["kp1_homo"], "kp2_homo": pm["kp2_homo"], "i1": pm["i1"], "i2": pm["i2"], "h": H, "w": W,
            "pair_idx": pm["pair_idx"]}


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)                      # bit-reproducible reference runs
    ref = RS.load_reference()
    diff = RS.build_reference_diffuser(seed=0)
    den = diff.model
    synth.randomize_norm_and_bias_(den)
    sd = den.state_dict()

    # ---- (1) schedule tables ---------------------------------------------------------------
    np.savez(os.path.join(OUT, "tables.npz"), **{n: getattr(diff, n).numpy() for n in O.TABLE_NAMES})

    # ---- (2) Denoiser.forward, (3) p_sample -------------------------------------------------
    g = torch.Generator().manual_seed(11)
    cases = {}
    for name, (B, N) in {"b2n20": (2, 20), "b1n7": (1, 7), "b3n33": (3, 33)}.items():
        x = torch.randn(B, N, 9, generator=g)
        z = synth.make_z(B, N, seed=1000)
        cases[f"{name}_x"], cases[f"{name}_z"] = x.numpy(), z.numpy()
        for t in (99, 50, 0):
            with torch.no_grad():
                cases[f"{name}_eps_t{t}"] = den(x, torch.full((B,), t, dtype=torch.long), z).numpy()
    B, N = 2, 20
    x, z = torch.from_numpy(cases["b2n20_x"]), torch.from_numpy(cases["b2n20_z"])
    for t in (99, 50, 10, 9, 0):
        torch.manual_seed(100 + t)
        with torch.no_grad():
            pred, x0 = diff.p_sample(x, t, z)
        torch.manual_seed(100 + t)
        noise = torch.randn_like(x) if t > 0 else torch.zeros_like(x)
        cases[f"ps_noise_t{t}"], cases[f"ps_pred_t{t}"], cases[f"ps_x0_t{t}"] = noise.numpy(), pred.numpy(), x0.numpy()
    cases["weight_checksum"] = weight_checksum(sd)
    np.savez(os.path.join(OUT, "denoiser.npz"), **cases)

    # ---- (4)-(7) GGS ----------------------------------------------------------------------
    N, H, W = 8, 224, 224
    enc = synth.make_cameras(N, seed=2000)
    md = synth.make_matches(enc, H, W, per_pair=60, seed=2000)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    pmr = proc_matches(pm, H, W)
    x0 = synth.perturb_pose(enc, seed=7)
    gg = {"enc": enc, "kp1": md["kp1"], "kp2": md["kp2"], "i12": md["i12"], "img_shape": np.array(md["img_shape"]), "x0": x0.numpy()}
    flag_sets = {"all": (True, True, True), "fl": (False, False, True), "r": (True, False, False), "t": (False, True, False)}
    for fname, (uR, uT, uF) in flag_sets.items():
        for smax in (10, 0.3):                       # 0.3 straddles the Sampson distribution (7)
            xr = x0.clone().requires_grad_(True)
            v, pr = ref.compute_sampson_distance(xr, 0, pmr, update_R=uR, update_T=uT, update_FL=uF, sampson_max=smax)
            loss = v.mean()
            (gr,) = torch.autograd.grad(loss, xr)
            tag = f"sam_{fname}_max{smax}"
            gg[tag + "_loss"], gg[tag + "_nvalid"], gg[tag + "_print"], gg[tag + "_grad"] = \
                loss.item(), len(v), pr.item(), gr.numpy()
    # focal clamp edges (7): some frames above 20, some below 0.1, some exactly inside
    xc = x0.clone()
    xc[0, 0, 7:9] = 2.0          # exp(3.8) = 44.7 -> clamped to 20
    xc[0, 1, 7:9] = -5.0         # exp(-3.2) = 0.04 -> clamped to 0.1
    xc[0, 2, 7] = 1.1957         # exp(2.9957) ~ 20.0 (just inside/outside in fp32)
    xr = xc.clone().requires_grad_(True)
    v, pr = ref.compute_sampson_distance(xr, 0, pmr)
    (gr,) = torch.autograd.grad(v.mean(), xr)
    gg["clamp_x"], gg["clamp_loss"], gg["clamp_nvalid"], gg["clamp_grad"] = xc.numpy(), v.mean().item(), len(v), gr.numpy()
    # (5) k iterations of GGS_optimize, clip/momentum state included in the result
    for fname, (uR, uT, uF) in {"all": (True, True, True), "fl": (False, False, True), "r": (True, False, False)}.items():
        for k in (1, 5, 20):
            cfg = dict(GGS_CFG, iter_num=k)
            xo = quiet(ref.GGS_optimize, x0.clone(), 0, pmr, update_R=uR, update_T=uT, update_FL=uF, **cfg)
            gg[f"opt_{fname}_k{k}"] = xo.numpy()
    # full five-stage call
    cfg = dict(GGS_CFG, iter_num=10)
    gg["guide_k10"] = quiet(ref.geometry_guided_sampling, x0.clone(), 3, md, cfg).numpy()
    # (6) early exit: nearly all matches are outliers -> valid/N < min_matches at the first iteration
    md_bad = synth.make_matches(enc, H, W, per_pair=60, outlier_frac=1.0, noise_px=30.0, seed=5)
    gg["bad_kp1"], gg["bad_kp2"], gg["bad_i12"] = md_bad["kp1"], md_bad["kp2"], md_bad["i12"]
    pmb = O.prepare_matches(md_bad["kp1"], md_bad["kp2"], md_bad["i12"], md_bad["img_shape"])
    vb, _ = ref.compute_sampson_distance(x0.clone(), 0, proc_matches(pmb, H, W), sampson_max=0.01)
    gg["bad_nvalid_max0.01"] = len(vb)
    cfgb = dict(GGS_CFG, iter_num=10, sampson_max=0.01)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        xb = ref.geometry_guided_sampling(x0.clone(), 3, md_bad, cfgb)
    gg["bad_out"], gg["bad_dropped"] = xb.numpy(), buf.getvalue().count("Drop this pair")
    np.savez(os.path.join(OUT, "ggs.npz"), **gg)

    # ---- (8) full 100-step GGS-off trajectory (fp32 reference, fp64 oracle) -----------------
    B, N = 1, 20
    z = synth.make_z(B, N, seed=1000)
    torch.manual_seed(0)
    with torch.no_grad():
        pose, process = diff.sample([B, N, 9], z)
    gen = torch.Generator().manual_seed(0)
    init, noises = O.draw_reference_noise((B, N, 9), gen)
    assert torch.equal(init, process[0]), "RNG replay does not match the reference's draw order"
    noise = np.zeros((101, B, N, 9), dtype=np.float32)
    noise[0] = init.numpy()
    for step in range(100):
        if noises[99 - step] is not None:
            noise[step + 1] = noises[99 - step].numpy()
    sd64 = O.cast_state_dict(sd, torch.float64)
    t64 = O.diffusion_tables(dtype=torch.float64)
    with torch.no_grad():
        p64, proc64 = O.p_sample_loop(sd64, t64, z.double(), init.double(), [None if n is None else n.double() for n in noises])
    np.savez(os.path.join(OUT, "trajectory.npz"), z=z.numpy(), noise=noise, process=process.numpy(), pose=pose.numpy(),
             process64=proc64.numpy())

    # ---- end-to-end sample() with the GGS plug-in (short) -----------------------------------
    N = 8
    z8 = synth.make_z(1, N, seed=1000)
    cfg = dict(GGS_CFG, iter_num=3)
    cond = partial(ref.geometry_guided_sampling, matches_dict=md, GGS_cfg=cfg)
    torch.manual_seed(0)
    with torch.no_grad():
        pose_g, process_g = quiet(diff.sample, [1, N, 9], z8, cond_fn=cond, cond_start_step=3)
    gen = torch.Generator().manual_seed(0)
    init, noises = O.draw_reference_noise((1, N, 9), gen, cond_start_step=3, has_cond=True)
    noise = np.zeros((101, 1, N, 9), dtype=np.float32)
    noise[0] = init.numpy()
    for step in range(100):
        if noises[99 - step] is not None:
            noise[step + 1] = noises[99 - step].numpy()
    np.savez(os.path.join(OUT, "guided.npz"), z=z8.numpy(), noise=noise, process=process_g.numpy(), pose=pose_g.numpy(),
             iter_num=3, cond_start_step=3)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def make_metrics():
    """tests/golden/metrics.npz: the reference's util/metric.py (executed in place) on seeded cameras: a prediction that
    is a perturbed, similarity-transformed copy of the ground truth; B = 2 sequences x N = 6 frames."""
    rm = RS.load_reference_metric()
    B, N = 2, 6
    enc = np.concatenate([synth.make_cameras(N, seed=900 + b) for b in range(B)])
    gt = O.pose_encoding_to_camera(torch.from_numpy(enc).float())
    pr = O.pose_encoding_to_camera(torch.cat([synth.perturb_pose(enc[b * N:(b + 1) * N], seed=910 + b, sigma_T=0.08, sigma_q=0.05)
                                              for b in range(B)], dim=1)[0])
    # edge cases: an exact copy of one pair (rotation angle ~ 0 -> the linear extrapolation branch), a zero translation
    pr["R"][1], pr["T"][1] = gt["R"][1].clone(), gt["T"][1].clone()
    pr["R"][0], pr["T"][0] = gt["R"][0].clone(), gt["T"][0].clone()
    r, t = rm.camera_to_rel_deg(rm.Cameras(R=pr["R"], T=pr["T"], focal_length=pr["focal_length"]),
                                rm.Cameras(R=gt["R"], T=gt["T"], focal_length=gt["focal_length"]), torch.device("cpu"), B)
    auc = rm.calculate_auc_np(r.numpy(), t.numpy(), max_threshold=30)
    are = rm.compute_ARE(pr["R"], gt["R"])
    np.savez(os.path.join(OUT, "metrics.npz"), B=B, N=N, R_pred=pr["R"].numpy(), T_pred=pr["T"].numpy(), R_gt=gt["R"].numpy(),
             T_gt=gt["T"].numpy(), rel_r_deg=r.numpy(), rel_t_deg=t.numpy(), auc30=auc, are_deg=are)
    print("metrics.npz", os.path.getsize(os.path.join(OUT, "metrics.npz")))



def make_preprocess():
    """tests/golden/preprocess.npz (+ tests/golden/images/*.png, a few KB): the reference's util/load_img_folder.py
    executed in place on three small synthetic images (portrait, landscape, square; lossless PNG) at image_size 32 and
    17; plus colmap_keypoint_to_pytorch3d of util/match_extraction.py on synthetic COLMAP-style inputs (that function is
    plain numpy; the module's hloc / pycolmap imports are stubbed for the import only)."""
    import importlib.util
    import types
    from PIL import Image
    img_dir = os.path.join(OUT, "images")
    os.makedirs(img_dir, exist_ok=True)
    rng = np.random.default_rng(77)
    for k, (h, w) in enumerate(((61, 97), (120, 80), (64, 64))):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 3) % 256], -1)
        im = np.clip(base + rng.integers(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(im, "RGB").save(os.path.join(img_dir, f"frame{k:02d}.png"))
    spec = importlib.util.spec_from_file_location("_ref_load_img", os.path.join(RS.REF_PKG, "util", "load_img_folder.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = {}
    for S in (32, 17):
        imgs, info = m.load_and_preprocess_images(img_dir, S)
        out[f"images_{S}"] = imgs.numpy()
        out[f"bboxes_{S}"] = info["bboxes_xyxy"]
        out[f"scales_{S}"] = info["resized_scales"]
        out[f"size_{S}"] = np.array(info["size"])
    # keypoint bookkeeping
    saved = {k: sys.modules.get(k) for k in ("pycolmap", "hloc", "hloc.triangulation", "hloc.utils", "hloc.utils.database", "hloc.reconstruction")}
    class _Anything:                      # whatever the module body asks of the absent packages at import time
        def __getattr__(self, n):
            return _Anything()

        def __call__(self, *a, **k):
            return _Anything()

    anyattr = type("Any", (types.ModuleType,), {"__getattr__": lambda self, n: _Anything()})
    for k in saved:
        sys.modules[k] = anyattr(k)
    try:
        spec = importlib.util.spec_from_file_location("_ref_match", os.path.join(RS.REF_PKG, "util", "match_extraction.py"))
        mm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mm)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    keypoints = {i + 1: rng.uniform(0, 60, size=(12, 2)) for i in range(3)}
    matches = {(1, 2): rng.integers(0, 12, size=(7, 2)), (1, 3): None, (2, 3): rng.integers(0, 12, size=(4, 2))}
    info = {"bboxes_xyxy": out["bboxes_32"], "resized_scales": out["scales_32"]}
    kp1, kp2, i12 = mm.colmap_keypoint_to_pytorch3d({k: v.copy() if v is not None else None for k, v in matches.items()},
                                                    {k: v.copy() for k, v in keypoints.items()}, info)
    for i in range(3):
        out[f"colmap_kp_{i + 1}"] = keypoints[i + 1]
    out["colmap_m_12"], out["colmap_m_23"] = matches[(1, 2)], matches[(2, 3)]
    out["kp1"], out["kp2"], out["i12"] = kp1, kp2, i12
    np.savez(os.path.join(OUT, "preprocess.npz"), **out)
    print("preprocess.npz", os.path.getsize(os.path.join(OUT, "preprocess.npz")))


def make_guided_free(seeds=(0, 1, 2), N=8, cond_start=3, per_pair=60, name="guided_free", store_matches=True):
    """tests/golden/guided_free.npz -- SURVEY.md section 8c's FREE-RUNNING GGS-on criterion, scaled down from BASELINE
    configs[2]: N = 8 frames, 28 pairs x 60 matches, 100 DDPM steps, the last `cond_start` = 3 of them guided with the
    FULL default schedule (5 optimisations = 700 iterations per guided step, cfgs/default.yaml:6-13).  Per seed:
      * the UNMODIFIED reference (models/gaussian_diffuser.py sample() + util/geometry_guided_sampling.py, fp32, CPU);
      * the fp64 oracle on the same z / noise / matches;
    and each one's final mean Sampson error over its valid matches (evaluated in fp64 at the final pose).
    With random-init weights the sampled poses are arbitrary, so -- as bench.py does -- the matches are synthesised
    to be epipolar-consistent (+0.5 px noise, 10 % outliers) with the fp32 model mean at the first guided step, which
    puts the guided steps in the basin the trained model + SuperGlue matches would give (all 2 100 iterations run).

    `make_guided_free(seeds=(0,), N=20, cond_start=10, per_pair=300, name="guided_free_full")` writes
    tests/golden/guided_free_full.npz: ONE seed of BASELINE configs[2] at its real size -- 20 frames, 190 pairs x 300 =
    57 000 matches, 224^2, 100 steps, the last 10 guided x 700 iterations = 7 000 iterations.
    (about 20 CPU-minutes: the reference and the fp64 oracle each run 7 000 iterations over 57 000 matches on one thread).

    `store_matches=False` (round 5: `guided_free_full_s12`, seeds 1 and 2 of the same full-size case) leaves the 57 000 matches out of
    the file: they are a pure function of the stored `mean_at_first_guided` and the seed (`synth.make_epipolar_matches(mean, 224, 224,
    per_pair, seed=2000 + s)`, numpy's `default_rng`), so the fixture stores their sha256 instead and `regenerate_matches` rebuilds them."""
    torch.set_num_threads(1)
    ref = RS.load_reference()
    diff = RS.build_reference_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    sd = diff.model.state_dict()
    sd32, sd64 = O.cast_state_dict(sd, torch.float32), O.cast_state_dict(sd, torch.float64)
    t32, t64 = O.diffusion_tables(), O.diffusion_tables(dtype=torch.float64)
    cfg = dict(GGS_CFG)                                      # iter_num 100 -> 200/100/100/100/200
    out = {"seeds": np.array(seeds), "cond_start_step": cond_start, "weight_checksum": weight_checksum(sd)}
    T = 100
    for s in seeds:
        z = synth.make_z(1, N, seed=1000 + s)
        init, noises = O.draw_reference_noise((1, N, 9), torch.Generator().manual_seed(s), cond_start_step=cond_start, has_cond=True)
        # the model mean GGS first sees (t = cond_start - 1), from the fp32 restatement of the unguided prefix
        with torch.no_grad():
            x = init
            for t in reversed(range(cond_start, T)):
                x, _ = O.p_sample(sd32, t32, x, t, z, noises[t])
            mean = O.p_mean_variance(sd32, t32, x, cond_start - 1, z)[0]
        md = synth.make_epipolar_matches(mean[0].numpy().astype(np.float64), 224, 224, per_pair, seed=2000 + s)
        pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        # (a) the reference itself, fp32
        buf = io.StringIO()
        torch.manual_seed(s)
        with torch.no_grad(), contextlib.redirect_stdout(buf):
            pose32, process32 = diff.sample([1, N, 9], z, cond_fn=partial(ref.geometry_guided_sampling, matches_dict=md, GGS_cfg=cfg),
                                            cond_start_step=cond_start)
        assert torch.equal(process32[0], init), "RNG replay does not match the reference's draw order"
        n_it = buf.getvalue().count("sampson=")                 # one print per GGS_optimize call (:124)
        assert "Drop this pair" not in buf.getvalue(), "an optimisation exited early: the fixture must run every iteration"
        # (b) the fp64 oracle
        with torch.no_grad():
            pass
        cond64 = lambda m, t: O.geometry_guided_sampling(m, t, md, cfg)       # noqa: E731
        pose64, process64 = O.p_sample_loop(sd64, t64, z.double(), init.double(), [None if n is None else n.double() for n in noises],
                                            cond_fn=cond64, cond_start_step=cond_start)
        sam = {}
        for tag, pose in (("32", pose32), ("64", pose64)):
            v, _ = O.compute_sampson_distance(pose.detach().double(), pm)
            sam[tag] = (float(v.mean()), len(v))
        noise = np.zeros((T + 1, 1, N, 9), dtype=np.float32)
        noise[0] = init.numpy()
        for step in range(T):
            if noises[T - 1 - step] is not None:
                noise[step + 1] = noises[T - 1 - step].numpy()
        dev = float(((pose32.double() - pose64).abs().max() / pose64.abs().max()))
        print(f"seed {s}: reference ran {n_it} GGS_optimize calls to completion; |pose| max {float(pose64.abs().max()):.2f}; ref32 vs fp64 {dev:.3e}; "
              f"final mean Sampson ref32 {sam['32'][0]:.6f} ({sam['32'][1]} valid), fp64 {sam['64'][0]:.6f} ({sam['64'][1]} valid)")
        if store_matches:
            out.update({f"s{s}_kp1": md["kp1"], f"s{s}_kp2": md["kp2"], f"s{s}_i12": md["i12"]})
        else:
            out[f"s{s}_matches_sha256"] = np.frombuffer(matches_digest(md), dtype=np.uint8)
        out.update({f"s{s}_z": z.numpy(), f"s{s}_noise": noise,
                    f"s{s}_pose32": pose32.numpy(), f"s{s}_pose64": pose64.detach().numpy(), f"s{s}_mean_at_first_guided": mean.numpy(),
                    f"s{s}_sampson32": np.array(sam["32"]), f"s{s}_sampson64": np.array(sam["64"]), f"s{s}_ref_optimize_calls": n_it})
    out["img_shape"] = np.array([N, 3, 224, 224])
    out["per_pair"] = per_pair
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name + ".npz", os.path.getsize(os.path.join(OUT, name + ".npz")))


def matches_digest(md):
    import hashlib
    h = hashlib.sha256()
    for k in ("kp1", "kp2", "i12"):
        h.update(np.ascontiguousarray(md[k]).tobytes())
    return h.digest()


def regenerate_matches(g, s):
    """The matches of seed `s` of a `store_matches=False` fixture `g` (a dict of its arrays), rebuilt from the stored model mean and
    checked against the stored sha256 (a numpy whose default_rng drew differently would fail here, not silently change the case)."""
    n = int(g["img_shape"][0])
    md = synth.make_epipolar_matches(g[f"s{s}_mean_at_first_guided"][0].astype(np.float64), 224, 224, int(g["per_pair"]), seed=2000 + s)
    assert matches_digest(md) == bytes(g[f"s{s}_matches_sha256"]), "regenerated matches differ from the ones the fixture was made with"
    assert int(md["img_shape"][0]) == n
    return md


def make_pred_x0():
    """tests/golden/pred_x0.npz -- GaussianDiffusion(objective="pred_x0") (models/gaussian_diffuser.py:105-108, :225-227): the
    UNMODIFIED reference's p_sample at five steps (B = 2, N = 20), its model_predictions pair at one step and one whole
    100-step GGS-off trajectory (B = 1, N = 8), all fp32 on the CPU, with the seeded weights of the other fixtures."""
    torch.set_num_threads(1)
    ref = RS.load_reference()
    base = RS.build_reference_diffuser(seed=0)
    synth.randomize_norm_and_bias_(base.model)
    diff = ref.GaussianDiffusion(beta_schedule="custom", objective="pred_x0")
    diff.model = base.model
    diff.eval()
    out = {"weight_checksum": weight_checksum(base.model.state_dict())}
    B, N = 2, 20
    x = torch.randn(B, N, 9, generator=torch.Generator().manual_seed(21))
    z = synth.make_z(B, N, seed=1000)
    out["x"], out["z"] = x.numpy(), z.numpy()
    for t in (99, 50, 10, 1, 0):
        torch.manual_seed(300 + t)
        with torch.no_grad():
            pred, x0 = diff.p_sample(x, t, z)
        torch.manual_seed(300 + t)
        noise = torch.randn_like(x) if t > 0 else torch.zeros_like(x)
        out[f"ps_noise_t{t}"], out[f"ps_pred_t{t}"], out[f"ps_x0_t{t}"] = noise.numpy(), pred.numpy(), x0.numpy()
    with torch.no_grad():
        mp = diff.model_predictions(x, torch.full((B,), 50, dtype=torch.long), z)
    out["mp_noise_t50"], out["mp_x0_t50"] = mp.pred_noise.numpy(), mp.pred_x_start.numpy()
    N = 8
    z8 = synth.make_z(1, N, seed=1001)
    torch.manual_seed(5)
    with torch.no_grad():
        pose, process = diff.sample([1, N, 9], z8)
    init, noises = O.draw_reference_noise((1, N, 9), torch.Generator().manual_seed(5))
    assert torch.equal(init, process[0]), "RNG replay does not match the reference's draw order"
    noise = np.zeros((101, 1, N, 9), dtype=np.float32)
    noise[0] = init.numpy()
    for step in range(100):
        if noises[99 - step] is not None:
            noise[step + 1] = noises[99 - step].numpy()
    sd64 = O.cast_state_dict(base.model.state_dict(), torch.float64)
    with torch.no_grad():
        p64, proc64 = O.p_sample_loop(sd64, O.diffusion_tables(dtype=torch.float64), z8.double(), init.double(),
                                      [None if n is None else n.double() for n in noises], objective="pred_x0")
    print(f"pred_x0 trajectory: |pose| max {float(pose.abs().max()):.3f}; reference fp32 vs fp64 oracle "
          f"{float((process.double() - proc64).abs().max() / proc64.abs().max()):.3e}")
    out.update({"traj_z": z8.numpy(), "traj_noise": noise, "traj_process": process.numpy(), "traj_pose": pose.numpy(),
                "traj_process64": proc64.numpy()})
    np.savez_compressed(os.path.join(OUT, "pred_x0.npz"), **out)
    print("pred_x0.npz", os.path.getsize(os.path.join(OUT, "pred_x0.npz")))


def make_decode_args():
    """tests/golden/decode_args.npz -- the reference's pose_encoding_to_camera (util/camera_transform.py:64-105, executed in place) with
    its three keyword parameters at the defaults and at two other settings, on encodings whose log focal lengths straddle every clamp
    (missing item 4: log_focal_length_bias / min_focal_length / max_focal_length are real parameters)."""
    ref = RS.load_reference()
    N = 12
    enc = torch.from_numpy(synth.make_cameras(N, seed=4242)).float().reshape(1, N, 9).clone()
    enc[0, :, 7] = torch.linspace(-6.0, 3.0, N)          # exp(x + bias) from far below every minimum to far above every maximum
    enc[0, :, 8] = torch.linspace(2.5, -5.5, N)
    out = {"enc": enc.numpy()}
    for tag, (bias, fmin, fmax) in {"default": (1.8, 0.1, 20), "a": (1.2, 0.5, 5.0), "b": (0.0, 0.01, 15.0)}.items():
        d = ref.pose_encoding_to_camera(enc, log_focal_length_bias=bias, min_focal_length=fmin, max_focal_length=fmax, return_dict=True)
        out[f"{tag}_args"] = np.array([bias, fmin, fmax], dtype=np.float64)
        out[f"{tag}_R"], out[f"{tag}_T"], out[f"{tag}_focal"] = d["R"].numpy(), d["T"].numpy(), d["focal_length"].numpy()
    np.savez(os.path.join(OUT, "decode_args.npz"), **out)
    print("decode_args.npz", os.path.getsize(os.path.join(OUT, "decode_args.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "decode_args":
        make_decode_args()      # only the decode-parameter fixture
    elif len(sys.argv) > 1 and sys.argv[1] == "pred_x0":
        make_pred_x0()          # only the objective="pred_x0" fixture
    elif len(sys.argv) > 1 and sys.argv[1] == "guided_free":
        make_guided_free()      # only the free-running GGS-on fixture
    elif len(sys.argv) > 1 and sys.argv[1] == "guided_free_full":
        make_guided_free(seeds=(0,), N=20, cond_start=10, per_pair=300, name="guided_free_full")   # configs[2] at real size
    elif len(sys.argv) > 1 and sys.argv[1] == "guided_free_full_s12":
        make_guided_free(seeds=(1, 2), N=20, cond_start=10, per_pair=300, name="guided_free_full_s12", store_matches=False)
    elif len(sys.argv) > 1 and sys.argv[1] == "metrics":
        make_metrics()          # only the N3 fixture (the others stay byte-identical)
    elif len(sys.argv) > 1 and sys.argv[1] == "preprocess":
        make_preprocess()       # only the N4 fixture
    else:
        main()
        make_metrics()
        make_preprocess()
        make_guided_free()
        make_guided_free(seeds=(0,), N=20, cond_start=10, per_pair=300, name="guided_free_full")
        make_decode_args()
