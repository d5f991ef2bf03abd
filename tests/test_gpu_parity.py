"""GPU (-m gpu): the HIP path through the C-ABI against the CPU oracle and the committed golden vectors.

Tolerance contract (BASELINE.json north_star: 1e-4 relative):
  * teacher-forced -- every engine step / GGS iteration fed the oracle's (or the reference fixture's)
    state must match to <= 1e-4 relative; fp32 kernels are expected at ~1e-6 and asserted at 2e-5;
  * free-running GGS-off trajectories are compared against the fp64 oracle next to the reference-fp32
    trajectory's own deviation (random-init weights make |pose| grow to ~70, so a blanket 1e-4 on the
    final pose is not attainable even by the reference against itself);
  * integer work (valid-match counts, iteration counts, early-exit decisions) is exact.
"""
import functools

import numpy as np
import pytest
import torch

from conftest import camera_errs, pose_err, rel_err
from oracle import pd_oracle as O
from posediffusion_amd import _lib, synth
from posediffusion_amd.engine import make_ggs_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLAGS = {"all": (True, True, True), "fl": (False, False, True), "r": (True, False, False), "t": (False, True, False)}
TOL = 2e-5          # asserted; the contract is 1e-4


def _upload(engine, g, slot=0, prefix=""):
    shape = tuple(int(v) for v in g["img_shape"])
    engine.set_matches(slot, g[prefix + "kp1"], g[prefix + "kp2"], g[prefix + "i12"], shape)
    return {"kp1": g[prefix + "kp1"], "kp2": g[prefix + "kp2"], "i12": g[prefix + "i12"], "img_shape": shape}


# ------------------------------------------------------------------------------------------------ denoiser
@pytest.mark.parametrize("case", ["b2n20", "b1n7", "b3n33"])
def test_denoiser_vs_reference_fixture(engine, golden, case):
    d = golden["denoiser"]
    x, z = torch.from_numpy(d[f"{case}_x"]).to(DEV), torch.from_numpy(d[f"{case}_z"]).to(DEV)
    for t in (99, 50, 0):
        assert rel_err(engine.denoise(x, z, t), d[f"{case}_eps_t{t}"]) < TOL


@pytest.mark.parametrize("B,N", [(1, 1), (1, 3), (8, 20), (1, 50), (5, 13), (2, 33)])
def test_denoiser_vs_oracle_shapes(engine, oracle_weights, B, N):
    """ragged token counts: M = B*N not a multiple of the 32-row MFMA tile, N = 1 (softmax over one key)."""
    g = torch.Generator().manual_seed(B * 1000 + N)
    x, z = torch.randn(B, N, 9, generator=g), synth.make_z(B, N, seed=5)
    t = (7 * B + N) % 100
    ref = O.denoiser_forward(oracle_weights, x, torch.full((B,), t, dtype=torch.long), z)
    assert rel_err(engine.denoise(x.to(DEV), z.to(DEV), t), ref) < TOL


def test_denoiser_wide_tile_path_large_batch(seeded_diffuser, oracle_weights):
    """M = 1 680 tokens: every GEMM has >= 200 32-wide tiles, so the 32x32x2 path is taken wherever it is legal;
    the 128-wide `_last.0` (4 N-tiles of 32, not a multiple of the 8 XCDs) must stay on 16-wide tiles
    (a regression here wrote past the activation buffer)."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N = 84, 20
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    eng = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N)
    g = torch.Generator().manual_seed(77)
    x, z = torch.randn(B, N, 9, generator=g), synth.make_z(B, N, seed=6)
    ref = O.denoiser_forward(oracle_weights, x, torch.full((B,), 42, dtype=torch.long), z)
    assert rel_err(eng.denoise(x.to(dev), z.to(dev), 42), ref) < TOL
    eng.close()


def test_large_pose_values_harmonic_embedding(engine, oracle_weights):
    """|x| ~ 50 puts harmonic arguments at 2.5e4 rad: range reduction of sin must hold (SURVEY 'error amplifiers')."""
    g = torch.Generator().manual_seed(9)
    x, z = 50.0 * torch.randn(2, 20, 9, generator=g), synth.make_z(2, 20)
    ref = O.denoiser_forward(oracle_weights, x, torch.full((2,), 3, dtype=torch.long), z)
    assert rel_err(engine.denoise(x.to(DEV), z.to(DEV), 3), ref) < 1e-4


def test_p_sample_pieces_vs_reference_fixture(engine, golden):
    d = golden["denoiser"]
    x, z = torch.from_numpy(d["b2n20_x"]).to(DEV), torch.from_numpy(d["b2n20_z"]).to(DEV)
    for t in (99, 50, 10, 9, 0):
        mean, x0 = engine.p_mean(x, z, t)
        noise = torch.from_numpy(d[f"ps_noise_t{t}"]).to(DEV) if t > 0 else None
        pred = engine.p_finish(mean, noise, t)
        assert pose_err(x0, d[f"ps_x0_t{t}"], "p_sample_pieces") < TOL              # T / quaternion / logFL columns each (conftest.pose_err)
        assert pose_err(pred, d[f"ps_pred_t{t}"], "p_sample_pieces") < TOL


def test_denoiser_module_api_and_per_sequence_timesteps(engine, seeded_diffuser, oracle_weights):
    """Drop-in Denoiser.forward(x, t[B], z) with different t per sequence."""
    g = torch.Generator().manual_seed(4)
    x, z = torch.randn(3, 6, 9, generator=g), synth.make_z(3, 6)
    t = torch.tensor([5, 77, 5])
    out = seeded_diffuser.model(x.to(DEV), t.to(DEV), z.to(DEV))
    assert rel_err(out, O.denoiser_forward(oracle_weights, x, t, z)) < TOL


# ------------------------------------------------------------------------------------------------ sampler
def test_sampler_teacher_forced_vs_reference_trajectory(engine, golden):
    tr = golden["trajectory"]
    z, noise = torch.from_numpy(tr["z"]).to(DEV), torch.from_numpy(tr["noise"]).to(DEV)
    proc_ref = torch.from_numpy(tr["process"])
    for step in list(range(0, 100, 7)) + [99]:
        t = 99 - step
        mean, _ = engine.p_mean(proc_ref[step].to(DEV), z, t)
        nxt = engine.p_finish(mean, noise[step + 1] if t > 0 else None, t)
        assert pose_err(nxt, proc_ref[step + 1], "sampler_teacher_forced") < TOL


def test_sampler_free_running_vs_fp64_oracle(engine, golden):
    tr = golden["trajectory"]
    z, noise = torch.from_numpy(tr["z"]).to(DEV), torch.from_numpy(tr["noise"]).to(DEV)
    outs = {}
    for use_graph in (False, True):
        pose, process, _ = engine.sample(z, noise, 0, None, use_graph=use_graph)
        outs[use_graph] = process.cpu()
        assert torch.equal(pose.cpu(), outs[use_graph][-1])
    assert torch.equal(outs[False], outs[True]), "hipGraph replay must be bitwise identical to eager launches"
    assert torch.equal(outs[True][0], torch.from_numpy(tr["noise"][0]))
    p64 = tr["process64"]
    ours, ref32 = rel_err(outs[True][-1], p64[-1]), rel_err(tr["process"][-1], p64[-1])
    # free-running to t = 0 with random-init weights is chaotic (|pose| grows to ~70, a 1e-7 per-step rounding
    # difference is amplified ~1000x): the engine's deviation from fp64 is a sample of the same distribution as the
    # reference-fp32's own deviation, not a fixed number -- two builds of this engine that differ only in FMA
    # contraction gave 0.9x and 2.4x of it.  Bound: same order of magnitude (4x, floor 1e-4); the tight checks are
    # the teacher-forced steps above (2e-5) and the bounded prefix below (1e-4).
    assert ours <= max(4.0 * ref32, 1e-4), (ours, ref32)
    # bounded part of the trajectory (first 30 steps, |pose| < 10): blanket tolerance
    assert rel_err(outs[True][30], p64[30]) < 1e-4


def test_guided_sampling_end_to_end_vs_reference_fixture(engine, golden):
    """pd_sample with the GGS plug-in (3 guided steps x 21 iterations) vs the reference's sample()."""
    g, gg = golden["guided"], golden["ggs"]
    _upload(engine, gg)
    z, noise = torch.from_numpy(g["z"]).to(DEV), torch.from_numpy(g["noise"]).to(DEV)
    cfg = dict(synth.GGS_CFG, iter_num=int(g["iter_num"]))
    pose, process, stats = engine.sample(z, noise, int(g["cond_start_step"]), cfg, use_graph=True)
    engine.check_async()
    ref = torch.from_numpy(g["process"])
    # unguided prefix: free-running but still bounded early on
    assert rel_err(process[20], ref[20]) < 1e-4
    # teacher-forced guided step: feed the reference's x_t at t = 2 and compare the guided result
    step = 97
    mean, _ = engine.p_mean(ref[step].to(DEV), z, 2)
    out, st = engine.ggs_guide(mean, 2, cfg)
    assert rel_err(out, ref[step + 1]) < 1e-4
    assert stats.shape == (3, 1, 5, 4) and torch.isfinite(pose).all()


def test_dropin_gaussian_diffusion_sample_api(engine, seeded_diffuser, golden):
    """GaussianDiffusion.sample(shape, z, cond_fn, cond_start_step) with the reference's own cond_fn partial
    and seed protocol (torch.manual_seed before the call)."""
    from util.geometry_guided_sampling import geometry_guided_sampling
    g, gg = golden["guided"], golden["ggs"]
    md = {"kp1": gg["kp1"], "kp2": gg["kp2"], "i12": gg["i12"], "img_shape": torch.Size(int(v) for v in gg["img_shape"])}
    cfg = dict(synth.GGS_CFG, iter_num=int(g["iter_num"]))
    cond = functools.partial(geometry_guided_sampling, matches_dict=md, GGS_cfg=cfg)
    z = torch.from_numpy(g["z"]).to(DEV)
    torch.manual_seed(0)
    pose, process = seeded_diffuser.sample([1, 8, 9], z, cond_fn=cond, cond_start_step=3)
    assert process.shape == (101, 1, 8, 9) and torch.isfinite(pose).all()
    torch.manual_seed(0)
    pose2, _ = seeded_diffuser.sample([1, 8, 9], z, cond_fn=cond, cond_start_step=3)
    assert torch.equal(pose, pose2), "same seed must reproduce bitwise"
    # an unknown guidance callable still runs (reference control flow, HIP arithmetic)
    calls = []

    def my_cond(mean, t):
        calls.append(t)
        return mean * 1.0

    torch.manual_seed(0)
    pose3, _ = seeded_diffuser.sample([1, 8, 9], z, cond_fn=my_cond, cond_start_step=2)
    assert calls == [1, 0] and torch.isfinite(pose3).all()


def test_pose_diffusion_model_forward_api(seeded_diffuser):
    models = synth._dropin()
    from posediffusion_amd.compat import AttrDict
    cfg = {"pose_encoding_type": "absT_quaR_logFL",
           "IMAGE_FEATURE_EXTRACTOR": AttrDict({"_target_": "models.MultiScaleImageFeatureExtractor", "freeze": False}),
           "DENOISER": AttrDict({"_target_": "models.Denoiser", "TRANSFORMER": AttrDict(synth.TRANSFORMER_CFG)}),
           "DIFFUSER": AttrDict({"_target_": "models.GaussianDiffusion", "beta_schedule": "custom"})}
    torch.manual_seed(0)
    model = models.PoseDiffusionModel(**cfg).to(DEV).eval()
    z = synth.make_z(2, 5).to(DEV)
    out = model(image=None, training=False, z=z)
    cams = out["pred_cameras"]
    assert cams.R.shape == (10, 3, 3) and cams.T.shape == (10, 3) and cams.focal_length.shape == (10, 2)
    ref = O.pose_encoding_to_camera(out["pose_encoding"].cpu())
    ce = camera_errs(cams.R, cams.T, cams.focal_length, ref)                        # R / T / focal separately (north_star: each within 1e-4)
    assert ce["R"] < 1e-5 and ce["focal"] < 1e-5 and ce["T"] == 0.0, ce
    assert torch.equal(cams.T.cpu(), ref["T"])
    with pytest.raises(NotImplementedError):
        model(image=None, training=True, z=z)
    # images in: the extractor (csrc/pd_vit.hip) feeds the sampler; same noise -> the same poses as with its z handed in
    img = torch.rand(2, 5, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(DEV)
    z_img = model.image_feature_extractor(img.reshape(10, 3, 224, 224)).reshape(2, 5, -1)
    assert z_img.shape == (2, 5, 384) and torch.isfinite(z_img).all()
    torch.manual_seed(11)
    a = model(image=img, training=False)["pose_encoding"]
    torch.manual_seed(11)
    b = model(image=None, training=False, z=z_img)["pose_encoding"]
    assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ GGS
@pytest.mark.parametrize("fname", list(FLAGS))
@pytest.mark.parametrize("smax", [10, 0.3])
def test_sampson_loss_and_gradient_vs_reference_fixture(engine, golden, fname, smax):
    g = golden["ggs"]
    _upload(engine, g)
    x = torch.from_numpy(g["x0"]).to(DEV)
    tag = f"sam_{fname}_max{smax}"
    for k in (1, 0):
        loss, grad = engine.ggs_loss_grad(x, *FLAGS[fname], cfg=make_ggs_cfg(sampson_max=smax, wgs_per_seq=k))
        engine.check_async()
        assert int(loss[0, 1].item()) == int(g[tag + "_nvalid"])                      # exact count
        assert abs(loss[0, 0].item() - float(g[tag + "_loss"])) < TOL * abs(float(g[tag + "_loss"]))
        assert abs(loss[0, 2].item() - float(g[tag + "_print"])) < TOL * abs(float(g[tag + "_print"]))
        # gradient: 1e-4 of the reference's fp32 autograd -- or, where cancellation in the sum amplifies fp32 rounding (the
        # 0.3 threshold cuts through the bulk of the distribution), no further from the fp64 gradient than twice the
        # reference's own fp32 distance from it
        err = rel_err(grad, g[tag + "_grad"])
        if err >= 1e-4:
            xd = torch.from_numpy(g["x0"]).double().requires_grad_(True)
            pm = O.prepare_matches(g["kp1"], g["kp2"], g["i12"], tuple(int(v) for v in g["img_shape"]))
            v64, _ = O.compute_sampson_distance(xd, pm, *FLAGS[fname], sampson_max=smax)
            (g64,) = torch.autograd.grad(v64.mean(), xd)
            assert len(v64) == int(g[tag + "_nvalid"])
            assert rel_err(grad, g64) <= 2.0 * rel_err(g[tag + "_grad"], g64), (err, rel_err(grad, g64), rel_err(g[tag + "_grad"], g64))
        if fname == "fl":
            assert (grad[0, :, :7] == 0).all()
        if fname == "r":
            assert (grad[0, :, :3] == 0).all() and (grad[0, :, 7:] == 0).all()


def test_focal_clamp_edges_vs_reference_fixture(engine, golden):
    g = golden["ggs"]
    _upload(engine, g)
    loss, grad = engine.ggs_loss_grad(torch.from_numpy(g["clamp_x"]).to(DEV))
    assert int(loss[0, 1].item()) == int(g["clamp_nvalid"])
    assert rel_err(grad, g["clamp_grad"]) < 1e-4
    assert (grad[0, 0, 7:9] == 0).all() and (grad[0, 1, 7:9] == 0).all()


@pytest.mark.parametrize("tag", ["default", "a", "b"])
def test_pose_decode_parameters_vs_reference_fixture(engine, golden, tag):
    """pose_encoding_to_camera(log_focal_length_bias, min_focal_length, max_focal_length) (util/camera_transform.py:64-70, :89-97): the
    reference executed in place at the defaults and at two other settings, log focal lengths on both sides of every clamp --
    through the C-ABI (pd_pose_to_camera_ex) and through the drop-in function; R, T and focal asserted separately."""
    from posediffusion_amd.dropin.util.camera_transform import pose_encoding_to_camera
    g = golden["decode_args"]
    enc = torch.from_numpy(g["enc"]).to(DEV)
    bias, fmin, fmax = (float(v) for v in g[f"{tag}_args"])
    ref = {"R": g[f"{tag}_R"], "T": g[f"{tag}_T"], "focal_length": g[f"{tag}_focal"]}
    R, T, f = engine.pose_to_camera(enc, bias, fmin, fmax)
    ce = camera_errs(R, T, f, ref)
    assert ce["R"] < 1e-6 and ce["T"] == 0.0 and ce["focal"] < 2e-6, ce
    clamped = (ref["focal_length"] == fmin) | (ref["focal_length"] == fmax)
    assert clamped.any() and not clamped.all()
    assert np.array_equal(f.cpu().numpy()[clamped], ref["focal_length"][clamped]), "clamped focal lengths are the bounds themselves"
    d = pose_encoding_to_camera(enc, log_focal_length_bias=bias, min_focal_length=fmin, max_focal_length=fmax, return_dict=True, engine=engine)
    assert torch.equal(d["R"], R) and torch.equal(d["T"], T) and torch.equal(d["focal_length"], f)
    if tag == "default":
        cams = pose_encoding_to_camera(enc, engine=engine)                           # defaults, PerspectiveCameras out
        assert torch.equal(cams.focal_length, f) and torch.equal(cams.R, R)
    nan = enc.clone()
    nan[0, 0, 7] = float("nan")
    assert torch.isnan(engine.pose_to_camera(nan, bias, fmin, fmax)[2][0, 0]), "torch.clamp keeps NaN"


@pytest.mark.parametrize("fname", ["all", "fl", "r"])
@pytest.mark.parametrize("k", [1, 5, 20])
def test_ggs_optimize_iterations_vs_reference_fixture(engine, golden, fname, k):
    g = golden["ggs"]
    _upload(engine, g)
    x0 = torch.from_numpy(g["x0"]).to(DEV)
    outs = []
    for wgs in (1, 0, 3):
        xo, st, _ = engine.ggs_optimize(x0, *FLAGS[fname], cfg=make_ggs_cfg(iter_num=k, wgs_per_seq=wgs))
        engine.check_async()
        assert int(st[0, 1].item()) == (2 * k if fname == "all" else k)
        assert pose_err(xo, g[f"opt_{fname}_k{k}"], f"ggs_optimize_{fname}_k{k}") < TOL
        outs.append(xo.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "results must not depend on workgroups per sequence"


def test_ggs_per_iteration_trace_vs_oracle(engine, golden):
    """Every one of 40 iterations (state after the clipped momentum step) against the oracle's trace."""
    g = golden["ggs"]
    md = _upload(engine, g)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = torch.from_numpy(g["x0"])
    trace = []
    O.ggs_optimize(x0.clone(), pm, iter_num=20, trace=trace)
    _, _, tr = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=20), trace=True)
    tr = tr.cpu()[0]
    for i, ref in enumerate(trace):
        assert pose_err(tr[i, :72], ref["x"].flatten(), "ggs_per_iteration_trace") < TOL
        assert int(tr[i, 73].item()) == ref["n_valid"]
        assert abs(tr[i, 72].item() - ref["loss"].item()) < 1e-4 * abs(ref["loss"].item())


def test_geometry_guided_sampling_vs_reference_fixture(engine, golden):
    g = golden["ggs"]
    _upload(engine, g)
    out, stats = engine.ggs_guide(torch.from_numpy(g["x0"]).to(DEV), 3, dict(synth.GGS_CFG, iter_num=10))
    engine.check_async()
    assert pose_err(out, g["guide_k10"], "geometry_guided_sampling") < TOL
    assert stats[0, :, 1].tolist() == [20.0, 10.0, 10.0, 10.0, 20.0]


def test_early_exit_is_not_an_error(engine, golden):
    """valid/N < min_matches at the first iteration: all five stages break, x is returned untouched."""
    g = golden["ggs"]
    _upload(engine, g, prefix="bad_")
    x0 = torch.from_numpy(g["x0"]).to(DEV)
    for wgs in (1, 0):
        out, stats = engine.ggs_guide(x0, 3, make_ggs_cfg(synth.GGS_CFG, iter_num=10, sampson_max=0.01, wgs_per_seq=wgs))
        engine.check_async()
        assert np.array_equal(out.cpu().numpy(), g["bad_out"])
        assert (stats[0, :, 1] == 0).all() and int(stats[0, 0, 2].item()) == int(g["bad_nvalid_max0.01"])


def test_ggs_input_validation(engine, golden):
    g = golden["ggs"]
    with pytest.raises(RuntimeError, match="out of range"):
        engine.set_matches(0, g["kp1"], g["kp2"], g["i12"] + 100, (8, 3, 224, 224))
    with pytest.raises(ValueError):
        engine.set_matches(0, g["kp1"][:, :1], g["kp2"], g["i12"], (8, 3, 224, 224))
    engine.set_matches(1, g["kp1"][:0], g["kp2"][:0], g["i12"][:0], (8, 3, 224, 224))      # empty clears the slot
    _upload(engine, g, slot=0)
    with pytest.raises(RuntimeError, match="no matches"):
        engine.ggs_guide(torch.zeros(2, 8, 9, device=DEV), 0, synth.GGS_CFG)
    with pytest.raises(RuntimeError, match="frames"):
        engine.ggs_guide(torch.zeros(1, 9, 9, device=DEV), 0, synth.GGS_CFG)


def test_ggs_ragged_and_ordered_pairs(engine):
    """Unequal matches per pair, both (i,j) and (j,i) present, a pair larger than one work item (>512),
    an isolated frame with no matches, shuffled (ungrouped) input order."""
    rng = np.random.default_rng(0)
    N = 7
    enc = synth.make_cameras(N, seed=11)
    md = synth.make_matches(enc[:6], 224, 224, per_pair=40, seed=11, ordered_pairs=True)     # frame 6 isolated
    big = synth.make_matches(enc[[0, 1]], 224, 224, per_pair=700, seed=12)
    kp1 = np.concatenate([md["kp1"], big["kp1"]])
    kp2 = np.concatenate([md["kp2"], big["kp2"]])
    i12 = np.concatenate([md["i12"], big["i12"]])
    keep = rng.random(len(kp1)) < 0.8
    perm = rng.permutation(int(keep.sum()))
    kp1, kp2, i12 = kp1[keep][perm], kp2[keep][perm], i12[keep][perm]
    shape = (N, 3, 224, 224)
    engine.set_matches(0, kp1, kp2, i12, shape)
    pm = O.prepare_matches(kp1, kp2, i12, shape)
    x0 = synth.perturb_pose(enc, seed=3)
    xo = x0.clone().requires_grad_(True)
    v, _ = O.compute_sampson_distance(xo, pm)
    (go,) = torch.autograd.grad(v.mean(), xo)
    loss, grad = engine.ggs_loss_grad(x0.to(DEV))
    assert int(loss[0, 1].item()) == len(v)
    assert rel_err(grad, go) < 1e-4
    assert (grad[0, 6, :7] == 0).all()          # isolated frame: no T/R gradient (focal is shared via the mean)
    ref, _, _ = O.ggs_optimize(x0.clone(), pm, iter_num=5)
    out, _, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=5))
    assert rel_err(out, ref) < TOL
    assert torch.equal(out[0, 6, :7].cpu(), x0[0, 6, :7])                                  # never stepped


# ------------------------------------------------------------------------------------------------ full size
def test_full_size_properties_n20_m57000(engine):
    """BASELINE configs[2] size (N = 20, M = 57 000): oracle check of value/gradient and 3 iterations, plus
    size-independent properties: invariance to the match order, independence of the workgroup count,
    batch slots independent of each other."""
    N = 20
    enc = synth.make_cameras(N, seed=2000)
    md = synth.make_matches(enc, 224, 224, per_pair=300, seed=2000)
    assert len(md["kp1"]) == 57000
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = synth.perturb_pose(enc, seed=7)
    engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    xo = x0.clone().requires_grad_(True)
    v, pr = O.compute_sampson_distance(xo, pm)
    (go,) = torch.autograd.grad(v.mean(), xo)
    loss, grad = engine.ggs_loss_grad(x0.to(DEV))
    assert int(loss[0, 1].item()) == len(v)
    assert rel_err(grad, go) < 1e-4
    ref, _, _ = O.ggs_optimize(x0.clone(), pm, iter_num=3)
    out_a, _, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=3))
    assert rel_err(out_a, ref) < TOL
    # (1) order of the matches inside the upload must not matter beyond fp32 summation noise
    perm = np.random.default_rng(1).permutation(57000)
    engine.set_matches(1, md["kp1"][perm], md["kp2"][perm], md["i12"][perm], md["img_shape"])
    both = torch.cat([x0, x0]).to(DEV)
    out_b, st, _ = engine.ggs_optimize(both, cfg=make_ggs_cfg(iter_num=3))
    engine.check_async()
    assert torch.equal(out_b[0], out_a[0]), "slot 0 must not depend on what runs in slot 1"
    assert rel_err(out_b[1], out_b[0]) < 1e-5
    # (2) workgroups per sequence: bitwise identical
    for wgs in (1, 5, 24):
        o, _, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=3, wgs_per_seq=wgs))
        engine.check_async()
        assert torch.equal(o, out_a)
    # (3) the full default schedule converges to the noise floor (0.25 px^2 for sigma = 0.5 px) like the reference
    out, stats = engine.ggs_guide(x0.to(DEV), 0, synth.GGS_CFG)
    engine.check_async()
    assert stats[0, :, 1].tolist() == [200.0, 100.0, 100.0, 100.0, 200.0]
    final_loss = stats[0, 4, 3].item()
    assert 0.2 < final_loss < 0.45, final_loss


@pytest.mark.parametrize("B", [64, 256])
def test_bench_default_shape_one_workgroup_per_sequence(seeded_diffuser, engine, B):
    """bench.py's launch shapes: one step's batch (64 sequences) and the default engine pass (256 sequences = a workgroup on
    every CU) in one persistent launch, one workgroup per sequence, all 190 pairs of a sequence on that workgroup.  Every slot
    must equal, bit for bit, the same sequence optimised alone (B = 1) at 1 and at 24 workgroups, and a guided sampling pass
    of the batch replayed from its hipGraph must equal the eager launches."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    N = 20
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
    mds, x0s = [], []
    for b in range(B):
        enc = synth.make_cameras(N, seed=3000 + b)
        md = synth.make_matches(enc, 224, 224, per_pair=40 + (b % 5) * 7, seed=3000 + b)          # ragged sizes across slots
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        mds.append(md)
        x0s.append(synth.perturb_pose(enc, seed=50 + b))
    x0 = torch.cat(x0s).to(dev)
    cfg1 = make_ggs_cfg(iter_num=6, wgs_per_seq=1, reserved=_lib.PD_GGS_CFG_NO_LANE_ITEMS)   # the wave-per-item kernels (bitwise across workgroup counts)
    out64, st64, _ = eng.ggs_optimize(x0, cfg=cfg1)
    eng.check_async()
    assert (st64.reshape(B, -1)[:, 1] == 12).all()                      # 6 iterations x 2 runs ("all" = T alone, then all) everywhere
    for b in (0, 17, 42, B - 1):
        md = mds[b]
        engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        for k in (1, 24):
            o, _, _ = engine.ggs_optimize(x0[b:b + 1], cfg=make_ggs_cfg(iter_num=6, wgs_per_seq=k, reserved=_lib.PD_GGS_CFG_NO_LANE_ITEMS))
            engine.check_async()
            assert torch.equal(o[0], out64[b]), (b, k)
    # a whole guided pass of the batch (2 guided steps): hipGraph replay equals eager launches bit for bit, (single sequences are not compared here: at 1 280 token rows the denoiser takes other GEMM
    # tilings than at 20, and 100 free-running steps amplify that rounding, cf. test_sampler_free_running_vs_fp64_oracle)
    z = synth.make_z(B, N, seed=9).to(dev)
    noise = torch.randn(101, B, N, 9, generator=torch.Generator().manual_seed(3)).to(dev)
    gcfg = make_ggs_cfg(dict(synth.GGS_CFG, iter_num=4), wgs_per_seq=1)
    pose_g, _, st_g = eng.sample(z, noise, 2, gcfg, use_graph=True, want_process=False)
    pose_g, st_g = pose_g.clone(), st_g.clone()
    pose_e, _, st_e = eng.sample(z, noise, 2, gcfg, use_graph=False, want_process=False)
    eng.check_async()
    assert torch.equal(pose_g, pose_e) and torch.isfinite(pose_g).all()
    assert torch.equal(st_g.nan_to_num(-1.0), st_e.nan_to_num(-1.0))      # (random poses leave too few valid matches: early exits, 0 iterations)
    eng.close()


def test_long_sequence_n50(engine):
    """BASELINE configs[4] shape: 50 frames, 1225 pairs, 336^2 (matches thinned to 40/pair to keep the CPU
    oracle in seconds)."""
    N = 50
    enc = synth.make_cameras(N, seed=50)
    md = synth.make_matches(enc, 336, 336, per_pair=40, seed=50)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = synth.perturb_pose(enc, seed=51)
    engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    xo = x0.clone().requires_grad_(True)
    v, _ = O.compute_sampson_distance(xo, pm)
    (go,) = torch.autograd.grad(v.mean(), xo)
    # k = 1 and the forced single-exchange kernel replicate the whole backward; the default for N > 32 is the two-hop
    # kernel (backward distributed over the workgroups, per-frame sums by frame owners): same arithmetic per pair, a
    # different (row-ordered) per-frame summation -> equal to rounding, not bitwise
    outs = {}
    for label, wgs, flags in (("k1", 1, 0), ("one_hop", 0, 1), ("two_hop", 0, 0), ("two_hop_k40", 40, 0)):
        cfg1 = make_ggs_cfg(wgs_per_seq=wgs, reserved=flags)
        loss, grad = engine.ggs_loss_grad(x0.to(DEV), cfg=cfg1)
        engine.check_async()
        assert int(loss[0, 1].item()) == len(v)
        assert rel_err(grad, go) < 1e-4
        o, st, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=3, wgs_per_seq=wgs, reserved=flags))
        engine.check_async()
        assert int(st[0, 1].item()) == 6
        outs[label] = o
    ref, _, _ = O.ggs_optimize(x0.clone(), pm, iter_num=3)
    assert torch.equal(outs["k1"], outs["one_hop"])
    for label in outs:
        assert rel_err(outs[label], ref) < TOL, label
    assert rel_err(outs["two_hop"], outs["k1"]) < 1e-5 and rel_err(outs["two_hop_k40"], outs["k1"]) < 1e-5
    # the frame owner sums a frame's rows in a fixed order (four partial sums over every fourth row, round 5): the bits do not depend on
    # how many workgroups share the sequence
    assert torch.equal(outs["two_hop"], outs["two_hop_k40"])


def test_two_hop_kernel_full_guide_n40_batch2(seeded_diffuser):
    """the five-stage guided step at N = 40 (780 pairs, two chunks), two sequences in one launch: the two-hop kernel
    against the single-exchange kernel (stage statistics, iteration counts, poses)."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N = 2, 40
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
    xs = []
    for b in range(B):
        enc = synth.make_cameras(N, seed=60 + b)
        md = synth.make_matches(enc, 224, 224, per_pair=60, seed=60 + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        xs.append(synth.perturb_pose(enc, seed=61 + b))
    x0 = torch.cat(xs).to(dev)
    # a short run (14 iterations over the five stages) must agree to rounding; over a long run the hard s < max
    # threshold amplifies rounding differences (the reference differs from itself by 3e-4..5e-4 per call when its
    # summation order changes), so there the comparison is on iteration counts and loss statistics
    for iters, tol_pose, tol_stat in ((2, 2e-5, 1e-4), (12, 5e-3, 2e-2)):
        res = {}
        for flags in (1, 0):
            out, stats = eng.ggs_guide(x0, 3, make_ggs_cfg(synth.GGS_CFG, iter_num=iters, reserved=flags))
            eng.check_async()
            res[flags] = (out.cpu(), stats.cpu())
        assert torch.equal(res[0][1][:, :, 1], res[1][1][:, :, 1])            # iterations per stage
        assert rel_err(res[0][0], res[1][0]) < tol_pose, (iters, rel_err(res[0][0], res[1][0]))
        assert torch.allclose(res[0][1], res[1][1], rtol=tol_stat, atol=1e-6), iters
    eng.close()


def test_two_engines_overlapped_on_two_streams_match_serial(seeded_diffuser):
    """bench.py double-buffers passes on two engine contexts / HIP streams.  Overlap must not change a bit:
    this caught a hipGraph memset-node ordering problem (exchange tags were zeroed late under concurrent replay),
    fixed by zeroing with a kernel node (pd_ggs_zero_kernel)."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, draw_noise
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N = 4, 12
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    data = []
    for e, eng in enumerate(engs):
        z = synth.make_z(B, N, seed=300 + 10 * e).to(dev)
        noise = torch.stack([draw_noise((N, 9), 100, dev, 4, True, generator=torch.Generator(device=dev).manual_seed(10 * e + b))
                             for b in range(B)], dim=1)
        for b in range(B):
            enc = synth.make_cameras(N, seed=40 + 10 * e + b)
            md = synth.make_matches(enc, 224, 224, per_pair=150, seed=40 + 10 * e + b)
            eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        data.append((z, noise))
    cfg = make_ggs_cfg(synth.GGS_CFG, iter_num=30, min_matches=0, wgs_per_seq=8)   # 4 seq x 8 WGs: both kernels co-resident
    torch.cuda.synchronize()
    refs = []
    for j in range(2):
        refs.append(engs[j].sample(data[j][0], data[j][1], 4, cfg, use_graph=True)[0].clone())
        torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        for i in range(4):
            j = i % 2
            with torch.cuda.stream(streams[j]):
                outs.append((j, engs[j].sample(data[j][0], data[j][1], 4, cfg, use_graph=True)[0]))
        torch.cuda.synchronize()
        for e in engs:
            e.check_async()
        for j, o in outs:
            assert torch.equal(o, refs[j]), f"engine {j}: overlapped result differs from its serial result"
    for e in engs:
        e.close()


def test_pipeline_gated_phases_match_whole_loop(seeded_diffuser):
    """SamplingPipeline (3 contexts; two-stage 2+2 and 1+2 streams, and whole-pass streams): pd_sample_phase UNGUIDED + event gate + GUIDED must give,
    bit for bit, what pd_sample gives for the same batch, whatever else is in flight; also without graphs and
    for a batch without guidance."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, draw_noise
    from posediffusion_amd.pipeline import SamplingPipeline
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N, D = 3, 10, 3
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(D)]
    data = []
    for e, eng in enumerate(engs):
        z = synth.make_z(B, N, seed=500 + 10 * e).to(dev)
        noise = torch.stack([draw_noise((N, 9), 100, dev, 5, True, generator=torch.Generator(device=dev).manual_seed(70 + 10 * e + b))
                             for b in range(B)], dim=1)
        for b in range(B):
            enc = synth.make_cameras(N, seed=80 + 10 * e + b)
            md = synth.make_matches(enc, 224, 224, per_pair=120, seed=80 + 10 * e + b)
            eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        data.append((z, noise))
    cfg = make_ggs_cfg(synth.GGS_CFG, iter_num=25, min_matches=0, wgs_per_seq=6)
    torch.cuda.synchronize()
    refs, refs_plain = [], []
    for j in range(D):
        pose, proc, stats = engs[j].sample(data[j][0], data[j][1], 5, cfg, use_graph=True)
        refs.append((pose.clone(), proc.clone(), stats.clone()))
        refs_plain.append(engs[j].sample(data[j][0], data[j][1], 0, None, use_graph=True)[0].clone())
        torch.cuda.synchronize()
    for use_graph, slots, ustreams in ((True, 2, 2), (False, 2, 1), (True, 3, 0)):
        pipe = SamplingPipeline(engs, slots, dev, unguided_streams=ustreams)   # two-stage (2+2, 1+2) and whole-pass streams
        pend = []
        for i in range(7):
            j = pipe.next_context()
            pend.append(pipe.submit(data[j][0], data[j][1], 5, cfg, use_graph=use_graph, want_process=True))
        j = pipe.next_context()
        plain = pipe.submit(data[j][0], data[j][1], 0, None, use_graph=use_graph)
        pipe.synchronize()
        pipe.check_async()
        for p in pend:
            pose, proc, stats = p.wait()
            assert torch.equal(pose, refs[p.context][0]) and torch.equal(proc, refs[p.context][1])
            assert torch.equal(stats, refs[p.context][2])
        assert torch.equal(plain.wait()[0], refs_plain[plain.context])
    for e in engs:
        e.close()


# ------------------------------------------------------------------------------------------------ per-XCD persistent denoiser
def test_pipeline_with_fresh_matches_per_batch(seeded_diffuser):
    """Streaming use: every submission uploads its own matches into its context first.  pd_ggs_set_matches waits
    for that engine's own work only (no device-wide synchronisation) and re-uses the slot's device blob; results are
    those of the same batch run alone."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, draw_noise
    from posediffusion_amd.pipeline import SamplingPipeline
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N, D, n_sub = 2, 9, 3, 7
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    engs = [PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N) for _ in range(D)]
    cfg = make_ggs_cfg(synth.GGS_CFG, iter_num=15, min_matches=0, wgs_per_seq=5)
    subs = []
    for i in range(n_sub):
        z = synth.make_z(B, N, seed=700 + i).to(dev)
        noise = torch.stack([draw_noise((N, 9), 100, dev, 4, True, generator=torch.Generator(device=dev).manual_seed(90 + 10 * i + b))
                             for b in range(B)], dim=1)
        mds = []
        for b in range(B):
            enc = synth.make_cameras(N, seed=100 + 10 * i + b)
            mds.append(synth.make_matches(enc, 224, 224, per_pair=100 + 17 * ((i + b) % 3), seed=100 + 10 * i + b))
        subs.append((z, noise, mds))
    refs = []
    for z, noise, mds in subs:                      # each batch alone on engine 0
        for b, md in enumerate(mds):
            engs[0].set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        refs.append(engs[0].sample(z, noise, 4, cfg, use_graph=True)[0].clone())
        torch.cuda.synchronize()
    pipe = SamplingPipeline(engs, 3, dev, unguided_streams=0)
    pend = []
    for z, noise, mds in subs:
        j = pipe.next_context()
        for b, md in enumerate(mds):
            engs[j].set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        pend.append(pipe.submit(z, noise, 4, cfg, use_graph=True))
    pipe.synchronize()
    pipe.check_async()
    for p, ref in zip(pend, refs):
        assert torch.equal(p.wait()[0], ref)
    for e in engs:
        e.close()


@pytest.mark.parametrize("case", range(8))
def test_ggs_random_match_structures(engine, case):
    """Randomised match structure against the oracle's autograd: frame counts on both sides of the 32-frame / 512-pair
    boundaries, random subsets of ordered and reversed pairs, per-pair counts from 1 to 900 (several work items per
    pair -> the two-hop kernel must step aside), isolated frames, shuffled input; every workgroup count in play."""
    rng = np.random.default_rng(1000 + case)
    N = [5, 12, 33, 34, 40, 47, 50, 21][case]
    dense = case in (2, 4, 6)
    enc = synth.make_cameras(N, seed=300 + case)
    full = synth.make_matches(enc, 224, 224, per_pair=4, seed=300 + case, ordered_pairs=True)
    i12 = full["i12"]
    pair_key = i12[:, 0] * N + i12[:, 1]
    keys = np.unique(pair_key)
    keep_pairs = keys if dense else keys[rng.random(len(keys)) < 0.6]
    if case == 7:                                                 # frame 20 isolated
        keep_pairs = keep_pairs[(keep_pairs // N != 20) & (keep_pairs % N != 20)]
    kp1, kp2, ii = [], [], []
    big = set(rng.choice(keep_pairs, size=min(2, len(keep_pairs)), replace=False).tolist()) if case % 2 == 1 else set()
    for key in keep_pairs:
        a, b2 = int(key // N), int(key % N)
        cnt = int(rng.integers(600, 900)) if key in big else int(rng.integers(1, 12))
        md = synth.make_matches(enc[[a, b2]], 224, 224, per_pair=cnt, seed=int(key) + 7 * case)
        sel = md["i12"][:, 0] == 0                                # pair (0, 1) of the two-camera scene = (a, b2)
        kp1.append(md["kp1"][sel]); kp2.append(md["kp2"][sel])
        ii.append(np.tile(np.array([[a, b2]], dtype=np.int64), (int(sel.sum()), 1)))
    kp1, kp2, ii = np.concatenate(kp1), np.concatenate(kp2), np.concatenate(ii)
    perm = rng.permutation(len(kp1))
    kp1, kp2, ii = kp1[perm], kp2[perm], ii[perm]
    shape = (N, 3, 224, 224)
    engine.set_matches(0, kp1, kp2, ii, shape)
    pm = O.prepare_matches(kp1, kp2, ii, shape)
    x0 = synth.perturb_pose(enc, seed=400 + case)
    xo = x0.clone().requires_grad_(True)
    v, _ = O.compute_sampson_distance(xo, pm)
    (go,) = torch.autograd.grad(v.mean(), xo)
    ref, _, _ = O.ggs_optimize(x0.clone(), pm, iter_num=2, min_matches=0)
    configs = ((0, 0), (0, 1), (1, 0), (3, 0), (17, 0))
    if case == 6:
        # 2 450 work items: only the two-hop kernel (own items in LDS) can hold them; the single-exchange kernel says so
        with pytest.raises(RuntimeError, match="LDS per workgroup"):
            engine.ggs_loss_grad(x0.to(DEV), cfg=make_ggs_cfg(wgs_per_seq=0, reserved=1, min_matches=0))
        configs = ((0, 0), (17, 0))
    for wgs, flags in configs:
        cfg = make_ggs_cfg(wgs_per_seq=wgs, reserved=flags, min_matches=0)
        loss, grad = engine.ggs_loss_grad(x0.to(DEV), cfg=cfg)
        engine.check_async()
        assert int(loss[0, 1].item()) == len(v), (wgs, flags)
        assert rel_err(grad, go) < 1e-4, (wgs, flags, rel_err(grad, go))
        out, _, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=2, wgs_per_seq=wgs, reserved=flags, min_matches=0))
        engine.check_async()
        assert rel_err(out, ref) < 5e-5, (wgs, flags, rel_err(out, ref))


# ------------------------------------------------------------------------------------------------ N3: evaluation metrics
def _metric_module():
    import importlib
    import sys
    import posediffusion_amd
    if posediffusion_amd.DROPIN_PATH not in sys.path:
        sys.path.insert(0, posediffusion_amd.DROPIN_PATH)
    return importlib.import_module("util.metric")


def test_metrics_vs_reference_fixture(golden):
    """camera_to_rel_deg / AUC / accuracies / ARE kernels against util/metric.py executed in place (tests/golden)."""
    M = _metric_module()
    from posediffusion_amd.compat import PerspectiveCameras
    g = golden["metrics"]
    B = int(g["B"])
    pred = PerspectiveCameras(focal_length=torch.ones(len(g["R_pred"]), 2), R=torch.from_numpy(g["R_pred"]), T=torch.from_numpy(g["T_pred"]))
    gt = PerspectiveCameras(focal_length=torch.ones(len(g["R_gt"]), 2), R=torch.from_numpy(g["R_gt"]), T=torch.from_numpy(g["T_gt"]))
    r, t = M.camera_to_rel_deg(pred, gt, torch.device(DEV), B)
    # degrees from fp32 traces: acos amplifies rounding near 0 / 180 degrees (d angle = d cos / sin angle)
    assert np.abs(r.cpu().numpy() - g["rel_r_deg"]).max() < 2e-2 and np.abs(t.cpu().numpy() - g["rel_t_deg"]).max() < 2e-2
    big = g["rel_r_deg"] > 2.0
    assert np.abs(r.cpu().numpy()[big] - g["rel_r_deg"][big]).max() < 2e-3
    s = M.metrics_summary(torch.from_numpy(g["rel_r_deg"]).to(DEV), torch.from_numpy(g["rel_t_deg"]).to(DEV), 30)
    assert abs(s["Auc_30"] - float(g["auc30"])) < 1e-6
    assert abs(M.calculate_auc_np(g["rel_r_deg"], g["rel_t_deg"], max_threshold=30) - float(g["auc30"])) < 1e-6
    for k in (5, 15, 30):
        assert abs(s[f"Racc_{k}"] - np.mean(g["rel_r_deg"] < k) * 100) < 1e-4
        assert abs(s[f"Tacc_{k}"] - np.mean(g["rel_t_deg"] < k) * 100) < 1e-4
    are = M.compute_ARE(torch.from_numpy(g["R_pred"]).to(DEV), torch.from_numpy(g["R_gt"]).to(DEV))
    assert np.abs(are - g["are_deg"]).max() < 2e-2
    i1, i2 = M.batched_all_pairs(B, int(g["N"]))
    assert len(i1) == len(g["rel_r_deg"]) and int(i1[1]) == 0 and int(i2[1]) == 2


def test_metrics_random_shapes_vs_oracle():
    M = _metric_module()
    from posediffusion_amd.compat import PerspectiveCameras
    for B, N, seed in ((1, 2, 0), (3, 5, 1), (2, 20, 2), (1, 50, 3)):
        g = torch.Generator().manual_seed(seed)
        Rg, Rp = O.quaternion_to_matrix(torch.randn(B * N, 4, generator=g)), O.quaternion_to_matrix(torch.randn(B * N, 4, generator=g))
        Tg, Tp = torch.randn(B * N, 3, generator=g), torch.randn(B * N, 3, generator=g)
        ro, to = O.camera_to_rel_deg(Rp.double(), Tp.double(), Rg.double(), Tg.double(), B)
        one = torch.ones(B * N, 2)
        r, t = M.camera_to_rel_deg(PerspectiveCameras(focal_length=one, R=Rp, T=Tp), PerspectiveCameras(focal_length=one, R=Rg, T=Tg),
                                   torch.device(DEV), B)
        assert np.abs(r.cpu().numpy() - ro.numpy()).max() < 5e-2 and np.abs(t.cpu().numpy() - to.numpy()).max() < 5e-2
        s = M.metrics_summary(r, t, 30)
        assert abs(s["Auc_30"] - O.calculate_auc_np(r.cpu().numpy().astype(np.float64), t.cpu().numpy().astype(np.float64))) < 1e-5


def test_camera_alignment_vs_oracle_and_exact_recovery():
    M = _metric_module()
    from posediffusion_amd.compat import PerspectiveCameras
    torch.manual_seed(5)
    n = 20
    R = O.quaternion_to_matrix(torch.randn(n, 4, dtype=torch.float64))
    T = torch.randn(n, 3, dtype=torch.float64) + torch.tensor([0.0, 0.0, 6.0], dtype=torch.float64)
    RA = O.quaternion_to_matrix(torch.randn(1, 4, dtype=torch.float64))[0]
    TA, s = torch.randn(3, dtype=torch.float64), 0.6
    Rt = RA.T[None] @ R
    Tt = s * T - (TA[None, None] @ Rt)[:, 0]
    one = torch.ones(n, 2)
    src = PerspectiveCameras(focal_length=one, R=R.float().to(DEV), T=T.float().to(DEV))
    # exact similarity: the aligned source cameras ARE the target cameras
    al = M.corresponding_cameras_alignment(src, PerspectiveCameras(focal_length=one, R=Rt.float().to(DEV), T=Tt.float().to(DEV)),
                                           estimate_scale=True, mode="extrinsics", eps=1e-9)
    assert (al.R.cpu().double() - Rt).abs().max() < 5e-6 and (al.T.cpu().double() - Tt).abs().max() < 5e-5
    assert float(M.compute_ARE(al.R, Rt.float().to(DEV)).mean()) < 0.06          # degrees; fp32 acos near 0
    # noisy target: the same least-squares solution as the oracle's restatement (torch SVD)
    Rn = O.quaternion_to_matrix(torch.randn(n, 4, dtype=torch.float64) * 0.05 + torch.tensor([1.0, 0, 0, 0], dtype=torch.float64)) @ Rt
    Tn = Tt + 0.05 * torch.randn(n, 3, dtype=torch.float64)
    Ro, To, _ = O.corresponding_cameras_alignment(R, T, Rn, Tn)
    al = M.corresponding_cameras_alignment(src, PerspectiveCameras(focal_length=one, R=Rn.float().to(DEV), T=Tn.float().to(DEV)))
    assert (al.R.cpu().double() - Ro).abs().max() < 5e-6 and (al.T.cpu().double() - To).abs().max() < 5e-5
    with pytest.raises(ValueError):
        M.corresponding_cameras_alignment(src, src, mode="centers")


# ------------------------------------------------------------------------------------------------ N4: image preprocessing
def test_image_preprocessing_vs_reference_fixture(golden):
    """load_and_preprocess_images (decode on the host, /255 + centre crop + bilinear resize in pd_preprocess_image)
    against util/load_img_folder.py executed in place on tests/golden/images (portrait, landscape, square)."""
    import importlib
    import os
    import sys
    import posediffusion_amd
    if posediffusion_amd.DROPIN_PATH not in sys.path:
        sys.path.insert(0, posediffusion_amd.DROPIN_PATH)
    li = importlib.import_module("util.load_img_folder")
    g = golden["preprocess"]
    img_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "images")
    for S in (32, 17):
        imgs, info = li.load_and_preprocess_images(img_dir, S)
        assert imgs.is_cuda and tuple(imgs.shape) == (3, 3, S, S)
        assert np.abs(imgs.cpu().numpy() - g[f"images_{S}"]).max() < 2e-6          # fp32 bilinear, FMA contraction only
        assert np.array_equal(info["bboxes_xyxy"], g[f"bboxes_{S}"]) and np.array_equal(np.array(info["size"]), g[f"size_{S}"])
        assert np.allclose(info["resized_scales"], g[f"scales_{S}"], rtol=0, atol=0)
    # up-sampling (crop smaller than the output) and a large frame against torch's own interpolate
    from PIL import Image
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=(300, 533, 3), dtype=np.uint8)
    path = os.path.join("/tmp", "pd_big_test.png")
    Image.fromarray(big, "RGB").save(path)
    for S in (224, 336, 700):
        imgs, info = li.load_and_preprocess_images(None, S, image_paths=[path])
        crop = big[:, (533 - 300) // 2:(533 - 300) // 2 + 300].transpose(2, 0, 1).astype(np.float32) / 255.0
        ref = torch.nn.functional.interpolate(torch.from_numpy(crop)[None], size=(S, S), mode="bilinear", align_corners=False)[0]
        assert np.abs(imgs[0].cpu().numpy() - ref.numpy()).max() < 2e-6
    with pytest.raises(NotImplementedError):
        li.load_and_preprocess_images(img_dir, 32, mode="nearest")


# ------------------------------------------------------------------------------------------------ N1: image features
@pytest.fixture(scope="module")
def vit_pair():
    from oracle import vit_oracle as VO
    from posediffusion_amd.vit import VitEngine, vit_state
    net = VO.make_vit(seed=0)
    eng = VitEngine(vit_state(net), torch.device(DEV))
    yield net, eng, VO
    eng.close()


def test_vit_single_scale_vs_oracle(vit_pair):
    """patch embedding, CLS / position tokens, 12 blocks, final LayerNorm at the trained 14 x 14 grid."""
    net, eng, VO = vit_pair
    x = torch.rand(3, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    ref = VO.multiscale_features(net, x, (1,))
    out = eng.multiscale(x.to(DEV), (1,)).cpu()
    assert rel_err(out, ref) < 2e-5, rel_err(out, ref)


def test_vit_multiscale_vs_oracle(vit_pair):
    """the reference's three scales (1, 1/2, 1/3): bilinear rescaling, resampled position grids (7 x 7, 4 x 4), average."""
    net, eng, VO = vit_pair
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    ref = VO.multiscale_features(net, x, (1, 1 / 2, 1 / 3))
    out = eng.multiscale(x.to(DEV), (1, 1 / 2, 1 / 3)).cpu()
    assert rel_err(out, ref) < 2e-5, rel_err(out, ref)
    for sf in (1 / 2, 1 / 3):           # each scale on its own
        assert rel_err(eng.multiscale(x.to(DEV), (sf,)).cpu(), VO.multiscale_features(net, x, (sf,))) < 2e-5


def test_dropin_feature_extractor_loads_dino_names_and_matches_oracle(vit_pair):
    """the drop-in module: DINO-named parameters (strict load), engine rebuilt when the parameters change, frozen flag,
    injected backbones, unknown names."""
    net, _, VO = vit_pair
    models = synth._dropin()
    ext = models.MultiScaleImageFeatureExtractor(modelname="dino_vits16", freeze=True).to(DEV)
    assert ext.get_output_dim() == 384 and not any(p.requires_grad for p in ext.parameters())
    ext._net.load_state_dict(net.state_dict(), strict=True)
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    ref = VO.multiscale_features(net, x, (1, 1 / 2, 1 / 3))
    assert rel_err(ext(x.to(DEV)).cpu(), ref) < 2e-5
    e0 = ext._engine()
    assert ext._engine() is e0                                   # cached while the parameters stand
    net2 = VO.make_vit(seed=9)
    ext._net.load_state_dict(net2.state_dict(), strict=True)      # in-place copy -> version bump -> engine rebuilt
    assert rel_err(ext(x.to(DEV)).cpu(), VO.multiscale_features(net2, x, (1, 1 / 2, 1 / 3))) < 2e-5
    assert ext._engine() is not e0
    ext.scale_factors = [1 / 2]
    assert rel_err(ext(x.to(DEV)).cpu(), VO.multiscale_features(net2, x, (1 / 2,))) < 2e-5
    ext.backbone = lambda im: im.mean(dim=(2, 3))
    assert ext(x.to(DEV)).shape == (2, 3)
    with pytest.raises(ValueError):
        models.MultiScaleImageFeatureExtractor(modelname="vgg16")
    other = models.MultiScaleImageFeatureExtractor(modelname="resnet50")
    with pytest.raises(RuntimeError, match="not implemented"):
        other(x)
    with pytest.raises(RuntimeError, match="AMD GPU"):
        models.MultiScaleImageFeatureExtractor()(x)               # parameters on the CPU: no fallback


@pytest.mark.parametrize("mode", [1, 0, 2])
def test_vit_large_batch_gemm_paths_vs_oracle(vit_pair, mode):
    """>= 1024 token rows take the streamed GEMMs (PD_VIT_OPT_EXACT_FP32): 1 exact fp32 (64 x 64 tiles); 0, the DEFAULT since round 6: the four Linear
    layers on fp16 hi + lo planes with static scales (22 bits: fp32-grade, like the denoiser's default), everything else fp32; 2 the bf16 planes of
    rounds 1-5 (16 bits).  Ragged last row tiles.  Against the fp64 network: the default must be as close to it as the exact mode is (VERDICT round 5,
    item 6: <= 2e-6 of max|z|), the legacy mode 5e-5; tolerance 2e-5 against the fp32 oracle, contract 1e-4."""
    net, eng, VO = vit_pair
    net64 = VO.make_vit(seed=0, dtype=torch.float64)
    eng.set_exact_fp32(mode)
    try:
        tol32, tol64 = (5e-5, 5e-5) if mode == 2 else (2e-5, 2e-6)
        x = torch.rand(6, 3, 224, 224, generator=torch.Generator().manual_seed(21))            # 1 182 rows at scale 1
        out1 = eng.multiscale(x.to(DEV), (1, 1 / 2)).cpu()
        x2 = torch.rand(11, 3, 240, 208, generator=torch.Generator().manual_seed(22))          # 2 156 rows, 15 x 13 grid
        out2 = eng.multiscale(x2.to(DEV), (1,)).cpu()
        e1, e2 = rel_err(out1, VO.multiscale_features(net, x, (1, 1 / 2))), rel_err(out2, VO.multiscale_features(net, x2, (1,)))
        d1 = rel_err(out1, VO.multiscale_features(net64, x.double(), (1, 1 / 2)))
        d2 = rel_err(out2, VO.multiscale_features(net64, x2.double(), (1,)))
        print(f"ViT mode {mode}: vs fp32 oracle {e1:.2e} / {e2:.2e}; vs fp64 network {d1:.2e} / {d2:.2e}")
        assert e1 < tol32 and e2 < tol32, (mode, e1, e2)
        assert d1 < tol64 and d2 < tol64, (mode, d1, d2)
        if mode != 1:      # a plane mode is a different rounding, not a different function: close to the exact path too, and not the exact path
            a = eng.multiscale(x.to(DEV), (1,))
            eng.set_exact_fp32(1)
            b = eng.multiscale(x.to(DEV), (1,))
            assert 0 < rel_err(a, b) < (5e-5 if mode == 2 else 3e-6)
    finally:
        eng.set_exact_fp32(0)


def test_vit_non_finite_weights_fall_back_to_exact_fp32():
    """A network with an inf weight has no static operand bound: the fp16 planes are not built and the default mode runs the exact-fp32 kernels,
    which propagate the value like torch does (no finite garbage)."""
    from oracle import vit_oracle as VO
    from posediffusion_amd.vit import VitEngine, vit_state
    net = VO.make_vit(seed=3)
    with torch.no_grad():
        net.blocks[0].mlp.fc1.weight[5, 7] = float("inf")
    eng = VitEngine(vit_state(net), torch.device(DEV))
    x = torch.rand(6, 3, 224, 224, generator=torch.Generator().manual_seed(4))                  # >= 1 024 rows: the streamed path
    out = eng.multiscale(x.to(DEV), (1,)).cpu()
    assert not torch.isfinite(out).all()
    eng.close()


def test_vit_shallow_network_and_rejected_shapes():
    """`depth` is a parameter of the engine (here 2 blocks); other widths are refused at creation, loudly."""
    from oracle import vit_oracle as VO
    from posediffusion_amd.vit import VitEngine, vit_state
    torch.manual_seed(0)
    net = VO.DinoViT(depth=2).eval()
    eng = VitEngine(vit_state(net), torch.device(DEV))
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    assert rel_err(eng.multiscale(x.to(DEV), (1, 1 / 2)).cpu(), VO.multiscale_features(net, x, (1, 1 / 2))) < 2e-5
    eng.close()
    wide = VO.DinoViT(dim=768, num_heads=12, depth=1).eval()
    with pytest.raises(RuntimeError, match="unsupported ViT shape"):
        VitEngine(vit_state(wide), torch.device(DEV))
    with pytest.raises(KeyError):
        vit_state(torch.nn.Linear(3, 3))
    with pytest.raises(KeyError, match="missing"):
        vit_state(torch.nn.ModuleDict({"blocks": torch.nn.ModuleList([torch.nn.Linear(3, 3)])}))


def test_vit_other_image_sizes_and_limits(vit_pair):
    """non-square and non-multiple-of-16 inputs (ragged token counts, resampled position grid); too many tokens raise."""
    net, eng, VO = vit_pair
    for (H, W, sc) in ((96, 160, (1, 1 / 2)), (130, 77, (1, 1 / 2)), (240, 240, (1, 1 / 2)), (336, 336, (1, 1 / 3)), (16, 16, (1,)),
                       (512, 512, (1,))):
        x = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(H))
        assert rel_err(eng.multiscale(x.to(DEV), sc).cpu(), VO.multiscale_features(net, x, sc)) < 2e-5, (H, W)
    with pytest.raises(RuntimeError, match="tokens per image"):
        eng.multiscale(torch.rand(1, 3, 528, 528).to(DEV), (1,))
    with pytest.raises(ValueError):
        eng.multiscale(torch.rand(1, 3, 224, 224).to(DEV), ())
