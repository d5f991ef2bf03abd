"""GPU (-m gpu), round 2: the parity holes VERDICT.md (round 1) lists -- all through the C-ABI, checked by the oracle.

  * BASELINE configs[4] at FULL size (N = 50, 1 225 pairs x 300 = 367 500 matches, 336^2) on the two-hop kernel;
  * SURVEY.md section 8c's free-running GGS-on criterion against the reference-generated fixture
    tests/golden/guided_free.npz (oracle/make_golden.py make_guided_free);
  * the `sampson < sampson_max` rule of geometry_guided_sampling.py:170 around the threshold;
  * hipGraph replay after a re-upload that changes the GGS launch shape.
"""
import numpy as np
import pytest
import torch

from conftest import pose_err, rel_err
from oracle import pd_oracle as O
from posediffusion_amd import synth
from posediffusion_amd.engine import make_ggs_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5          # asserted on teacher-forced pieces; the contract is 1e-4 (BASELINE.json north_star)


# ------------------------------------------------------------------------------------------------ configs[4], full size
def test_long_sequence_n50_full_size_m367500(engine):
    """BASELINE configs[4] exactly: 50 frames, all 1 225 pairs x 300 matches = 367 500, 336 x 336.  Value, valid count
    (exact up to matches within the contract tolerance of sampson_max) and analytic gradient against the oracle's
    autograd; 3 GGS_optimize iterations (x2: all flags, :86-87) against the oracle; two-hop kernel (default for N > 32),
    single-exchange kernel and k = 1 agree."""
    N = 50
    enc = synth.make_cameras(N, seed=50)
    md = synth.make_matches(enc, 336, 336, per_pair=300, seed=50)
    assert len(md["kp1"]) == 367500 and len(np.unique(md["i12"][:, 0] * N + md["i12"][:, 1])) == 1225
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    x0 = synth.perturb_pose(enc, seed=51)
    engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    s_all, _ = O.compute_sampson_distance(x0, pm, sampson_max=float("inf"))
    n_oracle = int((s_all < 10).sum())
    s_sorted = torch.sort(s_all.detach()).values

    def oracle_at(n_valid):
        """Oracle value / gradient / 3 iterations for the valid set of `n_valid` matches: at sampson_max = 10 when that is the
        oracle's own count, else at the threshold that admits exactly the n_valid smallest Sampson values -- the engine's set
        when one of the matches that sit within the contract tolerance of 10 falls on the other side (at most a few of
        367 500 do; the rule is pinned by test_sampson_threshold_rule_around_sampson_max)."""
        smax = 10.0
        if n_valid != n_oracle:
            smax = float(0.5 * (s_sorted[n_valid - 1].double() + s_sorted[n_valid].double()))
            assert abs(smax - 10.0) < 1e-3, (n_valid, smax)
        xo = x0.clone().requires_grad_(True)
        v, pr = O.compute_sampson_distance(xo, pm, sampson_max=smax)
        assert len(v) == n_valid
        (go,) = torch.autograd.grad(v.mean(), xo)
        return v.mean().item(), go

    ref, _, ref_steps = O.ggs_optimize(x0.clone(), pm, iter_num=3)
    assert ref_steps == 6
    lo_n, hi_n = int((s_all < 10 * (1 - 1e-4)).sum()), int((s_all < 10 * (1 + 1e-4)).sum())
    outs, cache = {}, {}
    for label, wgs, flags in (("two_hop", 0, 0), ("two_hop_k64", 64, 0), ("one_hop", 0, 1), ("k1", 1, 0)):
        cfg = make_ggs_cfg(wgs_per_seq=wgs, reserved=flags)
        loss, grad = engine.ggs_loss_grad(x0.to(DEV), cfg=cfg)
        engine.check_async()
        n_valid = int(loss[0, 1].item())
        assert lo_n <= n_valid <= hi_n, (label, n_valid, lo_n, hi_n)
        if n_valid not in cache:
            cache[n_valid] = oracle_at(n_valid)
        mean_o, go = cache[n_valid]
        assert abs(loss[0, 0].item() - mean_o) < TOL * mean_o, label
        assert rel_err(grad, go) < 1e-4, (label, rel_err(grad, go))
        o, st, _ = engine.ggs_optimize(x0.to(DEV), cfg=make_ggs_cfg(iter_num=3, wgs_per_seq=wgs, reserved=flags))
        engine.check_async()
        assert int(st[0, 1].item()) == 6, label
        # 6 free-running iterations: a straddling match moves the result by ~1e-5 of |x| per iteration it flips in
        assert pose_err(o, ref, "configs4_full_size_6_iterations") < (TOL if n_valid == n_oracle else 1e-4), (label, rel_err(o, ref))
        outs[label] = o
    print("configs[4] full size: engine valid counts", sorted(cache), "oracle", n_oracle)
    assert torch.equal(outs["k1"], outs["one_hop"])
    assert rel_err(outs["two_hop"], outs["k1"]) < 1e-5 and rel_err(outs["two_hop_k64"], outs["k1"]) < 1e-5
    assert torch.equal(outs["two_hop"], outs["two_hop_k64"])          # (the two-hop kernel's bits do not depend on the workgroup count)


# ------------------------------------------------------------------------------------------------ free-running GGS-on
def _free_running_case(engine, g, s, cfg):
    """One seed of a free-running GGS-on fixture through the engine: (pose deviation from fp64, the reference-fp32's own,
    relative gap of the final mean Sampson error to the fp64 oracle's, the reference-fp32's own gap)."""
    cond_start = int(g["cond_start_step"])
    shape = tuple(int(v) for v in g["img_shape"])
    kp1, kp2, i12 = g[f"s{s}_kp1"], g[f"s{s}_kp2"], g[f"s{s}_i12"]
    engine.set_matches(0, kp1, kp2, i12, shape)
    z, noise = torch.from_numpy(g[f"s{s}_z"]).to(DEV), torch.from_numpy(g[f"s{s}_noise"]).to(DEV)
    outs = []
    for use_graph in (True, False):
        pose, _, stats = engine.sample(z, noise, cond_start, cfg, use_graph=use_graph)
        engine.check_async()
        outs.append(pose.cpu())
    assert torch.equal(outs[0], outs[1]), "hipGraph replay must equal eager launches bit for bit"
    assert stats[:, 0, :, 1].sum().item() == 700 * cond_start, "every guided step must run its 700 iterations"
    pose = outs[0]
    p64, p32 = g[f"s{s}_pose64"], g[f"s{s}_pose32"]
    pm = O.prepare_matches(kp1, kp2, i12, shape)
    v, _ = O.compute_sampson_distance(pose.double(), pm)
    s64, s32 = float(g[f"s{s}_sampson64"][0]), float(g[f"s{s}_sampson32"][0])
    return rel_err(pose, p64), rel_err(p32, p64), abs(float(v.mean()) - s64) / s64, abs(s32 - s64) / s64


def test_free_running_ggs_on_criterion_vs_reference_fixture(engine, golden):
    """SURVEY.md section 8c, free-running GGS-on (scaled-down BASELINE configs[2]: N = 8, 28 pairs x 60 matches, 100
    steps, the last 3 guided with the full 700-iteration schedule; three seeds).  The fixture holds, per seed, the
    UNMODIFIED reference's fp32 result and the fp64 oracle's on the same z / noise / matches.

    Pass criterion as SURVEY states it, PER SEED (no summing across seeds): engine-vs-fp64 pose
    deviation <= 2 x (reference-fp32-vs-fp64 deviation).  Final mean Sampson error: SURVEY asks for 1 % of the
    oracle's; the fixture shows the reference itself misses that by far (fp32 vs fp64: 2.7 %, 4.0 %, 50 % on these
    seeds: |pose| ~ 45 with random-init weights, a hard threshold, 2 100 iterations), so the bound per seed is
    max(1 %, 2 x the reference's own relative gap for that seed)."""
    g = golden["guided_free"]
    cfg = dict(synth.GGS_CFG)
    rows = [_free_running_case(engine, g, s, cfg) for s in g["seeds"].tolist()]
    print("free-running GGS-on (N = 8): (engine dev, reference dev, engine Sampson gap, reference gap) per seed:", rows)
    for dev, ref_dev, gap, ref_gap in rows:
        assert dev <= 2.0 * ref_dev, rows
        assert gap <= max(0.01, 2.0 * ref_gap), rows


def test_free_running_ggs_on_full_size_configs2_vs_reference_fixture(engine, golden):
    """The same criterion at the benchmark's REAL size: one seed of BASELINE configs[2] exactly --
    N = 20, 190 pairs x 300 = 57 000 matches, 224^2, 100 steps, the last 10 guided x 700 iterations = 7 000 iterations --
    through the unmodified reference in fp32 and the fp64 oracle (oracle/make_golden.py guided_free_full; the reference
    ran all 50 GGS_optimize calls to completion).  Per-seed bounds: pose deviation from fp64 <= 2 x the reference's own
    (5.8e-4), final mean Sampson gap to the fp64 oracle's <= max(1 %, 2 x the reference's own gap) (the reference's
    fp32 run ends at 0.383 against fp64's 0.302: 27 %)."""
    g = golden["guided_free_full"]
    assert int(g["cond_start_step"]) == 10 and len(g["s0_kp1"]) == 57000 and int(g["s0_ref_optimize_calls"]) == 50
    cfg = dict(synth.GGS_CFG)
    dev, ref_dev, gap, ref_gap = _free_running_case(engine, g, 0, cfg)
    print(f"free-running GGS-on, configs[2] full size: engine vs fp64 {dev:.3e} (reference fp32 {ref_dev:.3e}); "
          f"final mean Sampson gap to fp64: engine {gap:.3%}, reference fp32 {ref_gap:.3%}")
    assert dev <= 2.0 * ref_dev, (dev, ref_dev)
    assert gap <= max(0.01, 2.0 * ref_gap), (gap, ref_gap)


# ------------------------------------------------------------------------------------------------ the hard threshold
def _all_sampson_fp32(x, pm):
    s, _ = O.compute_sampson_distance(x, pm, sampson_max=float("inf"))
    return s.detach()


def test_sampson_threshold_rule_around_sampson_max(engine, golden):
    """geometry_guided_sampling.py:170 keeps `sampson < sampson_max` on torch's IEEE quotient.  The engine decides on the
    IEEE quotient too (csrc/pd_ggs.hip sampson_step2: fast 1-ulp pass, exact re-run of an item that has a match within
    16 ulp of the threshold), but ITS top / bottom differ from torch's by fp32 rounding order (F is built in another
    order), so the documented rule is: the valid set equals the reference's except for matches whose Sampson value lies
    within the contract tolerance (1e-4 relative) of sampson_max.  Checked with sampson_max placed EXACTLY on oracle
    Sampson values, one ulp above and one below."""
    gg = golden["ggs"]
    shape = tuple(int(v) for v in gg["img_shape"])
    engine.set_matches(0, gg["kp1"], gg["kp2"], gg["i12"], shape)
    pm = O.prepare_matches(gg["kp1"], gg["kp2"], gg["i12"], shape)
    x0 = torch.from_numpy(gg["x0"])
    s = _all_sampson_fp32(x0, pm)
    order = torch.argsort(s)
    picks = [order[int(q * (len(s) - 1))].item() for q in (0.05, 0.3, 0.5, 0.7, 0.9, 0.97)]
    for m in picks:
        for smax in (np.nextafter(np.float32(s[m]), np.float32(0)), np.float32(s[m]), np.nextafter(np.float32(s[m]), np.float32(np.inf))):
            smax = float(smax)
            lo = int((s < smax * (1 - 1e-4)).sum())
            hi = int((s < smax * (1 + 1e-4)).sum())
            for k in (1, 0):
                loss, _ = engine.ggs_loss_grad(x0.to(DEV), cfg=make_ggs_cfg(sampson_max=smax, wgs_per_seq=k, min_matches=0))
                engine.check_async()
                assert lo <= int(loss[0, 1].item()) <= hi, (m, smax, lo, int(loss[0, 1].item()), hi)


def test_sampson_threshold_fast_and_exact_paths_agree(engine):
    """One match replicated 96 times in a pair (identical Sampson value in every copy, whatever the engine's rounding
    of F).  (a) The engine's switching point -- the smallest sampson_max that makes the copies valid, found by bisection
    over the float representation -- lies within the contract tolerance of the oracle's Sampson value.  (b) Sweeping
    sampson_max over the 97 consecutive floats around that point switches the copies exactly ONCE, all together,
    monotonically.  (c) For an unchanged valid set, loss and gradient just inside the 16-ulp band (exact IEEE path) equal
    those well outside it (fast 1-ulp path)."""
    N = 6
    enc = synth.make_cameras(N, seed=77)
    md = synth.make_matches(enc, 224, 224, per_pair=20, seed=77)
    kp1, kp2, i12 = md["kp1"].copy(), md["kp2"].copy(), md["i12"].copy()
    x0 = synth.perturb_pose(enc, seed=78)
    s0 = _all_sampson_fp32(x0, O.prepare_matches(kp1, kp2, i12, md["img_shape"]))
    m = None
    for c in torch.nonzero((s0 > 0.5) & (s0 < 5.0)).flatten().tolist():      # a match with no neighbour within 1 %
        if ((s0 - s0[c]).abs() < 0.01 * s0[c]).sum() == 1:
            m = c
            break
    assert m is not None
    copies = 96
    kp1 = np.concatenate([kp1, np.repeat(kp1[m:m + 1], copies, 0)])
    kp2 = np.concatenate([kp2, np.repeat(kp2[m:m + 1], copies, 0)])
    i12 = np.concatenate([i12, np.repeat(i12[m:m + 1], copies, 0)])
    engine.set_matches(0, kp1, kp2, i12, md["img_shape"])
    below = int((s0 < s0[m] * 0.99).sum())                                   # matches certainly below the window

    def probe(bits):
        smax = float(np.array([bits], dtype=np.int32).view(np.float32)[0])
        loss, grad = engine.ggs_loss_grad(x0.to(DEV), cfg=make_ggs_cfg(sampson_max=smax, min_matches=0))
        return int(loss[0, 1].item()), loss[0, 0].item(), grad.cpu()

    as_bits = lambda f: int(np.array([f], dtype=np.float32).view(np.int32)[0])   # noqa: E731  (positive floats order like ints)
    lo, hi = as_bits(float(s0[m]) * (1 - 5e-3)), as_bits(float(s0[m]) * (1 + 5e-3))
    assert probe(lo)[0] == below and probe(hi)[0] == below + copies + 1
    while hi - lo > 1:                                                       # invariant: lo invalid, hi valid
        mid = (lo + hi) // 2
        if probe(mid)[0] > below:
            hi = mid
        else:
            lo = mid
    engine.check_async()
    s_switch = float(np.array([lo], dtype=np.int32).view(np.float32)[0])     # largest smax with the copies still invalid = the engine's s
    assert abs(s_switch - float(s0[m])) <= 1e-4 * float(s0[m]), (s_switch, float(s0[m]))
    res = {d: probe(hi + d) for d in list(range(-48, 49))}
    counts = np.array([res[d][0] for d in range(-48, 49)])
    assert (np.diff(counts) >= 0).all() and set(counts.tolist()) == {below, below + copies + 1}
    assert counts[47] == below and counts[48] == below + copies + 1          # the single jump sits at the bisected point
    for inside, outside in ((2, 44), (-3, -44)):                             # exact path vs fast path, same valid set
        assert res[inside][0] == res[outside][0]
        assert abs(res[inside][1] - res[outside][1]) <= 2e-6 * abs(res[outside][1])
        assert rel_err(res[inside][2], res[outside][2]) < 2e-6
    engine.check_async()


# ------------------------------------------------------------------------------------------------ graph cache vs re-upload
def test_graph_replay_after_reupload_with_other_launch_shape(seeded_diffuser):
    """the captured GGS nodes bake in workgroups per sequence, item slots and LDS size, all derived
    from the uploaded matches.  Sampling with use_graph, re-uploading matches with MORE work items (demo.py / test.py
    flow: same model, next sequence) and sampling again must not replay the first launch shape: every replayed result
    must equal eager launches bit for bit, and going back to the first matches must re-use the first graph."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    N = 10
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=1, max_N=N)
    enc = synth.make_cameras(N, seed=600)
    small = synth.make_matches(enc, 224, 224, per_pair=30, seed=600)                       # 45 items
    big = synth.make_matches(enc, 224, 224, per_pair=700, seed=601, ordered_pairs=True)    # 90 pairs x 2 items = 180 items
    other_n = synth.make_matches(enc[:7], 224, 224, per_pair=30, seed=602)                 # uploaded for 7 frames
    z = synth.make_z(1, N, seed=9).to(dev)
    noise = torch.randn(101, 1, N, 9, generator=torch.Generator().manual_seed(5)).to(dev)
    cfg = make_ggs_cfg(dict(synth.GGS_CFG, iter_num=5), min_matches=0)

    def run(md, use_graph):
        eng.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        pose, _, st = eng.sample(z, noise, 2, cfg, use_graph=use_graph, want_process=False)
        eng.check_async()
        return pose.clone(), st.clone()

    for md in (small, big, small, big):
        pg, sg = run(md, True)
        pe, se = run(md, False)
        assert torch.equal(pg, pe) and torch.equal(sg.nan_to_num(-1.0), se.nan_to_num(-1.0))
        assert torch.isfinite(pg).all()
    # a replay must not skip pd_ggs_launch's checks either: matches uploaded for another frame count
    eng.set_matches(0, other_n["kp1"], other_n["kp2"], other_n["i12"], other_n["img_shape"])
    with pytest.raises(RuntimeError, match="uploaded for 7 frames"):
        eng.sample(z, noise, 2, cfg, use_graph=True, want_process=False)
    eng.close()


# ------------------------------------------------------------------------------------------------ asynchronous match ingestion
def _ragged_batch(N, seeds, ordered=False, shuffle=True, per_pair=lambda b: 30 + 17 * b, big_pair=None):
    rng = np.random.default_rng(4242)
    mds, x0s = [], []
    for b, seed in enumerate(seeds):
        enc = synth.make_cameras(N, seed=seed)
        md = synth.make_matches(enc, 224, 224, per_pair=per_pair(b), seed=seed, ordered_pairs=ordered)
        if big_pair is not None and b == 0:              # one pair with > 512 matches -> several work items
            extra = synth.make_matches(enc[[0, 1]], 224, 224, per_pair=big_pair, seed=seed + 1)
            sel = extra["i12"][:, 0] == 0
            md = {"kp1": np.concatenate([md["kp1"], extra["kp1"][sel]]), "kp2": np.concatenate([md["kp2"], extra["kp2"][sel]]),
                  "i12": np.concatenate([md["i12"], np.tile(np.array([[0, 1]], dtype=np.int64), (int(sel.sum()), 1))]),
                  "img_shape": md["img_shape"]}
        if shuffle:
            perm = rng.permutation(len(md["kp1"]))
            md = {"kp1": md["kp1"][perm], "kp2": md["kp2"][perm], "i12": md["i12"][perm], "img_shape": md["img_shape"]}
        mds.append(md)
        x0s.append(synth.perturb_pose(enc, seed=seed + 9))
    return mds, torch.cat(x0s)


@pytest.mark.parametrize("where", ["device", "pinned"])
@pytest.mark.parametrize("case", ["n9_ragged_shuffled", "n20_full", "n12_ordered_big_pair", "n40_two_hop"])
def test_async_match_ingestion_is_bitwise_the_host_upload(seeded_diffuser, case, where):
    """pd_ggs_set_matches_csr_async (stable counting sort + table build on the device, no host sync) must leave exactly
    the state pd_ggs_set_matches (host counting sort) leaves: the same batch uploaded both ways into two engines gives
    bitwise-equal GGS results -- shuffled input (stability of the sort), ragged sizes, pairs split into several items,
    both orders of a pair, the two-hop kernel with hints, with and without hints."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, pack_matches
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    hints = {}
    if case == "n9_ragged_shuffled":
        N, (mds, x0) = 9, _ragged_batch(9, [800, 801, 802, 803, 804])
    elif case == "n20_full":
        N, (mds, x0) = 20, _ragged_batch(20, [810, 811], shuffle=False, per_pair=lambda b: 300)
        hints = dict(max_pairs=190, max_matches_per_pair=512, one_order=True)     # (i < j pairs: the same launch plan as the host-built tables)
    elif case == "n12_ordered_big_pair":
        N, (mds, x0) = 12, _ragged_batch(12, [820, 821, 822], ordered=True, big_pair=1300)
    else:
        N, (mds, x0) = 40, _ragged_batch(40, [830, 831], shuffle=True, per_pair=lambda b: 6 + b)
        hints = dict(max_pairs=780, max_matches_per_pair=64)
    B = len(mds)
    tables = {k: v for k, v in diff.named_buffers(recurse=False)}
    e_host = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N)
    e_dev = PoseEngine(denoiser_state(diff.model), tables, device=dev, max_B=B, max_N=N)
    for b, md in enumerate(mds):
        e_host.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    kp1, kp2, i12, off, shape = pack_matches(mds, pin=True)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):                            # upload on ANOTHER stream than the GGS launches below
        if where == "device":                                # (the copies are ordered before the ingestion on that stream)
            kp1, kp2, i12 = kp1.to(dev, non_blocking=True), kp2.to(dev, non_blocking=True), i12.to(dev, non_blocking=True)
        e_dev.set_matches_async(0, kp1, kp2, i12, off, shape, **hints)
    del kp1, kp2, i12                                        # the engine keeps them alive until the upload ran
    x0 = x0.to(dev)
    for wgs in (0, 1, 3):
        cfg = make_ggs_cfg(iter_num=4, wgs_per_seq=wgs, min_matches=0)
        lh, gh = e_host.ggs_loss_grad(x0, cfg=cfg)
        ld, gd = e_dev.ggs_loss_grad(x0, cfg=cfg)
        assert torch.equal(lh, ld) and torch.equal(gh, gd), (case, wgs)
        oh, sh, _ = e_host.ggs_optimize(x0, cfg=cfg)
        od, sd, _ = e_dev.ggs_optimize(x0, cfg=cfg)
        e_host.check_async()
        e_dev.check_async()
        assert torch.equal(oh, od) and torch.equal(sh.nan_to_num(-1.0), sd.nan_to_num(-1.0)), (case, wgs)
    # a second upload into the same slots (other data, no allocation) while nothing waits on the host
    mds2 = list(reversed(mds))
    for b, md in enumerate(mds2):
        e_host.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    e_dev.set_matches_async(0, *pack_matches(mds2, pin=True), **hints)
    cfg = make_ggs_cfg(iter_num=3, min_matches=0)
    oh, _, _ = e_host.ggs_optimize(x0, cfg=cfg)
    od, _, _ = e_dev.ggs_optimize(x0, cfg=cfg)
    e_dev.check_async()
    assert torch.equal(oh, od)
    e_host.close()
    e_dev.close()


def test_async_match_ingestion_flags_bad_input(seeded_diffuser):
    """Violated hints / out-of-range frame indices cannot raise synchronously (the host never sees the data): they set
    the asynchronous error word, pd_check_async_error reports which, and the word is cleared."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, pack_matches
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    N = 8
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=2, max_N=N)
    mds, x0 = _ragged_batch(N, [900, 901], shuffle=False, per_pair=lambda b: 40)
    packed = pack_matches(mds, pin=True)
    eng.set_matches_async(0, *packed, max_pairs=10)                       # there are 28 pairs
    with pytest.raises(RuntimeError, match="pd_match_hints violated"):
        eng.check_async()
    eng.check_async()                                                     # cleared
    eng.set_matches_async(0, *packed, max_matches_per_pair=39)            # 40 per pair
    with pytest.raises(RuntimeError, match="pd_match_hints violated"):
        eng.check_async()
    bad = [dict(md) for md in mds]
    bad[1]["i12"] = bad[1]["i12"].copy()
    bad[1]["i12"][5, 1] = N                                               # frame index == n_frames
    eng.set_matches_async(0, *pack_matches(bad, pin=True))
    with pytest.raises(RuntimeError, match="frame index outside"):
        eng.check_async()
    eng.set_matches_async(0, *packed)                                     # and a good upload works afterwards
    out, _, _ = eng.ggs_optimize(x0.to(dev), cfg=make_ggs_cfg(iter_num=2))
    eng.check_async()
    assert torch.isfinite(out).all()
    with pytest.raises(ValueError, match="pinned"):
        eng.set_matches_async(0, torch.zeros(4, 2, dtype=torch.float64), torch.zeros(4, 2, dtype=torch.float64),
                              torch.zeros(4, 2, dtype=torch.int64), [0, 4], (N, 3, 224, 224))
    eng.close()


# ------------------------------------------------------------------------------------------------ fast mode (split precision)
def test_split_precision_denoiser_fast_mode_deviation(seeded_diffuser, oracle_weights):
    """PD_OPT_DENOISER_SPLIT (fast mode, never the default): the encoder GEMMs of the large-batch path as bf16 hi + lo, three
    bf16 MFMA products, fp32 accumulation.  Measured here, not assumed: (a) one denoiser step at 1 040 token rows against the
    fp32 oracle -- the split mode must stay inside the 1e-4 contract (the exact mode is asserted at 2e-5); (b) 100 free-running
    steps against the fp64 oracle next to the exact mode's own deviation (chaotic with random-init weights: same order of
    magnitude is the bound, as for the exact mode in test_sampler_free_running_vs_fp64_oracle)."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state, draw_noise
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    B, N = 52, 20                                                       # 1 040 rows: the streamed path
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
    g = torch.Generator().manual_seed(77)
    x, z = torch.randn(B, N, 9, generator=g), synth.make_z(B, N, seed=3)
    devs = {}
    for t in (99, 40, 0):
        ref = O.denoiser_forward(oracle_weights, x, torch.full((B,), t, dtype=torch.long), z)
        for mode in (False, True):
            eng.set_split_precision(mode)
            devs[(t, mode)] = rel_err(eng.denoise(x.to(dev), z.to(dev), t), ref)
    print("denoiser step vs fp32 oracle, (t, split) -> rel:", {k: f"{v:.2e}" for k, v in devs.items()})
    assert all(v < TOL for (t, m), v in devs.items() if not m)
    assert all(v < 1e-4 for (t, m), v in devs.items() if m)
    # free-running: 100 steps, GGS off, against the fp64 oracle (8 sequences of the batch are enough for the oracle's clock)
    noise = draw_noise((B, N, 9), 100, dev, generator=torch.Generator(device=dev).manual_seed(5))
    finals = {}
    for mode in (False, True):
        eng.set_split_precision(mode)
        pose, _, _ = eng.sample(z.to(dev), noise, 0, None, use_graph=True, want_process=False)
        finals[mode] = pose.cpu()
    eng.set_split_precision(False)
    sub = slice(0, 4)
    sd64 = {k: v.double() for k, v in oracle_weights.items()}
    t64 = O.diffusion_tables(dtype=torch.float64)
    nz = noise.cpu().double()
    with torch.no_grad():
        p64, _ = O.p_sample_loop(sd64, t64, z[sub].double(), nz[0][sub], [None if t == 0 else nz[100 - t][sub] for t in range(100)])
    d_exact, d_split = rel_err(finals[False][sub], p64), rel_err(finals[True][sub], p64)
    print(f"free-running 100 steps vs fp64: exact fp32 mode {d_exact:.2e}, split mode {d_split:.2e}; split vs exact {rel_err(finals[True], finals[False]):.2e}")
    assert d_split <= max(4.0 * d_exact, 1e-3), (d_split, d_exact)
    eng.close()


@pytest.mark.gpu
def test_one_workgroup_per_sequence_kernel_variants_agree_bitwise(seeded_diffuser):
    """At one workgroup per sequence (the bench shape: 190 items, 24 rounds per wave) pd_ggs_plan picks the LDS-staged kernel with
    12 waves (three per SIMD, one staging buffer per wave).  pd_ggs_cfg.reserved switches back to 8 waves (PD_GGS_CFG_WAVES8 = 4)
    and to the register-streamed match pass (PD_GGS_CFG_NO_LDS_STAGING = 2): same arithmetic in the same order -> the same bits,
    for a guided step with the full stage schedule and for the per-stage statistics."""
    from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
    from posediffusion_amd.host import denoiser_state
    dev = torch.device("cuda:0")
    diff = seeded_diffuser.to(dev)
    B, N = 6, 20
    eng = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
    xs = []
    for b in range(B):
        enc = synth.make_cameras(N, seed=900 + b)
        md = synth.make_matches(enc, 224, 224, per_pair=(300, 130, 64, 333, 200, 90)[b], seed=900 + b)   # 2..6 staging pieces per item
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        xs.append(synth.perturb_pose(enc, seed=910 + b))
    x0 = torch.cat(xs).to(dev)
    res = {}
    for flags in (0, 4, 2, 6):
        out, stats = eng.ggs_guide(x0, 3, make_ggs_cfg(synth.GGS_CFG, iter_num=5, wgs_per_seq=1, reserved=flags | 16))   # (16 = PD_GGS_CFG_NO_LANE_ITEMS: the wave-per-item kernels)
        eng.check_async()
        res[flags] = (out.clone(), stats.clone())
    for flags in (4, 2, 6):
        assert torch.equal(res[0][0], res[flags][0]), flags
        assert torch.equal(res[0][1].nan_to_num(-1.0), res[flags][1].nan_to_num(-1.0)), flags
    assert torch.isfinite(res[0][0]).all()
    eng.close()
