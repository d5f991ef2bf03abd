"""GPU (-m gpu), round 6: the lane-per-item cut rule as the engine builds it (host and device builders) against its Python mirror
(bench_legs.lane_stream_fraction, from which bench.py reports streamed bytes); the in-kernel launch stamps bench.py's roofline is made
of; the engine's stage table; mixed waves (masked steps two at a time) against the wave-per-item kernels.  Everything through the C-ABI.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pose_err, rel_err
from posediffusion_amd import _lib, synth
from posediffusion_amd.engine import make_ggs_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LANE, NOLANE = _lib.PD_GGS_CFG_LANE_ITEMS, _lib.PD_GGS_CFG_NO_LANE_ITEMS


def _matches_with_counts(enc, counts, seed):
    """synthetic matches whose frame pairs (i < j, in order) hold exactly `counts` matches each"""
    N = len(enc) if hasattr(enc, "__len__") else enc.shape[0]
    md = synth.make_matches(enc, 224, 224, per_pair=int(max(counts)), seed=seed)
    key = md["i12"][:, 0] * N + md["i12"][:, 1]
    keep = np.zeros(len(key), dtype=bool)
    pairs = sorted(set(key.tolist()))
    assert len(pairs) == len(counts)
    for k, c in zip(pairs, counts):
        idx = np.nonzero(key == k)[0][: int(c)]
        keep[idx] = True
    return {"kp1": md["kp1"][keep], "kp2": md["kp2"][keep], "i12": md["i12"][keep], "img_shape": md["img_shape"]}


def _pair_counts(md, N):
    key = md["i12"][:, 0] * N + md["i12"][:, 1]
    return [int(c) for c in np.unique(key, return_counts=True)[1]]


CASES = {
    "bench_190x300": (20, lambda rng: [300] * 190),                               # -> waves [75 x 4, 38 x 4], 504 items (k = 2, d = 4)
    "uniform_190x96": (20, lambda rng: [96] * 190),
    "ragged_50_to_600": (20, lambda rng: rng.integers(50, 601, size=190).tolist()),
    "ragged_3_to_400_n12": (12, lambda rng: rng.integers(3, 401, size=66).tolist()),
    "two_sizes": (20, lambda rng: [500] * 40 + [120] * 150),
    "tiny_190x7": (20, lambda rng: [7] * 190),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_lane_cut_rule_engine_vs_python_mirror(engine, case):
    """pd_ggs_set_matches (host builder) and pd_ggs_set_matches_csr_async (device builder) cut a sequence's pairs into lane items by
    pd_lane_rank + pd_lane_pass_cost (csrc/pd_internal.h, round 6: k more cuts for the spare / k - d longest pairs, (k, d) by the modelled
    match pass); bench_legs.lane_stream_fraction restates the rule in Python for the bench line's `fabric` object.  The three must agree on
    the number of lane items and on every wave's steps -- for uniform, ragged and two-sized pair counts."""
    import bench_legs as L
    N, gen = CASES[case]
    counts = gen(np.random.default_rng(11))
    enc = synth.make_cameras(N, seed=77)
    md = _matches_with_counts(enc, counts, seed=78)
    counts = _pair_counts(md, N)
    _, items_py, waves_py = L.lane_stream_fraction(counts)
    engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    items_h, nw_h, _, waves_h = engine.lane_tables(0)
    assert (items_h, waves_h) == (items_py, waves_py), (case, items_h, waves_h, items_py, waves_py)
    if case == "bench_190x300":
        assert waves_h == [75, 75, 75, 75, 38, 38, 38, 38] and items_h == 504
    off = np.array([0, len(md["kp1"])], dtype=np.int64)
    kp1, kp2, i12 = (torch.from_numpy(np.ascontiguousarray(md[k])).to(DEV) for k in ("kp1", "kp2", "i12"))
    engine.set_matches_async(0, kp1, kp2, i12, off, md["img_shape"], max_pairs=N * (N - 1) // 2, max_matches_per_pair=int(max(counts)), one_order=True)
    torch.cuda.synchronize()
    engine.check_async()
    items_d, nw_d, _, waves_d = engine.lane_tables(0)
    assert (items_d, waves_d) == (items_h, waves_h), (case, "device builder", items_d, waves_d)


def test_mixed_waves_masked_pairs_match_the_wave_kernels(engine):
    """Pair counts of two sizes leave a wave that mixes item lengths: its steps past the shortest item run masked, two at a time since round 6
    (PD_LANE_STEP2(true, ..)).  Same valid counts and iterations as the wave-per-item kernels, poses within the teacher-forced bound per
    column group after 6 iterations."""
    N = 20
    enc = synth.make_cameras(N, seed=91)
    md = _matches_with_counts(enc, [433] * 30 + [97] * 160, seed=92)
    engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    _, _, _, waves = engine.lane_tables(0)
    x0 = synth.perturb_pose(enc, seed=5).to(DEV)
    res = {}
    for tag, flags in (("lane", LANE), ("wave", NOLANE)):
        loss, grad = engine.ggs_loss_grad(x0, cfg=make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=flags))
        o, st, _ = engine.ggs_optimize(x0, cfg=make_ggs_cfg(synth.GGS_CFG, iter_num=3, wgs_per_seq=1, reserved=flags))
        engine.check_async()
        res[tag] = (loss.cpu(), grad.cpu(), o.cpu(), st.cpu())
    l, w = res["lane"], res["wave"]
    assert len(set(waves)) > 1, waves
    assert torch.equal(l[0][:, 1], w[0][:, 1]) and torch.equal(l[3][:, 1], w[3][:, 1])          # valid counts, iterations stepped
    assert rel_err(l[0][:, 0], w[0][:, 0]) < 2e-6 and rel_err(l[1], w[1]) < 2e-5
    assert pose_err(l[2], w[2], "mixed_waves_lane_vs_wave_6_iterations") < 2e-5


def test_ggs_launch_stamps_and_stage_table(engine, golden):
    """pd_ggs_launch_stamps: the lane kernel stamps every launch (start of workgroup 0, end of the last workgroup; wall_clock64 ticks) -- what
    bench.py's roofline.frac is made of.  A step-level pd_ggs_guide uses slot 0; the stamped duration lies inside the hipEvent bracket of the
    call.  pd_ggs_stage_iters: the engine's own (2 k, k, k, k, 2 k) stage table."""
    g = golden["ggs"]
    engine.set_matches(0, g["kp1"], g["kp2"], g["i12"], tuple(int(v) for v in g["img_shape"]))
    cfg = make_ggs_cfg(synth.GGS_CFG, iter_num=10, wgs_per_seq=1, reserved=LANE)
    x0 = torch.from_numpy(g["x0"]).to(DEV)
    engine.ggs_guide(x0, 3, cfg)                                       # warm
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    engine.ggs_guide(x0, 3, cfg)
    e1.record()
    st, khz = engine.ggs_launch_stamps(2)
    torch.cuda.synchronize()
    engine.check_async()
    a = st.cpu().numpy()
    assert khz > 1000 and a[0, 0] > 0 and a[0, 1] > a[0, 0], (khz, a)
    ms = (a[0, 1] - a[0, 0]) / khz
    assert 0.0 < ms <= e0.elapsed_time(e1) + 0.05, (ms, e0.elapsed_time(e1))
    out = torch.zeros(2, 2, dtype=torch.int64, device=DEV)
    st2, _ = engine.ggs_launch_stamps(2, out=out)
    torch.cuda.synchronize()
    assert st2.data_ptr() == out.data_ptr() and np.array_equal(out.cpu().numpy(), a)     # no launch in between: the same slots
    with pytest.raises(ValueError):
        engine.ggs_launch_stamps(2, out=torch.zeros(3, 2, dtype=torch.int64, device=DEV))
    it = (C.c_int * 5)()
    _lib.check(engine.lib.pd_ggs_stage_iters(C.byref(make_ggs_cfg(iter_num=7)), it), "pd_ggs_stage_iters")
    assert list(it) == [14, 7, 7, 7, 14]


def test_lane_kernel_launch_of_more_workgroups_than_cus(seeded_diffuser, engine):
    """A guided step of 640 sequences in ONE launch of pd_ggs_lane_kernel (one workgroup per sequence: 2.5 x the chip's 256 CUs, the dispatcher back-fills a CU
    as its workgroup finishes -- what `bench.py --engine-batch 512 / 768` launches; geometry_guided_sampling.py:67-172).  The slots cycle through five distinct
    sequences: EVERY slot must be bitwise the sequence run alone (one workgroup on an idle chip, an engine of its own), all 700 iterations stepped, and the
    launch repeated bitwise itself."""
    from posediffusion_amd.engine import PoseEngine
    from posediffusion_amd.host import denoiser_state
    B, N, K = 640, 20, 5
    dev = torch.device(DEV)
    diff = seeded_diffuser.to(dev)
    cfg = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=LANE)
    mds, x0s, alone = [], [], []
    for s in range(K):
        enc = synth.make_cameras(N, seed=8100 + s)
        mds.append(synth.make_matches(enc, 224, 224, per_pair=300, seed=8100 + s))
        x0s.append(synth.perturb_pose(enc, seed=40 + s))
        engine.set_matches(0, mds[s]["kp1"], mds[s]["kp2"], mds[s]["i12"], mds[s]["img_shape"])
        o, st = engine.ggs_guide(x0s[s].to(dev), 0, cfg)
        engine.check_async()
        assert float(st[0, :, 1].sum()) == 700.0
        alone.append((o[0].clone(), st[0].clone()))
    big = PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)
    try:
        for b in range(B):
            md = mds[b % K]
            big.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        x0 = torch.cat([x0s[b % K] for b in range(B)]).to(dev)
        plan = (C.c_int * 8)()
        _lib.check(big.lib.pd_debug_ggs_plan(big._h, B, N, C.byref(cfg), plan), "pd_debug_ggs_plan")
        assert plan[0] == 1 and plan[6] == 1, list(plan)
        o1, st1 = big.ggs_guide(x0, 0, cfg)
        big.check_async()
        o2, st2 = big.ggs_guide(x0, 0, cfg)
        big.check_async()
        assert torch.equal(o1, o2) and torch.equal(st1, st2)
        bad = [b for b in range(B) if not (torch.equal(o1[b], alone[b % K][0]) and torch.equal(st1[b], alone[b % K][1]))]
        assert not bad, (len(bad), bad[:16])
    finally:
        big.close()
