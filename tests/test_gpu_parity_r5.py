"""GPU (-m gpu), round 5 ("Next round" items 1, 2, 6): the kernel the headline is made of -- pd_ggs_lane_kernel<14> --
AT THE HEADLINE LAUNCH (256 workgroups x 190 pairs x 300 matches: 256 LDS rings competing for the fabric) and at the 160- / 80-sequence
rank shapes of the multi-GPU run, against the oracle and against every compared sequence run ALONE on the same kernel; the free-running
criterion at configs[2]'s real size on all three seeds; the step-invariant part of `_first` hoisted out of the diffusion steps; the
noise draw that writes straight into its slots.  Everything goes through the C-ABI.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pose_err, rel_err
from oracle import pd_oracle as O
from posediffusion_amd import _lib, synth
from posediffusion_amd.engine import PoseEngine, make_ggs_cfg
from posediffusion_amd.host import denoiser_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-5          # asserted teacher-forced bound (contract 1e-4 relative, BASELINE.json north_star)
LANE, NOLANE = _lib.PD_GGS_CFG_LANE_ITEMS, _lib.PD_GGS_CFG_NO_LANE_ITEMS
N20, PER_PAIR = 20, 300


def _engine(diff, B, N):
    dev = torch.device(DEV)
    diff = diff.to(dev)
    return PoseEngine(denoiser_state(diff.model), {k: v for k, v in diff.named_buffers(recurse=False)}, device=dev, max_B=B, max_N=N)


def _plan(eng, B, N, cfg):
    plan = (C.c_int * 8)()
    c = cfg if isinstance(cfg, _lib.pd_ggs_cfg) else make_ggs_cfg(cfg)
    _lib.check(eng.lib.pd_debug_ggs_plan(eng._h, B, N, C.byref(c), plan), "pd_debug_ggs_plan")
    return list(plan)


@pytest.fixture(scope="module")
def headline_batch(seeded_diffuser):
    """256 DISTINCT sequences of the bench's size (20 frames, 190 pairs x 300 = 57 000 matches, 224^2), uploaded once to an engine of
    256 slots; the 160- and 80-sequence rank shapes launch its first slots."""
    B = 256
    eng = _engine(seeded_diffuser, B, N20)
    mds, x0s = [], []
    for b in range(B):
        enc = synth.make_cameras(N20, seed=6000 + b)
        md = synth.make_matches(enc, 224, 224, per_pair=PER_PAIR, seed=6000 + b)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        mds.append(md)
        x0s.append(synth.perturb_pose(enc, seed=300 + b))
    yield eng, mds, torch.cat(x0s)
    eng.close()


@pytest.mark.parametrize("B", [256, 160, 80])
def test_lane_kernel_at_the_headline_launch(engine, headline_batch, B):
    """geometry_guided_sampling.py:67-172 on pd_ggs_lane_kernel<14> at bench.py's launch: B workgroups (one per sequence; 256 = every CU),
    each streaming its own 57 000 matches through a per-wave LDS ring fed by hand-issued LDS-DMA whose only guard between the DMA write
    and the ds_read of a slot is a counted `s_waitcnt vmcnt` (pd_ggs_lane.inc).  What only this launch has -- B rings competing for the
    fabric, DMA landing latencies several times those of a 3-sequence launch -- is what a miscounted wait would need to corrupt a slot.
      (a) 20 iterations of GGS_optimize (the ring wraps ~19 x 20 times per wave): nine slots spread over all XCDs are compared BITWISE
          with the same sequence run alone (a launch of ONE workgroup on an idle chip: DMA latency at its minimum); against the oracle's
          GGS_optimize the teacher-forced bound 2e-5 is asserted after 6 iterations of the same launch shape (measured 1e-6); the deviation
          after the 20 is printed only -- a free-running trajectory through a hard threshold, and the CPU oracle's own sums depend on the
          box's thread count: one slot read 8.1e-5 on one box and 1.6e-4 on another with the engine's bits unchanged;
      (b) a full geometry_guided_sampling (5 stages, 700 iterations: ~13 000 ring turns per wave): the same nine slots bitwise with the
          sequence run alone, every slot finite with all 700 iterations stepped, and the whole launch repeated: bitwise the same."""
    eng, mds, x0_all = headline_batch
    x0 = x0_all[:B].to(DEV)
    cfg_s = make_ggs_cfg(iter_num=10, wgs_per_seq=1, reserved=LANE)
    cfg_f = make_ggs_cfg(synth.GGS_CFG, wgs_per_seq=1, reserved=LANE)
    plan = _plan(eng, B, N20, cfg_f)
    assert plan[0] == 1 and plan[6] == 1, plan                    # one workgroup per sequence on the lane-per-item kernel
    o20, st20, _ = eng.ggs_optimize(x0, cfg=cfg_s)
    eng.check_async()
    o6, st6, _ = eng.ggs_optimize(x0, cfg=make_ggs_cfg(iter_num=3, wgs_per_seq=1, reserved=LANE))
    eng.check_async()
    of, stf = eng.ggs_guide(x0, 0, cfg_f)
    eng.check_async()
    of2, stf2 = eng.ggs_guide(x0, 0, cfg_f)
    eng.check_async()
    assert torch.equal(of, of2) and torch.equal(stf, stf2), "the same launch twice must give the same bits"
    assert torch.isfinite(of).all() and torch.isfinite(o20).all()
    assert (st20[:, 1] == 20).all(), st20[:, 1]
    assert (stf[:, :, 1].sum(dim=1) == 700).all(), "every sequence of the launch must step its 700 iterations"
    slots = sorted({(33 * i) % B for i in range(8)} | {B - 1})     # 0, 33, 66, ...: block b runs on XCD b % 8 -> all eight XCDs
    assert {b % 8 for b in slots} == set(range(8))
    assert _plan(engine, 1, N20, cfg_f)[6] == 1
    worst = worst6 = 0.0
    for b in slots:
        md = mds[b]
        engine.set_matches(0, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        a20, sa20, _ = engine.ggs_optimize(x0[b:b + 1], cfg=cfg_s)
        af, saf = engine.ggs_guide(x0[b:b + 1], 0, cfg_f)
        engine.check_async()
        assert torch.equal(a20[0], o20[b]) and torch.equal(sa20[0], st20[b]), f"slot {b} of the {B}-sequence launch differs from the sequence run alone (20 iterations)"
        assert torch.equal(af[0], of[b]) and torch.equal(saf[0], stf[b]), f"slot {b} of the {B}-sequence launch differs from the sequence run alone (700 iterations)"
        pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
        ref, _, steps = O.ggs_optimize(x0[b:b + 1].cpu().clone(), pm, iter_num=10)
        assert steps == 20
        worst = max(worst, rel_err(o20[b:b + 1], ref))
        ref6, _, steps6 = O.ggs_optimize(x0[b:b + 1].cpu().clone(), pm, iter_num=3)
        assert steps6 == 6 == int(st6[b, 1])
        worst6 = max(worst6, pose_err(o6[b:b + 1], ref6, f"lane_headline_launch_{B}_6_iterations"))
    print(f"lane kernel, {B}-sequence launch: slots {slots} bitwise = alone (20 and 700 iterations); worst deviation from the oracle after 6 / 20 iterations "
          f"{worst6:.2e} / {worst:.2e}")
    assert worst6 < TOL, (worst6, worst)      # (the 20-iteration figure is printed, not asserted: see the docstring)


def test_free_running_full_size_seeds_1_and_2(engine, golden):
    """SURVEY 8c's free-running criterion at configs[2]'s real size (N = 20, 57 000 matches, 100 steps, 10 guided x 700 iterations) on the
    two seeds round 4 generated and dropped for size (fixture guided_free_full_s12: the unmodified reference in fp32 and the fp64 oracle per
    seed; its matches are regenerated from the stored model mean + seed and checked against the stored sha256).  Both kernel families, per
    seed: pose deviation from fp64 <= 2 x the reference-fp32's own; final mean Sampson gap <= max(1 %, 2 x the reference's own gap) for the lane
    kernel, <= 2 x the largest reference gap for the wave-per-item kernels (see below)."""
    from oracle.make_golden import regenerate_matches
    from test_gpu_parity_r2 import _free_running_case
    g = dict(golden["guided_free_full_s12"])
    assert int(g["cond_start_step"]) == 10 and g["seeds"].tolist() == [1, 2]
    rows = {}
    for s in g["seeds"].tolist():
        md = regenerate_matches(g, s)
        assert len(md["kp1"]) == 57000 and int(g[f"s{s}_ref_optimize_calls"]) == 50
        g.update({f"s{s}_kp1": md["kp1"], f"s{s}_kp2": md["kp2"], f"s{s}_i12": md["i12"]})
        for tag, cfg in (("lane", dict(synth.GGS_CFG, wgs_per_seq=1, reserved=LANE)), ("wave", dict(synth.GGS_CFG))):
            if tag == "lane":
                assert _plan(engine, 1, N20, cfg)[6] == 1
            rows[(s, tag)] = _free_running_case(engine, g, s, cfg)
    print("free-running GGS-on, configs[2] full size, seeds 1 and 2: (seed, kernel) -> (engine dev, reference dev, engine Sampson gap, reference gap):",
          {k: tuple(f"{v:.3e}" for v in r) for k, r in rows.items()})
    worst_ref_gap = max(r[3] for r in rows.values())
    for (s, tag), (dev, ref_dev, gap, ref_gap) in rows.items():
        assert dev <= 2.0 * ref_dev, (s, tag, dev, ref_dev)
        # the final mean Sampson error is a chaotic statistic of 7 000 iterations through a hard threshold (the reference's own fp32-vs-fp64 gap
        # on the three full-size seeds: 27 %, 15 %, 25 %): the lane kernel -- the throughput default -- is held to the per-seed bound
        # (measured 8 %, 23 %), the wave-per-item kernels (38 %, 27 % here; 22 % on seed 0) to 2 x the largest reference gap, as in round 3
        assert gap <= max(0.01, 2.0 * (ref_gap if tag == "lane" else worst_ref_gap)), (s, tag, gap, ref_gap)


def test_noise_drawn_straight_into_its_slots():
    """host.draw_noise writes randn into the slots of one [T + 1, ...] tensor (`out=`): on the GPU generator too that must be, bit for
    bit, the reference's sequence of separate randn calls (gaussian_diffuser.py:289, :276-278)."""
    from posediffusion_amd.host import draw_noise
    for shape, start, has_cond in (((1, 20, 9), 10, True), ((3, 7, 9), 0, False), ((64, 20, 9), 10, True)):
        a = draw_noise(shape, 100, DEV, start, has_cond, generator=torch.Generator(device=DEV).manual_seed(11))
        g = torch.Generator(device=DEV).manual_seed(11)
        ref = [torch.randn(shape, device=DEV, generator=g)]
        for step in range(100):
            t = 99 - step
            guided = has_cond and t < start
            ref.append(torch.randn(shape, device=DEV, generator=g) if (not guided and t > 0) else torch.zeros(shape, device=DEV))
        assert torch.equal(a, torch.stack(ref)), (shape, start, has_cond)
    torch.manual_seed(3)
    a = draw_noise((1, 20, 9), 100, DEV)
    torch.manual_seed(3)
    assert torch.equal(a[0], torch.randn((1, 20, 9), device=DEV)) and torch.equal(a[1], torch.randn((1, 20, 9), device=DEV))


@pytest.mark.parametrize("B,N", [(256, 20), (103, 20), (40, 32), (160, 7), (50, 24), (64, 17)])
def test_fused_qkv_attention_is_bitwise_the_two_launch_path(seeded_diffuser, oracle_weights, B, N):
    """models/denoiser.py:88-97 (in_proj + attention of the encoder layers) in the fp16-plane mode: pd_qkv_attn_kernel -- one workgroup per
    (group of 95 // N whole sequences, head), Q / K / V only ever in LDS -- against the two launches it replaces (pd_gemm_strip_kernel ->
    fp32 QKV in memory -> pd_attn_mma_kernel): the same sums in the same order, so a whole denoiser evaluation must agree BIT FOR BIT, at the
    bench's 5 120 rows, with a ragged last group (103 = 25 x 4 + 3), two sequences of 32 frames per workgroup (one row tile idle), thirteen
    of 7 (five rounds of the three attention teams), three of 24 and five of 17 (85 rows); and within the teacher-forced bound of the fp64
    oracle on a few sequences (first / last group, a ragged one)."""
    eng = _engine(seeded_diffuser, B, N)
    assert eng.get_option(_lib.PD_OPT_DENOISER_SPLIT) == 2 and eng.get_option(_lib.PD_OPT_DENOISER_FUSED_ATTN) == 1
    g = torch.Generator().manual_seed(7 * B + N)
    x, z = torch.randn(B, N, 9, generator=g), synth.make_z(B, N, seed=B + N)
    sd64 = {k: v.double() for k, v in oracle_weights.items()}
    G = 95 // N
    sub = sorted({0, 1, G - 1, G, B // 2, B - G - 1, B - 2, B - 1})
    for t in (99, 40, 0):
        eng.set_option(_lib.PD_OPT_DENOISER_FUSED_ATTN, 2)          # (2 = always; the default 1 takes the fused kernel only where it fills the chip)
        fused = eng.denoise(x.to(DEV), z.to(DEV), t)
        eng.set_option(_lib.PD_OPT_DENOISER_FUSED_ATTN, 0)
        plain = eng.denoise(x.to(DEV), z.to(DEV), t)
        assert torch.isfinite(fused).all()
        bad = (fused != plain).reshape(B, -1).any(dim=1).nonzero().flatten().tolist()
        assert not bad, f"t={t}: sequences {bad[:12]} (of {len(bad)}) differ between the fused and the two-launch attention"
        with torch.no_grad():
            ref = O.denoiser_forward(sd64, x[sub].double(), torch.full((len(sub),), t, dtype=torch.long), z[sub].double())
        worst = max(rel_err(fused[s], ref[i]) for i, s in enumerate(sub))
        assert worst < 3e-6, (t, worst)
    eng.set_option(_lib.PD_OPT_DENOISER_FUSED_ATTN, 2)
    # the option is part of the graph key: a sampling pass replayed from its graph must follow the switch
    noise = torch.randn(101, B, N, 9, generator=g).to(DEV)
    p1 = eng.sample(z.to(DEV), noise, 0, None, use_graph=True, want_process=False)[0].clone()
    eng.set_option(_lib.PD_OPT_DENOISER_FUSED_ATTN, 0)
    p0 = eng.sample(z.to(DEV), noise, 0, None, use_graph=True, want_process=False)[0].clone()
    pe = eng.sample(z.to(DEV), noise, 0, None, use_graph=False, want_process=False)[0]
    assert torch.equal(p1, p0) and torch.equal(p0, pe)
    eng.close()


@pytest.mark.parametrize("B,N", [(1, 20), (3, 7), (64, 20)])
def test_first_layer_hoist_matches_stepwise_and_the_oracle(seeded_diffuser, oracle_weights, B, N):
    """models/denoiser.py:56-70: `_first` reads [pose embedding | t_emb | z | pivot]; z and t_emb do not depend on the sample, so the
    engine evaluates their columns outside the loop (z W_z^T + b once per sampling call, W_t t_emb(t) as a table built at creation) and
    only the 192-column pose piece per step.  The sampling loop (z piece prepared once) and the step-level API (prepared per call) agree
    along a teacher-forced trajectory, a second z through the same engine leaves nothing behind (a stale z piece would show), and the
    step agrees with the fp64 oracle at the teacher-forced bound."""
    eng = _engine(seeded_diffuser, B, N)
    sd64 = {k: v.double() for k, v in oracle_weights.items()}
    for seed in (1, 2):
        g = torch.Generator().manual_seed(100 * seed + B + N)
        z = synth.make_z(B, N, seed=40 + seed).to(DEV)
        noise = torch.randn(101, B, N, 9, generator=g).to(DEV)
        _, process, _ = eng.sample(z, noise, 0, None, use_graph=True)
        process = process.clone()
        z_other = synth.make_z(B, N, seed=90 + seed).to(DEV)
        for step in (0, 1, 50, 98, 99):
            t = 99 - step
            mean, _ = eng.p_mean(process[step], z, t)
            nxt = eng.p_finish(mean, noise[step + 1] if t > 0 else None, t)
            assert pose_err(nxt, process[step + 1], "first_hoist_step_api") < 1e-6, (seed, step, rel_err(nxt, process[step + 1]))     # (the loop's update is fused into the tail kernel)
            other, _ = eng.p_mean(process[step], z_other, t)             # another z through the same engine ...
            again, _ = eng.p_mean(process[step], z, t)                   # ... must not leave its z piece behind
            assert torch.equal(again, mean) and not torch.equal(other, mean), (seed, step)
        _, process2, _ = eng.sample(z, noise, 0, None, use_graph=True)
        assert torch.equal(process2, process), seed
        for t in (99, 17, 0):
            xs = process[99 - t].cpu()
            eps = eng.denoise(xs.to(DEV), z, t)
            with torch.no_grad():
                ref = O.denoiser_forward(sd64, xs.double(), torch.full((B,), t, dtype=torch.long), z.cpu().double())
            assert rel_err(eps, ref) < TOL, (seed, t, rel_err(eps, ref))
    eng.close()
