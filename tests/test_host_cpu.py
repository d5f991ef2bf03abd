"""CPU: host logic, the drop-in interface contract, and the C-ABI surface (no compute calls)."""
import ctypes
import functools
import os
import re

import numpy as np
import pytest
import torch

from oracle import pd_oracle as O
from posediffusion_amd import _lib, host, schedule, shard, synth


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.isfile(_lib.LIB_PATH), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    header = open(_lib.HEADER_PATH).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.pd_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.pd_version()


def test_struct_layouts_match_header():
    # 10 int32 + 6 pointers + 16 layers x 12 pointers + 11 pointers
    assert ctypes.sizeof(_lib.pd_layer_weights) == 12 * 8
    assert ctypes.sizeof(_lib.pd_weights) == 10 * 4 + 6 * 8 + 16 * 12 * 8 + 11 * 8
    assert ctypes.sizeof(_lib.pd_ggs_cfg) == 32
    # the flag / option constants Python uses are the header's
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pd_engine.h")).read()
    for name in ("PD_GGS_CFG_FORCE_ONE_HOP", "PD_GGS_CFG_NO_LDS_STAGING", "PD_GGS_CFG_WAVES8", "PD_GGS_CFG_LANE_ITEMS",
                 "PD_GGS_CFG_NO_LANE_ITEMS", "PD_OPT_DENOISER_SPLIT", "PD_WEIGHTS_PRED_X0"):
        m = re.search(r"#define\s+" + name + r"\s+(\d+)", hdr)
        assert m and int(m.group(1)) == getattr(_lib, name), name


def test_engine_fails_loudly_without_gpu(seeded_diffuser):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="GPU"):
        host.get_engine(seeded_diffuser.model, seeded_diffuser, 1, 20)
    x, z = torch.zeros(1, 5, 9), torch.zeros(1, 5, 384)
    with pytest.raises(RuntimeError, match="no CPU fallback|GPU"):
        seeded_diffuser.model(x, torch.zeros(1, dtype=torch.long), z)
    with pytest.raises(RuntimeError):
        seeded_diffuser.sample([1, 5, 9], z)


def test_schedule_buffers_equal_oracle_and_names(seeded_diffuser):
    t = O.diffusion_tables()
    b = schedule.diffusion_buffers()
    assert tuple(b) == O.TABLE_NAMES == schedule.BUFFER_NAMES
    for n in O.TABLE_NAMES:
        assert torch.equal(b[n], t[n]) and torch.equal(getattr(seeded_diffuser, n), t[n])
    with pytest.raises(ValueError):
        schedule.make_betas("nope", 100, 1e-4, 0.1)


def test_state_dict_key_contract(seeded_diffuser):
    """Checkpoint keys of the reference (SURVEY.md section 5 'Checkpoint'): strict load must work."""
    class Wrapper(torch.nn.Module):
        def __init__(s, d):
            super().__init__()
            s.diffuser = d
    keys = set(Wrapper(seeded_diffuser).state_dict().keys())
    must = {"diffuser.betas", "diffuser.posterior_log_variance_clipped", "diffuser.p2_loss_weight",
            "diffuser.model.time_embed.linear.0.weight", "diffuser.model.time_embed.linear.2.bias",
            "diffuser.model._first.weight", "diffuser.model._trunk.layers.7.self_attn.in_proj_weight",
            "diffuser.model._trunk.layers.0.self_attn.out_proj.bias", "diffuser.model._trunk.layers.3.linear2.weight",
            "diffuser.model._trunk.layers.3.norm2.bias", "diffuser.model._last.0.weight", "diffuser.model._last.1.bias",
            "diffuser.model._last.3.weight"}
    assert must <= keys
    assert sum(p.numel() for p in seeded_diffuser.model.parameters()) == 17_298_697
    # a checkpoint from an older pytorch3d carries the harmonic frequencies: tolerated, still strict
    sd = seeded_diffuser.model.state_dict()
    sd["pose_embed._emb_pose._frequencies"] = torch.zeros(10)
    seeded_diffuser.model.load_state_dict(sd, strict=True)


def test_dropin_registry_and_error_conventions():
    models = synth._dropin()
    for name in ("PoseDiffusionModel", "Denoiser", "TransformerEncoderWrapper", "GaussianDiffusion",
                 "MultiScaleImageFeatureExtractor"):
        assert hasattr(models, name)
    from util.camera_transform import pose_encoding_to_camera
    with pytest.raises(ValueError, match="Unknown pose encoding"):
        pose_encoding_to_camera(torch.zeros(1, 2, 9), pose_encoding_type="nope")
    d = models.GaussianDiffusion()
    with pytest.raises(NotImplementedError):
        d.p_mean_variance(torch.zeros(1, 2, 9), torch.zeros(1, dtype=torch.long), torch.zeros(1, 2, 384), clip_denoised=True)
    with pytest.raises(NotImplementedError):
        d.forward(torch.zeros(1, 2, 9))
    from posediffusion_amd.compat import instantiate, AttrDict
    cfg = {"_target_": "models.GaussianDiffusion", "beta_schedule": "custom"}
    assert isinstance(instantiate(AttrDict(cfg), _recursive_=False), models.GaussianDiffusion)
    # the reference's embedding modules called piecewise run on the HIP path only (util/embedding.py; no CPU fallback)
    from util.embedding import PoseEmbedding, TimeStepEmbedding
    te, pe = TimeStepEmbedding(), PoseEmbedding(target_dim=9)
    assert te.out_dim == 128 and pe.out_dim == 189
    assert set(te.state_dict()) == {"linear.0.weight", "linear.0.bias", "linear.2.weight", "linear.2.bias"} and not pe.state_dict()
    with pytest.raises(RuntimeError, match="only on an AMD GPU"):
        te(torch.zeros(2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="only on an AMD GPU"):
        pe(torch.zeros(1, 2, 9))
    with pytest.raises(ValueError):
        PoseEmbedding(target_dim=9, n_harmonic_functions=6)


def test_cond_fn_recognition():
    from util.geometry_guided_sampling import geometry_guided_sampling
    md, cfg = {"kp1": np.zeros((1, 2))}, {"iter_num": 3}
    p = functools.partial(geometry_guided_sampling, matches_dict=md, GGS_cfg=cfg)
    got = host.parse_ggs_cond_fn(p)
    assert got is not None and got[0] is md and got[1] == cfg
    assert host.parse_ggs_cond_fn(lambda m, t: m) is None
    assert host.parse_ggs_cond_fn(functools.partial(geometry_guided_sampling, matches_dict=md)) is None


def test_noise_draw_order_matches_reference_protocol():
    T, shape = 100, (2, 5, 9)
    for has_cond, start in [(False, 0), (True, 10)]:
        a = host.draw_noise(shape, T, "cpu", start, has_cond, generator=torch.Generator().manual_seed(7))
        init, noises = O.draw_reference_noise(shape, torch.Generator().manual_seed(7), T, start, has_cond)
        assert torch.equal(a[0], init)
        for step in range(T):
            t = T - 1 - step
            if noises[t] is None:
                assert (a[step + 1] == 0).all()
            else:
                assert torch.equal(a[step + 1], noises[t])


def test_ggs_print_lines_include_the_drop_line():
    """geometry_guided_sampling.py:104-108, :124: a GGS_optimize call that leaves through the `min_matches` break prints the drop line,
    then -- like every call -- its `t=.. | sampson=..` line.  The engine reports iterations stepped per stage; fewer than given = the break."""
    import io
    stats = torch.zeros(1, 5, 4)
    stats[0, :, 0] = torch.tensor([1.5, 0.25, 3.0, 0.125, 9.75])
    stats[0, :, 1] = torch.tensor([200.0, 100.0, 37.0, 100.0, 0.0])       # stage 2 broke after 37 iterations, stage 4 at once
    buf = io.StringIO()
    host.print_ggs_stats(stats, 7, 100, out=buf)
    one = ["t=07 | sampson=1.500000", "t=07 | sampson=0.250000", "Drop this pair because of insufficient valid matches", "t=07 | sampson=3.000000",
           "t=07 | sampson=0.125000", "Drop this pair because of insufficient valid matches", "t=07 | sampson=9.750000"]
    assert buf.getvalue().splitlines() == one                              # B = 1: the reference's lines exactly
    # a batch (the list-of-matches extension): every sequence's lines, prefixed -- a break in sequence 1 must not go unseen
    two = torch.cat([stats, stats])
    two[1, :, 1] = torch.tensor([200.0, 100.0, 100.0, 100.0, 200.0])
    two[1, 1, 1] = 5.0
    buf = io.StringIO()
    host.print_ggs_stats(two, 7, 100, out=buf)
    lines = buf.getvalue().splitlines()
    assert lines[:7] == ["[0] " + l for l in one]
    assert lines[7:] == ["[1] t=07 | sampson=1.500000", "[1] Drop this pair because of insufficient valid matches", "[1] t=07 | sampson=0.250000",
                         "[1] t=07 | sampson=3.000000", "[1] t=07 | sampson=0.125000", "[1] t=07 | sampson=9.750000"]
    # the iterations given come from the engine's own stage table (pd_ggs_stage_iters: no GPU needed)
    assert host.ggs_stage_iters(100) == (200, 100, 100, 100, 200) and host.ggs_stage_iters(3) == (6, 3, 3, 3, 6)


def test_partition_covers_everything():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard.partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_matches_are_consistent():
    enc = synth.make_cameras(6, seed=3)
    md = synth.make_matches(enc, 224, 224, per_pair=50, outlier_frac=0.0, noise_px=0.0, seed=3)
    assert md["kp1"].dtype == np.float64 and md["i12"].dtype == np.int64 and md["kp1"].shape == (15 * 50, 2)
    pm = O.prepare_matches(md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    v, _ = O.compute_sampson_distance(torch.from_numpy(enc[None]), pm)      # fp64, exact cameras
    assert len(v) == 750 and v.max().item() < 1e-16
    wild = np.random.default_rng(0).normal(0, 20, (6, 9))
    md2 = synth.make_epipolar_matches(wild, 224, 224, 50, noise_px=0.0, outlier_frac=0.0, seed=1)
    pm2 = O.prepare_matches(md2["kp1"], md2["kp2"], md2["i12"], md2["img_shape"])
    v2, _ = O.compute_sampson_distance(torch.from_numpy(wild[None]), pm2)
    assert len(v2) == 750 and v2.max().item() < 1e-12


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|importlib\.import_module\(\s*[\"']oracle", re.M)
    offenders = []
    for base, _, files in list(os.walk(os.path.join(root, "posediffusion_amd"))) + list(os.walk(os.path.join(root, "tools"))):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="replace").read()
                if pat.search(txt) or "oracle/_ref" in txt or "sys.path" in txt and "oracle" in txt:
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders
    # bench.py itself never imports the oracle; bench_legs.py only inside cpu_baseline()
    assert not pat.search(open(os.path.join(root, "bench.py")).read())
    legs = open(os.path.join(root, "bench_legs.py")).read()
    uses = [m.start() for m in pat.finditer(legs)]
    lo = legs.index("def cpu_baseline")
    hi = legs.index("\ndef ", lo + 1)
    assert uses and all(lo < u < hi for u in uses), "bench_legs.py may import the oracle only inside cpu_baseline()"


def test_bench_legs_are_callable_pieces():
    """bench.py = command line + timed region + JSON line; the legs are functions of bench_legs.py: the ones that need
    no GPU are exercised here -- the cut-rule mirror, the in-pipe launch statistics, the argument defaults."""
    import bench
    import bench_legs as L
    frac, items, waves = L.lane_stream_fraction([300] * 190)
    assert waves == [75, 75, 75, 75, 38, 38, 38, 38] and items == 504          # pd_lane_pass_cost's choice on the bench shape (csrc/pd_internal.h)
    assert abs(frac - sum(max(t - L.LANE_RESIDENT_STEPS, 0) for t in waves) * 64 * 32 / (16.0 * 57000)) < 1e-12
    assert L.lane_stream_fraction([7] * 190)[1] <= 512 and L.lane_stream_fraction([40] * 28)[0] == 0.0
    # launch stamps -> durations: {0, 0} slots (another kernel ran) and unordered pairs are dropped
    st = torch.tensor([[1000, 2600], [0, 0], [5000, 4000], [7000, 8650]], dtype=torch.int64)
    assert L.in_pipe_launches([(st, 100.0)]) == [16.0, 16.5] and L.in_pipe_launches([]) == []
    # ... and the wall time inside at least one launch: two launches that share the chip (overlapping stamps) count their common time once
    assert L.in_pipe_busy([(st, 100.0)]) == (16.25, 0) and L.in_pipe_busy([]) == (None, 0)
    st2 = torch.tensor([[1000, 4000], [1100, 4200], [5000, 6600]], dtype=torch.int64)
    busy, over = L.in_pipe_busy([(st2[:1], 100.0), (st2[1:], 100.0)])
    assert abs(busy - (32.0 + 16.0) / 3) < 1e-9 and over == 2
    a = bench.parse_args([])
    assert (a.gpus, a.steps, a.warmup, a.scaling, a.pipeline_depth, a.engine_batch) == (1, 24, 4, "strong", 3, 256)
    for leg in ("cpu_baseline", "measure_config", "per_config", "from_images", "pass_latency", "cold_single_batch", "exact_mode", "fresh_inputs",
                "headline_slots_equal_alone", "roofline_ggs", "roofline_denoiser", "rank_emulation", "stream_ceiling", "pmc_traffic"):
        assert callable(getattr(L, leg)), leg
    assert L.CPU_GGS_THREADS == 16 and L.CPU_DEN_THREADS == 8                   # fixed thread counts of the cpu_baseline (stated in its `sample`)


def test_colmap_keypoint_bookkeeping_vs_reference_fixture(golden):
    """dropin util/match_extraction.colmap_keypoint_to_pytorch3d against the reference's function run in place
    (tests/golden/preprocess.npz): COLMAP keypoints -> cropped + resized frame coordinates, (kp1, kp2, i12)."""
    import importlib
    import sys
    import posediffusion_amd
    if posediffusion_amd.DROPIN_PATH not in sys.path:
        sys.path.insert(0, posediffusion_amd.DROPIN_PATH)
    me = importlib.import_module("util.match_extraction")
    g = golden["preprocess"]
    keypoints = {i + 1: g[f"colmap_kp_{i + 1}"].copy() for i in range(3)}
    matches = {(1, 2): g["colmap_m_12"], (1, 3): None, (2, 3): g["colmap_m_23"]}
    info = {"bboxes_xyxy": g["bboxes_32"], "resized_scales": g["scales_32"]}
    kp1, kp2, i12 = me.colmap_keypoint_to_pytorch3d(matches, keypoints, info)
    assert np.array_equal(i12, g["i12"]) and kp1.dtype == np.float64
    assert np.array_equal(kp1, g["kp1"]) and np.array_equal(kp2, g["kp2"])
    assert np.array_equal(keypoints[1], g["colmap_kp_1"])                     # the caller's dict is not modified
    with pytest.raises(ImportError):
        me.extract_match(image_folder_path="x")


def test_strong_scaling_schedule_covers_every_sequence_once():
    """bench.py's strong-scaling plan (posediffusion_amd/shard.py): K steps x 64 sequences over 1/2/4/8 (and an uneven 3)
    ranks -- every (step, sequence) lands on exactly one rank, in one pass, at the rows `step_rows` says; a rank's engine
    passes hold at most `engine_batch` sequences, are equally long except the last, and are balanced: no more passes than
    ceil(S / engine_batch) (two for a run that fits one), the last one at least half as long as the others."""
    from posediffusion_amd import shard
    for world in (1, 2, 3, 4, 8):
        for K in (1, 5, 20, 24, 96):
            for EB in (64, 256):
                seen = {}
                for rank in range(world):
                    g0, g1, group, passes = shard.strong_schedule(K, 64, world, rank, EB)
                    b = g1 - g0
                    assert sum(passes) == K * b and all(p == group * b for p in passes[:-1]) and 0 < passes[-1] <= group * b
                    assert group * b <= max(EB, b)
                    want = max(-(-K * b // EB), min(2, K))
                    assert group == max(1, min(-(-K // want), max(1, EB // b))) and len(passes) == -(-K // group), (world, K, EB, passes)
                    for step in range(K):
                        p, r0, r1 = shard.step_rows(step, group, b)
                        assert r1 - r0 == b and r1 <= passes[p]
                        for q in range(b):
                            key = (step, g0 + q)
                            assert key not in seen
                            seen[key] = (rank, p, r0 + q)
                assert len(seen) == K * 64
    # the shapes the sweep behind the defaults measured (profiles/round2_inflight_sweep.txt): 20 steps on 1 / 2 / 4 / 8 GPUs
    assert shard.strong_schedule(20, 64, 1, 0, 256)[3] == [256] * 5
    assert shard.strong_schedule(20, 64, 2, 0, 256)[3] == [224, 224, 192]
    assert shard.strong_schedule(20, 64, 4, 0, 256)[3] == [160, 160]
    assert shard.strong_schedule(20, 64, 8, 0, 256)[3] == [80, 80]
    assert shard.strong_schedule(24, 64, 1, 0, 256)[3] == [256] * 6
    # bench.py --min-passes (tools/rank_shape_sweep.sh): an 8-GPU rank's 160 sequences as one pass, or as three / four shorter ones
    assert shard.strong_schedule(20, 64, 8, 3, 256, min_passes=1)[3] == [160]
    assert shard.strong_schedule(20, 64, 8, 3, 256, min_passes=3)[3] == [56, 56, 48]
    assert shard.strong_schedule(20, 64, 8, 3, 256, min_passes=4)[3] == [40] * 4
    assert shard.strong_schedule(20, 64, 1, 0, 256, min_passes=1)[3] == [256] * 5          # a full-size run is cut by the engine batch either way
    for mp in (1, 2, 3, 4):
        for world in (1, 2, 4, 8):
            g0, g1, group, passes = shard.strong_schedule(20, 64, world, world - 1, 256, min_passes=mp)
            assert sum(passes) == 20 * (g1 - g0) and max(passes) <= 256 and len(passes) >= min(mp, 20)


def test_advice_round2_host_side_guards(seeded_diffuser):
    """ADVICE.md (round 2), host side: a step shard larger than an engine pass is refused by the schedule (not by an unrelated engine
    error later); the upload cache's fingerprint sees a row permutation (the sums alone do not); a batch in which only SOME
    sequences have matches is refused instead of silently sampled without guidance."""
    from functools import partial
    with pytest.raises(ValueError, match="more than one engine pass"):
        shard.strong_schedule(20, 64, 1, 0, 32)
    enc = synth.make_cameras(6, seed=3)
    md = synth.make_matches(enc, 224, 224, per_pair=40, seed=3)
    fp = host._match_fingerprint(md)
    perm = np.random.default_rng(0).permutation(len(md["kp1"]))
    md2 = {"kp1": md["kp1"][perm], "kp2": md["kp2"][perm], "i12": md["i12"][perm], "img_shape": md["img_shape"]}
    assert host._match_fingerprint(md2) != fp and host._match_fingerprint(dict(md)) == fp
    from posediffusion_amd.dropin.util.geometry_guided_sampling import geometry_guided_sampling
    empty = {"kp1": None, "kp2": None, "i12": None, "img_shape": md["img_shape"]}
    cond = partial(geometry_guided_sampling, matches_dict=[md, empty], GGS_cfg=dict(synth.GGS_CFG))
    with pytest.raises(ValueError, match="have no matches"):
        seeded_diffuser.sample([2, 6, 9], torch.zeros(2, 6, 384), cond_fn=cond, cond_start_step=10)


def test_trace_tools_separate_the_ggs_launch_shapes(tmp_path):
    """tools/rocpd_stats.py and tools/coresident_from_trace.py produce the committed rocprofv3 evidence (profiles/round*_kernel_stats.txt,
    *_coresident.txt).  One bench run launches the GGS kernels in several shapes (256-workgroup engine passes; the cold-single-batch leg
    on 64 and 4 x 64 workgroups): the tools must analyse the engine passes alone and list the rest -- checked on a synthetic rocpd database
    (three contexts' 256-workgroup launches, two of them overlapping; cold-batch launches of both kernel families)."""
    import sqlite3
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "bench_results.db")
    db = sqlite3.connect(path)
    db.execute("create table rocpd_kernel_dispatch_x (start int, end int, kernel_id int, grid_size_x int, workgroup_size_x int)")
    db.execute("create table rocpd_info_kernel_symbol_x (id int, kernel_name text)")
    db.execute("insert into rocpd_info_kernel_symbol_x values (1, '_Z18pd_ggs_lane_kernelILi12EEv11PdGgsParamsi.kd'), "
               "(2, '_Z13pd_ggs_kernelILi5ELb0ELi8EEv11PdGgsParamsiiii.kd'), (3, '_Z17pd_ln_rows_kernelILi512ELi2EEvPKfPfiff.kd')")
    ms = 1_000_000
    t = 0
    for _ in range(4):                                             # engine passes, alone: 17 ms each
        db.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 1, ?, 512)", (t, t + 17 * ms, 256 * 512))
        t += 20 * ms
    for k in range(3):                                             # three contexts' launches issued together: they queue, 51 ms for the set
        db.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 1, ?, 512)", (t + k, t + 17 * (k + 1) * ms, 256 * 512))
    t += 60 * ms
    for _ in range(3):                                             # the cold single batch: another shape of the same kernel ...
        db.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 1, ?, 512)", (t, t + 13 * ms, 64 * 512))
        t += 20 * ms
    for _ in range(3):                                             # ... and the wave-per-item kernel on 4 x 64 workgroups
        db.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 2, ?, 512)", (t, t + 8 * ms, 256 * 512))
        t += 20 * ms
    db.execute("insert into rocpd_kernel_dispatch_x values (?, ?, 3, ?, 256)", (t, t + 6000, 1280 * 256))
    db.commit()
    db.close()
    flops = 256 * 57000 * 100.0 * 700
    co = subprocess.run([sys.executable, os.path.join(root, "tools", "coresident_from_trace.py"), path, str(flops)],
                        capture_output=True, text=True, check=True).stdout
    assert "7 launches, average duration" in co and "grid of 131072 threads" in co                  # 4 alone + the set of 3, nothing else
    assert "3 pd_ggs_kernel launches" in co and "3 pd_ggs_lane_kernel launches with a grid of 32768 threads" in co
    assert "sets of 1 overlapping launch(es): 4 sets, wall 17.000 ms" in co
    assert "sets of 3 overlapping launch(es): 1 sets, wall 51.000 ms" in co
    frac = 3 * flops / 51e-3 / 1e12 / 157.3 * 100
    assert f"{frac:.1f} % of the fp32 vector ALU peak" in co
    st = subprocess.run([sys.executable, os.path.join(root, "tools", "rocpd_stats.py"), path, "8"], capture_output=True, text=True, check=True).stdout
    lines = [ln for ln in st.splitlines() if ln.startswith("# ")]
    assert any("pd_ggs_lane_kernel" in ln and " 256 workgroups:     7 launches" in ln for ln in lines), st
    assert any("pd_ggs_lane_kernel" in ln and "  64 workgroups:     3 launches, average   13.000 ms" in ln for ln in lines), st
    assert any("pd_ggs_kernel" in ln and " 256 workgroups:     3 launches, average    8.000 ms" in ln for ln in lines), st
    assert st.splitlines()[1].startswith("_Z18pd_ggs_lane_kernel")                                   # the table itself: by total time


def test_fused_attention_block_mapping_covers_every_group_and_head_once():
    """pd_qkv_attn_kernel's block -> (sequence group, head) mapping (csrc/pd_qkv_attn.h, PD_QA_XCD_MAP = 1, restated here): in chunks of 16 blocks the XCDs
    0 - 3 (= block % 8) host heads {0, 1}, XCDs 4 - 7 heads {2, 3}, XCD x the groups = x mod 4; the groups past the last whole chunk keep the neighbour
    mapping.  Every (group, head) must be some block's, exactly once, for any number of groups; inside whole chunks a block's XCD decides its head pair."""
    for ngrp in list(range(1, 40)) + [64, 65, 103]:
        seen = set()
        for b in range(4 * ngrp):
            head, grp = b & 3, b >> 2
            c, r = b >> 4, b & 15
            if 4 * c + 4 <= ngrp:
                head, grp = 2 * ((r & 7) >> 2) + (r >> 3), 4 * c + (r & 3)
                assert head >> 1 == (b % 8) >> 2 and grp % 4 == (b % 8) % 4
            assert 0 <= head < 4 and 0 <= grp < ngrp
            seen.add((grp, head))
        assert len(seen) == 4 * ngrp, ngrp
