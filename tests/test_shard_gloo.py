"""CPU, world_size = 2 over gloo: the N > 1 path = shard independent sequences, no data-path collective,
one final all_gather (posediffusion_amd/shard.py).  The per-rank "engine" here is the CPU oracle on a tiny
2-layer model so the test runs in seconds; results must not depend on the world size."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_work(g0, g1):
    """Deterministic per-sequence result (a few oracle diffusion steps); depends only on the global index."""
    from oracle import pd_oracle as O
    from posediffusion_amd import synth
    torch.set_num_threads(1)
    diff = synth.make_diffuser(seed=0, num_layers=2)
    sd = O.cast_state_dict(diff.model.state_dict(), torch.float32)
    tables = O.diffusion_tables()
    out = []
    for gidx in range(g0, g1):
        z = synth.make_z(1, 4, seed=1000 + gidx)
        x = torch.randn(1, 4, 9, generator=torch.Generator().manual_seed(gidx))
        for t in (99, 98):
            x, _ = _p_sample2(O, sd, tables, x, t, z, gidx)
        out.append(x)
    return torch.cat(out) if out else torch.zeros(0, 4, 9)


def _p_sample2(O, sd, tables, x, t, z, gidx):
    tt = torch.full((1,), t, dtype=torch.long)
    eps = O.denoiser_forward(sd, x, tt, z, num_layers=2)
    x0 = tables["sqrt_recip_alphas_cumprod"][t] * x - tables["sqrt_recipm1_alphas_cumprod"][t] * eps
    mean = tables["posterior_mean_coef1"][t] * x0 + tables["posterior_mean_coef2"][t] * x
    nz = torch.randn(1, 4, 9, generator=torch.Generator().manual_seed(100 * gidx + t))
    return mean + torch.exp(0.5 * tables["posterior_log_variance_clipped"][t]) * nz, x0


def _worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from posediffusion_amd import shard
    r, w, _ = shard.init_distributed("gloo")
    assert (r, w) == (rank, world)
    g0, g1 = shard.partition(total, world, rank)
    local = _local_work(g0, g1)
    shard.barrier()
    full = shard.gather_poses(local, total)
    tmax = shard.max_over_ranks(float(rank + 1), "cpu")
    if rank == 0:
        q.put((full.clone(), tmax))
    shard.barrier()                      # nobody tears the group down while a peer is still inside a collective
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    total = 5                                  # odd on purpose: ranks hold 3 and 2 sequences
    ref = _local_work(0, total)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    # generous: on a cold box each spawned interpreter pages torch in first (1-2 minutes), two of them at once
    full, tmax = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    assert full.shape == (total, 4, 9)
    # same arithmetic per sequence on every rank; only BLAS threading may differ between processes (this test is about
    # the partition / gather order, so the tolerance is generous; a wrong order is off by O(1))
    assert torch.allclose(full, ref, rtol=1e-4, atol=1e-5), "sharded result differs from the single-process result"
    assert tmax == 2.0


# ------------------------------------------------------------------------------------------------ real engine, two ranks
def _engine_work(g0, g1, n_frames=6):
    """The product path per rank: a PoseEngine on cuda:0, guided sampling of the global sequences g0..g1-1 (inputs seeded by
    GLOBAL index, so the result of a sequence cannot depend on which rank owns it)."""
    from posediffusion_amd import synth
    from posediffusion_amd.engine import make_ggs_cfg
    from posediffusion_amd.host import draw_noise, get_engine
    dev = torch.device("cuda:0")
    diff = synth.make_diffuser(seed=0)
    synth.randomize_norm_and_bias_(diff.model)
    diff = diff.to(dev)
    B = g1 - g0
    if B == 0:
        return torch.zeros(0, n_frames, 9)
    eng = get_engine(diff.model, diff, B, n_frames)
    z = torch.cat([synth.make_z(1, n_frames, seed=1000 + g) for g in range(g0, g1)]).to(dev)
    noise = torch.stack([draw_noise((n_frames, 9), 100, dev, 3, True, generator=torch.Generator(device=dev).manual_seed(50 + g))
                         for g in range(g0, g1)], dim=1)
    for b, g in enumerate(range(g0, g1)):
        md = synth.make_matches(synth.make_cameras(n_frames, seed=300 + g), 224, 224, per_pair=40 + 5 * (g % 3), seed=300 + g)
        eng.set_matches(b, md["kp1"], md["kp2"], md["i12"], md["img_shape"])
    cfg = make_ggs_cfg(dict(synth.GGS_CFG, iter_num=6))     # (random-weight poses: most optimisations exit early, as the reference would)
    pose, _, _ = eng.sample(z, noise, 3, cfg, use_graph=True, want_process=False)
    eng.check_async()
    return pose.cpu()


def _engine_worker(rank, world, port, total, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from posediffusion_amd import shard
    shard.init_distributed("gloo")       # two ranks share the one GPU of the box: RCCL needs a device per rank, the gather runs on gloo
    g0, g1 = shard.partition(total, world, rank)
    local = _engine_work(g0, g1)
    full = shard.gather_poses(local, total)
    if rank == 0:
        q.put(full.clone())
    shard.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_ranks_with_the_real_engine_are_world_size_independent():
    """The N > 1 path with the HIP engine on each rank (two processes on one MI355X): shard.partition + per-rank guided
    sampling + ONE final gather must reproduce the single-process poses.  Token counts stay inside one MFMA row tile on
    every rank, so the denoiser takes the same tiling everywhere and equality is exact."""
    total = 5
    ref = _engine_work(0, total)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    assert full.shape == ref.shape and torch.isfinite(ref).all()
    assert torch.equal(full, ref), f"differing sequences {(full != ref).flatten(1).any(1).tolist()}, max abs difference {(full - ref).abs().max().item():.3e}"
