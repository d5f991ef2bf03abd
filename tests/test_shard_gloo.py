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
    full, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full.shape == (total, 4, 9)
    # same arithmetic per sequence on every rank; only BLAS threading may differ between processes (this test is about
    # the partition / gather order, so the tolerance is generous; a wrong order is off by O(1))
    assert torch.allclose(full, ref, rtol=1e-4, atol=1e-5), "sharded result differs from the single-process result"
    assert tmax == 2.0
