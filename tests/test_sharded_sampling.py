"""N>1 path on CPU: world_size-2 gloo processes shard a batch of independent trajectories and gather
the final latents once (the only collective on the sampling path, tools/utils_uvit.py:277)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uspace_amd.sampling import gather_batch, shard_bounds, sharded_sample
    g = torch.Generator().manual_seed(7)
    z = torch.randn(n_total, 4, 8, 8, generator=g)            # same global batch on every rank
    cond = torch.arange(n_total, dtype=torch.float32)

    def solve(z_local, c_local):                                # stand-in for score_model.decode
        assert z_local.shape[0] == c_local.shape[0]
        return z_local * 2.0 + c_local.view(-1, 1, 1, 1)

    out = sharded_sample(solve, z, cond)
    lo, hi = shard_bounds(n_total, world, rank)
    ok = torch.equal(out, z * 2.0 + cond.view(-1, 1, 1, 1)) and out.shape[0] == n_total
    ok = ok and torch.equal(gather_batch(z[lo:hi], n_total), z)
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_two_rank_shard_and_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_bounds_cover_everything():
    from uspace_amd.sampling import shard_bounds
    for n in (1, 7, 64, 256, 513):
        for w in (1, 2, 4, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    from uspace_amd.sampling import sharded_sample
    z = torch.randn(5, 3)
    assert torch.equal(sharded_sample(lambda a: a + 1, z), z + 1)
