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


# ------------------------------------------------------------------------------------------------------------------
# Adaptive step control over a sharded batch (SURVEY.md 8(e) option B): one all-reduced error norm per step attempt
# makes every rank follow the step sequence of the single-process solve.
# ------------------------------------------------------------------------------------------------------------------
class _CpuOps:
    """Stand-in for HipStateOps on CPU tensors (test infrastructure): same interface, same group reduction
    (uspace_amd.odeint.allreduce_mean_square is the product code under test)."""

    def __init__(self, group=None):
        self.group = group

    def prepare(self, y):
        return y.detach().to(torch.float64).contiguous()

    def combine(self, y, ks, coefs):
        out = y.clone()
        for k, c in zip(ks, coefs):
            out = out + float(c) * k
        return out

    def scaled_norm(self, y0, y1, ks, coefs, rtol, atol):
        from uspace_amd.odeint import allreduce_mean_square
        err = torch.zeros_like(y0)
        for k, c in zip(ks, coefs):
            err = err + float(c) * k
        q = (err / (atol + rtol * torch.maximum(y0.abs(), y1.abs()))) ** 2
        if self.group is None:
            return float(q.mean().sqrt()) if q.numel() else 0.0
        pair = torch.tensor([float(q.sum()), float(q.numel())], dtype=torch.float64)
        return float(allreduce_mean_square(pair, self.group)) ** 0.5


def _field(rates):
    # dy/dt = -r_b * y + sin(3 t): every trajectory has its own stiffness, so shards see different local error norms
    def f(t, y):
        return -rates.view(-1, 1) * y + float(torch.sin(torch.tensor(3.0 * t)))
    return f


def _solve(y0, rates, group, trace):
    from uspace_amd.odeint import Stats, odeint

    def f(t, y):
        trace.append(float(t))
        return _field(rates)(t, y)
    st = Stats()
    out = odeint(f, y0, 0.0, 1.0, method="dopri5", rtol=1e-6, atol=1e-6, ops=_CpuOps(group), stats=st)
    return out, st


def _adaptive_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uspace_amd.sampling import gather_batch, shard_bounds
    g = torch.Generator().manual_seed(11)
    y0 = torch.randn(n_total, 6, generator=g, dtype=torch.float64)
    rates = torch.linspace(0.5, 40.0, n_total, dtype=torch.float64)       # the last rank holds the stiff trajectories
    lo, hi = shard_bounds(n_total, world, rank)
    t_single, t_group, t_local = [], [], []
    ref, st_ref = _solve(y0, rates, None, t_single)                        # the whole batch in one process
    out, st = _solve(y0[lo:hi], rates[lo:hi], True, t_group)               # sharded, group-controlled
    loc, st_loc = _solve(y0[lo:hi], rates[lo:hi], None, t_local)           # sharded, per-rank control (option A)
    full = gather_batch(out.float(), n_total).double()
    close = lambda a, b: len(a) == len(b) and all(abs(x - y) < 1e-9 for x, y in zip(a, b))   # noqa: E731
    res = dict(rank=rank, same_steps=close(t_group, t_single), nfe=(st.nfe, st_ref.nfe, st_loc.nfe),
               acc=(st.accepted, st_ref.accepted), rej=(st.rejected, st_ref.rejected),
               err=float((full - ref).abs().max()), err_local=float((loc - ref[lo:hi]).abs().max()) if hi > lo else 0.0,
               local_differs=not close(t_local, t_single))
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])
def test_group_controlled_adaptive_solve_follows_the_single_process_step_sequence(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_adaptive_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r["same_steps"], r                      # every evaluation time equals the single-process solve's
        assert r["nfe"][0] == r["nfe"][1] and r["acc"][0] == r["acc"][1] and r["rej"][0] == r["rej"][1], r
        assert r["err"] < 1e-6, r                      # gathered result == unsharded result (fp32 gather)
    if n_total >= 5:
        # per-rank control is a different (still valid) discretisation: some rank takes other steps
        assert any(r["local_differs"] for r in res), res
        assert all(r["err_local"] < 1e-3 for r in res)
