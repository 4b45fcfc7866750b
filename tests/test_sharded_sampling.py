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


# ------------------------------------------------------------------------------------------------------------------
# Eight ranks (the node the driver scales to; VERDICT r5 task 5): the shard sizes of BASELINE configs 4 and 5 (512 and 256 rows
# over 8 GPUs), an uneven job, the group-controlled adaptive solve, and a rank that fails mid-solve.
# ------------------------------------------------------------------------------------------------------------------
def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uspace_amd.sampling import gather_batch, shard_bounds, sharded_sample
    ok = True
    for n_total, want in ((512, [64] * 8), (256, [32] * 8), (250, [32, 32, 31, 31, 31, 31, 31, 31]), (5, [1, 1, 1, 1, 1, 0, 0, 0])):
        sizes = [shard_bounds(n_total, world, r)[1] - shard_bounds(n_total, world, r)[0] for r in range(world)]
        ok = ok and sizes == want
        g = torch.Generator().manual_seed(7)
        z = torch.randn(n_total, 4, 4, 4, generator=g)
        cond = torch.arange(n_total, dtype=torch.float32)
        seen = []

        def solve(z_local, c_local):
            seen.append(z_local.shape[0])
            return z_local * 2.0 + c_local.view(-1, 1, 1, 1) + 1000.0 * 0      # rows keep their global identity through cond
        out = sharded_sample(solve, z, cond)
        lo, hi = shard_bounds(n_total, world, rank)
        ok = ok and seen == [hi - lo] and out.shape[0] == n_total and torch.equal(out, z * 2.0 + cond.view(-1, 1, 1, 1))
        ok = ok and torch.equal(gather_batch(z[lo:hi], n_total), z)             # an empty shard gathers too
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def _run(target, world, extra=(), timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(extra) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    return procs, q


def test_eight_rank_shard_and_gather_at_the_baseline_shard_sizes():
    procs, q = _run(_worker8, 8)
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == {r: True for r in range(8)}


def test_eight_rank_group_controlled_adaptive_solve():
    """CNF.norm_group on 8 ranks: 250 trajectories (31 / 32 per rank), the stiff ones on the last ranks -- every rank follows the
    step sequence of the unsharded solve; per-rank control takes other steps on some rank."""
    procs, q = _run(_adaptive_worker, 8, extra=(250,))
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r["same_steps"] and r["nfe"][0] == r["nfe"][1] and r["acc"][0] == r["acc"][1] and r["rej"][0] == r["rej"][1], r
        assert r["err"] < 1e-6, r
    assert any(r["local_differs"] for r in res) and len({r["nfe"][2] for r in res}) > 1       # per-rank control: different NFE per rank
    assert all(r["err_local"] < 1e-3 for r in res)


def _failing_worker(rank, world, port, bad_rank, q):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))
    from uspace_amd.sampling import sharded_sample
    z = torch.randn(256, 4, 4, 4, generator=torch.Generator().manual_seed(3))
    calls = [0]

    def solve(z_local):
        for step in range(5):                                   # a rank dies in the middle of its solve, before the gather
            calls[0] += 1
            if rank == bad_rank and step == 2:
                raise ValueError(f"synthetic failure on rank {rank} at step {step}")
            z_local = z_local * 0.9
        return z_local
    import time
    t0 = time.time()
    def leave(msg, code):                                       # (flush the queue's feeder thread: os._exit would drop the message)
        q.put(msg)
        q.close()
        q.join_thread()
        os._exit(code)                                          # the process ends as a crashed rank does: no barrier, no destroy
    try:
        sharded_sample(solve, z)
        leave((rank, "returned", time.time() - t0, ""), 0)
    except ValueError as ex:
        leave((rank, "own", time.time() - t0, str(ex)), 3)
    except RuntimeError as ex:
        leave((rank, "peer", time.time() - t0, str(ex)), 4)


def test_a_rank_that_fails_mid_solve_does_not_leave_the_others_in_the_gather():
    """One of 8 ranks raises inside its solve.  The others are in the all_gather of the final latents by then: each of them must come
    back with an error that names the collective and tells where to look -- within the group's timeout (30 s here; gloo notices the
    closed connections at once), not hang.  What bench.py adds on top: the same bound on its process group (USPACE_BENCH_PG_TIMEOUT_S,
    default 600 s), a line on stderr naming the failing rank, and under torch.distributed.run the agent ends the other ranks itself."""
    procs, q = _run(_failing_worker, 8, extra=(5,))
    res = {}
    for _ in procs:
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode is not None
    assert res[5][1] == "own" and "synthetic failure on rank 5" in res[5][3]
    for r in range(8):
        if r == 5:
            continue
        kind, dt, msg = res[r][1], res[r][2], res[r][3]
        assert kind == "peer", res[r]
        assert "gather_batch" in msg and "a peer rank has failed" in msg and f"rank {r} of 8" in msg, msg
        assert dt < 60, res[r]
    assert [p.exitcode for p in procs] == [4, 4, 4, 4, 4, 3, 4, 4]
