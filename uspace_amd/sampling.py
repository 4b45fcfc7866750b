"""Multi-GPU sampling: independent trajectories sharded over ranks, one gather at the end.

Mirrors the reference's only sampling-path collective (``accelerator.gather`` after each
mini-batch, tools/utils_uvit.py:264-277, tools/utils_vis.py:241): every rank solves its own
contiguous slice of the batch -- no data-path communication during the solve -- and the final
latents ([B/G,4,32,32] fp32, 16 KB per sample) are all-gathered once over RCCL/xGMI.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous, balanced split of n items: first (n % world) ranks get one extra."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_slice(t, world_size, rank):
    lo, hi = shard_bounds(t.shape[0], world_size, rank)
    return t[lo:hi]


def gather_batch(local, n_total, group=None):
    """All-gather variable-length shards along dim 0 and return the first n_total rows in rank order."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(n_total, world, r) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < max_rows:
        pad = torch.cat([local, local.new_zeros((max_rows - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    try:
        dist.all_gather(bufs, pad.contiguous(), group=group)
    except Exception as ex:       # a peer that died mid-solve (or the group's timeout): say which collective and who noticed
        raise RuntimeError(
            f"uspace_amd.sampling.gather_batch: the all_gather of the final latents failed on rank {dist.get_rank(group)} of {world} "
            f"({n_total} rows in all) -- a peer rank has failed or did not arrive within the process group's timeout; its own error "
            f"is in that rank's output.  [{type(ex).__name__}: {str(ex)[:200]}]") from ex
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def sharded_sample(solve_fn, z, *conds, group=None):
    """Run ``solve_fn(z_local, *conds_local)`` on this rank's slice of the global batch and gather.

    ``z`` and every tensor in ``conds`` hold the GLOBAL batch (same on every rank, e.g. drawn from
    a shared seed); per-row python lists (``target_context_ids``) can be sliced with shard_bounds.
    """
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    n = z.shape[0]
    lo, hi = shard_bounds(n, world, rank)
    parts = [c[lo:hi] if torch.is_tensor(c) else c for c in conds]
    out = solve_fn(z[lo:hi], *parts)
    return gather_batch(out, n, group)
