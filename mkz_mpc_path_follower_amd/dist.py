"""Multi-GPU sharding of a batch of independent problems (one process per GPU).

The reference has no cross-problem state, so the batch shards in contiguous blocks with no
data-path collective; the only exchange is one all-gather of the [B/G, 2] (accel, steer)
solution blocks (RCCL over xGMI when the backend is "nccl", gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """contiguous block [lo, hi) of problems owned by `rank`; blocks differ by at most one problem"""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_solutions(u0_local, B, group=None):
    """u0_local [b_r, 2] on every rank -> [B, 2] on every rank, in problem order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return u0_local
    rank = dist.get_rank(group)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((B, 2), dtype=u0_local.dtype, device=u0_local.device)
        dist.all_gather_into_tensor(out, u0_local.contiguous(), group=group)
        return out
    mx = max(sizes)  # ragged shards: pad to the largest, gather, strip
    pad = torch.zeros((mx, 2), dtype=u0_local.dtype, device=u0_local.device)
    pad[:sizes[rank]] = u0_local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
