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


class SolutionGather:
    """Double-buffered, asynchronous all-gather of the (accel, steer) blocks for a stream of batches.

    `submit(slot, u0_local)` starts the gather of one batch on the collective's own stream (RCCL: the copy engines / xGMI run it
    while the next batch's solve kernel computes); `wait(slot)` makes the current stream wait for it and returns the [B, 2]
    result.  The caller alternates its slots (two, or as many as the steps a rank may run ahead of the others) and must `wait(slot)`
    before it lets a solve overwrite the `u0_local` buffer that slot's gather reads -- `bench.py` does exactly that, so a step's solution exchange costs no time on the solve stream.
    Equal shards only (the ragged case uses `all_gather_solutions`)."""

    def __init__(self, B, group=None, slots=2, force_collective=False):
        self.B, self.group = B, group
        self.force = force_collective  # run the collective even in a 1-rank group (exercises the RCCL path on one GPU)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if self.world > 1 and B % self.world:
            raise ValueError("SolutionGather needs equal shards (B %% world == 0)")
        self.work = [None] * slots
        self.out = [None] * slots

    def submit(self, slot, u0_local):
        if self.world == 1 and not self.force:
            self.out[slot] = u0_local
            return
        if self.out[slot] is None or self.out[slot].dtype != u0_local.dtype or self.out[slot].device != u0_local.device:
            self.out[slot] = torch.empty((self.B, 2), dtype=u0_local.dtype, device=u0_local.device)
        self.work[slot] = dist.all_gather_into_tensor(self.out[slot], u0_local.contiguous(), group=self.group, async_op=True)

    def wait(self, slot):
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None
        return self.out[slot]


def spawn_ranks(script, nproc, args, master_port=None, extra_env=None, timeout=None):
    """Start `nproc` ranks of `script` on this node (one process per GPU) as a `torch.distributed.run` CHILD process and return its
    exit code.  Must be called before the calling process touches the GPU (the caller stays a plain launcher: it neither forks
    GPU state nor replaces itself).  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import os
    import socket
    import subprocess
    import sys
    if master_port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            master_port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL / cross-process tensors)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
           "--master-addr", "127.0.0.1", "--master-port", str(master_port), script] + list(args)
    return subprocess.call(cmd, env=env, timeout=timeout)
