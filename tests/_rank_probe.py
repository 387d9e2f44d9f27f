"""child script of tests/test_host_logic.py::test_spawn_ranks_launch_path (not a test module): what one rank of `bench.py --gpus N`
does around the solve, on CPU tensors with the gloo backend -- group of the requested size, shard-local block, all-gather, rank 0 reports."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd.dist import SolutionGather, shard_range  # noqa: E402

want = int(sys.argv[1])
out = sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
assert dist.get_world_size() == want
B = 4 * world
lo, hi = shard_range(B, rank, world)
full = torch.arange(B * 2, dtype=torch.float64).reshape(B, 2)
g = SolutionGather(B)
g.submit(0, full[lo:hi].clone())
ok = bool(torch.equal(g.wait(0), full))
t = torch.tensor([1.0 if ok else 0.0])
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    with open(out, "w") as f:
        json.dump({"n_gpus": dist.get_world_size(), "ok": bool(t.item() == 1.0)}, f)
dist.destroy_process_group()
