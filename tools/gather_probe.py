"""Diagnostic: the asynchronous double-buffered all-gather of bench.py's N > 1 loop, exercised through RCCL on ONE GPU
(1-rank group, collective forced): every step's gathered block must equal that step's solution, and the loop time must not exceed
the plain loop's by more than the collective's launch cost."""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.dist import SolutionGather
from mkz_mpc_path_follower_amd.synthetic import make_batch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
N, B, steps = 20, 4096, 40
s = BatchMPC(N=N)
batches = [make_batch(B, N, cfg_id=2, seed=500 + k) for k in range(3)]
dins = [{k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")} for d in batches]
ref_u0 = [s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])["u0"].clone() for d in dins]
for force in (False, True):
    g = SolutionGather(B, force_collective=force)
    outs = [None, None]; got = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        sl = i & 1
        prev = g.wait(sl)
        if prev is not None and i >= 2: got.append((i - 2, prev.clone()))
        d = dins[i % 3]
        outs[sl] = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], out=outs[sl])
        g.submit(sl, outs[sl]["u0"])
    for i in (steps - 2, steps - 1): got.append((i, g.wait(i & 1).clone()))
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ok = all(torch.equal(u, ref_u0[i % 3]) for i, u in got)
    print("force_collective=%s: %d gathered blocks checked, all equal to the step's own solution: %s; %.3f ms per step" % (force, len(got), ok, 1e3 * el / steps))
dist.destroy_process_group()
