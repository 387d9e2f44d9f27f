"""Diagnostic: launch time vs batch size and vs iteration-count spread (not part of the product or tests)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = 20
s = BatchMPC(N=N)
def timeit(d, reps=10):
    dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(3):
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, o["iters"].float()
full = make_batch(262144, N, cfg_id=2)
for B in (256, 1024, 2048, 4096, 8192, 16384, 65536, 262144):
    d = {k: v[:B] for k, v in full.items()}
    ms, it = timeit(d, reps=5 if B > 20000 else 10)
    print("B=%6d  %.3f ms  %.2f Msolves/s  iters mean %.2f max %d  -> %.1f ns per (problem*iteration)" % (B, ms, B / ms / 1e3, it.mean().item(), int(it.max().item()), ms * 1e6 / (B * it.mean().item())))
d = {k: np.repeat(v[:1], 4096, axis=0) for k, v in full.items()}
ms, it = timeit(d)
print("4096 copies of problem 0: %.3f ms, iters %d" % (ms, int(it.max().item())))
order = np.argsort(-timeit({k: v[:4096] for k, v in full.items()}, reps=1)[1].cpu().numpy())
d = {k: v[:4096][order] for k, v in full.items()}
ms, it = timeit(d)
print("4096 sorted by decreasing iteration count: %.3f ms" % ms)
