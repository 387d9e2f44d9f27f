"""how much does chip load slow the launch's slowest problem?  launch time of sub-batches that all contain the bench batch's slowest problem"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 4096
d = make_batch(B, N, cfg_id=2)
s = BatchMPC(N=N)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
it = o["iters"].cpu().numpy()
order = np.argsort(-it, kind="stable")
print("slowest:", it[order[:5]])
def t(idx, steps=20):
    dev = {k: torch.as_tensor(d[k][idx], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    oo = None
    for _ in range(3): oo = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=oo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): oo = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=oo)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps
rest = order[1:][::-1]   # easiest first
for K in (1, 64, 256, 512, 1024, 1536, 2048, 3072, 4096):
    idx = np.concatenate([order[:1], rest[:K - 1]])
    print("K=%4d (slowest + %4d easiest): %.4f ms" % (K, K - 1, t(idx)))
for K in (1024, 2048):
    idx = order[:K]
    print("K=%4d hardest: %.4f ms" % (K, t(idx)))
