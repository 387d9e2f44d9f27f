"""Diagnostic: latency of single problems of a seeded batch, each launched alone (B = 1) and as the first 32 / 256 of the start order.  usage: one_probe.py seed idx [idx ...]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
seed = int(sys.argv[1]); idx = [int(x) for x in sys.argv[2:]]
N, B = 20, 4096
d = make_batch(B, N, cfg_id=2, seed=seed)
s = BatchMPC(N=N)
def t(sel, reps=20):
    dev = {k: torch.as_tensor(d[k][sel], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, o
for b in idx:
    ms, o = t(np.array([b]))
    print("problem %d alone: %.4f ms, iters %d" % (b, ms, int(o["iters"][0])))
ms, o = t(np.arange(B))
it = o["iters"].cpu().numpy()
print("full batch: %.4f ms; slowest by iterations:" % ms, np.argsort(-it)[:6], it[np.argsort(-it)[:6]])
for k in (64, 512, 2048):
    sel = np.argsort(-it, kind="stable")[:k]
    print("the %d longest alone: %.4f ms" % (k, t(sel)[0]))
