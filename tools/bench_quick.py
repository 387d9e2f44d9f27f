"""Diagnostic: B=4096 N=20 fp64 launch time only (HIP events), for quick A/B of kernel changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])  # A/B of diagnostic builds
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("QN", 20)); B = int(os.environ.get("QB", 4096))
s = BatchMPC(N=N)
d = make_batch(B, N, cfg_id=2, seed=int(os.environ["QSEED"]) if "QSEED" in os.environ else None)
dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
o = None
for _ in range(5): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
it = o["iters"].float()
print("%s B=%d N=%d: %.4f ms/launch  %.3f Msolves/s  iters mean %.2f max %d  status!=0: %d" % (" ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("X_")), B, N, best, B / best / 1e3, it.mean().item(), int(it.max().item()), int((o["status"] != 0).sum().item())))
