"""Diagnostic: B=4096 N=20 fp64 launch time only (HIP events), for quick A/B of kernel changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])  # A/B of diagnostic builds
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("QN", 20)); B = int(os.environ.get("QB", 4096))
import numpy as np
F32 = os.environ.get("QDT", "f64") == "f32"   # QDT=f32: single precision; QWARM=1: time warm-started solves from the own solution (the fleet loop's regime)
s = BatchMPC(N=N, dtype=torch.float32 if F32 else torch.float64)
d = make_batch(B, N, cfg_id=2, seed=int(os.environ["QSEED"]) if "QSEED" in os.environ else None, dtype=np.float32 if F32 else np.float64)
dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
WARM = os.environ.get("QWARM") == "1"
if WARM:
    wu0 = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], want_U=True)["U"].clone()
    _solve = s.solve
    s.solve = lambda *a, **k: _solve(*a, warm_U=wu0.clone(), warm=True, **k)
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
