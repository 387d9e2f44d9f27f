"""Diagnostic: status / iteration statistics of large synthetic batches over horizons and both precisions (fast kernels)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])  # A/B of diagnostic builds
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
B = int(os.environ.get("QB", 262144))
for N in [int(x) for x in os.environ.get("QNS", "8,12,16,20,24,28").split(",")]:
    for dt in ("f64", "f32"):
        d = make_batch(B, N, cfg_id=int(os.environ.get("QCFG", 2)), dtype=np.float64 if dt == "f64" else np.float32)
        s = BatchMPC(N=N, dtype=torch.float64 if dt == "f64" else torch.float32)
        dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o); e1.record(); torch.cuda.synchronize()
        it = o["iters"].cpu().numpy(); st = o["status"].cpu().numpy()
        print("N=%2d %s B=%d: status %s iters mean %.3f p99.9 %d max %d  cost sum %.6e  %.2f Msolves/s" % (N, dt, B, np.bincount(st, minlength=4), it.mean(), np.percentile(it, 99.9), it.max(), o["cost"].double().sum().item(), B / e0.elapsed_time(e1) / 1e3), flush=True)
