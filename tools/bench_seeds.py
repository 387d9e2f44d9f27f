"""Diagnostic: mean / worst B=4096 launch time over several synthetic batches (different seeds), per horizon."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])  # A/B of diagnostic builds
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
B = int(os.environ.get("QB", 4096))
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("X_"))
for N in [int(x) for x in os.environ.get("QNS", "20").split(",")]:
    s = BatchMPC(N=N)
    ts = []
    for seed in range(100, 100 + int(os.environ.get("QSEEDS", 8))):
        d = make_batch(B, N, cfg_id=2, seed=seed)
        dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
        o = None
        for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts = np.array(ts)
    print("%-16s N=%2d B=%d: mean %.4f ms (%.2f Msolves/s)  min %.3f max %.3f" % (tag, N, B, ts.mean(), B / ts.mean() / 1e3, ts.min(), ts.max()), flush=True)
