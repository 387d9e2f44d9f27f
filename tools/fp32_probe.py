"""Diagnostic: fp32 kernel at BASELINE configs[2] (B=262144, N=20) -- throughput and accuracy vs the fp64 kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 262144
d = make_batch(B, N, cfg_id=3)
res = {}
for dt in (torch.float64, torch.float32):
    s = BatchMPC(N=N, dtype=dt)
    dev = {k: torch.as_tensor(d[k], dtype=dt, device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(2): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    r = {k: v.double().cpu().numpy() for k, v in o.items()}
    res[dt] = r
    print(dt, "%.2f ms  %.2f Msolves/s" % (ms, B / ms / 1e3), "status", np.bincount(r["status"].astype(int), minlength=4), "iters mean %.2f max %d" % (r["iters"].mean(), r["iters"].max()), "viol max %.3g" % r["viol"].max())
a, b = res[torch.float64], res[torch.float32]
ok = (b["status"] == 0) & (a["status"] == 0)
rel = np.abs(b["cost"] - a["cost"]) / np.maximum(1.0, np.abs(a["cost"]))
print("fp32 vs fp64: rel cost err  p50 %.2e p99 %.2e max %.2e (optimal in both: %d)" % (np.percentile(rel[ok], 50), np.percentile(rel[ok], 99), rel[ok].max(), ok.sum()))
print("fp32 vs fp64: |u0 diff| p99 %.2e max %.2e" % (np.percentile(np.abs(b["u0"] - a["u0"])[ok], 99), np.abs(b["u0"] - a["u0"])[ok].max()))
bad = np.where(b["status"] != 0)[0]
print("fp32 non-optimal:", len(bad), "rel cost err of those: max %.2e" % (rel[bad].max() if len(bad) else 0))
