"""Diagnostic: host-side cost of one BatchMPC.solve call at B = 1 (asynchronous issue time, synchronous latency).  Not part of the product or the tests."""
import time, torch, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
for N in (8, 20):
    s = BatchMPC(N=N)
    d = make_batch(1, N, cfg_id=2)
    dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"])
    torch.cuda.synchronize()
    # host cost of an async call
    t0 = time.perf_counter()
    for _ in range(2000): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("N=%d: host time per async call %.1f us; total per call incl. drain %.1f us" % (N, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
    lat = []
    for _ in range(500):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
        torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    print("   sync latency p50 %.1f us" % (np.percentile(lat, 50) * 1e6))
