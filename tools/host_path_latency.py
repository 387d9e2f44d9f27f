"""Latency of the drop-in B = 1 path as the reference's node uses it (mpc_cmd_pub.jl:115-141): KinematicMPC.update_init_cond / update_reference / solve_model /
update_current_input / get_solver_results on HOST arrays through kmpc_solve_batch_host, warm-started, N = 8 -- next to the device-resident BatchMPC.solve call."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import KinematicMPC
for N in (8, 20):
    k = KinematicMPC(N=N)
    k.update_cost(9, 9, 10, 0, 100, 1000, 0, 0)
    lat, its = [], []
    x = 0.0
    for i in range(600):
        v = min(15.0, 0.1 * i)
        xr = x + v * 0.2 * np.arange(N + 1) + 0.3
        t0 = time.perf_counter()
        k.update_init_cond(x, 0.05 * np.sin(0.05 * i), 0.01, v)
        k.update_reference(xr, np.zeros(N + 1), np.zeros(N + 1), v)
        a, d, st = k.solve_model()
        k.update_current_input(d, a)
        res = k.get_solver_results()
        lat.append(time.perf_counter() - t0); its.append(k.iters)
        x += v * 0.1
        assert st == "Optimal", st
    lat = np.array(lat[100:]) * 1e6
    print("KinematicMPC N=%d (host arrays, kmpc_solve_batch_host): one control step p50 %.1f us p99 %.1f us, mean iterations %.2f" % (N, np.percentile(lat, 50), np.percentile(lat, 99), np.mean(its[100:])))
