"""Frenet functor (kmpc_config.model = 1): compile-time-horizon kernel (kernel_variant 0) vs generic kernel (1): launch time, solves/s, agreement"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mkz_mpc_path_follower_amd import BatchMPC
from test_frenet import _cases
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for N in (8, 20):
    z0, kp, vt, up = _cases(B, N, seed=5)
    dev = [torch.as_tensor(x, device="cuda") for x in (z0, kp, vt, up)]
    res = {}
    for v in (0, 1):
        s = BatchMPC(N=N, dtype=torch.float64, model=1, kernel_variant=v)
        o = None
        for _ in range(3): o = s.solve_frenet(*dev, out=o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): o = s.solve_frenet(*dev, out=o)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[v] = (ms, o["cost"].cpu().numpy(), o["iters"].float().mean().item(), int((o["status"] != 0).sum()))
        print("N=%2d B=%d variant %d: %.3f ms per launch, %.2f M solves/s, mean iterations %.2f, not Optimal %d" % (N, B, v, ms, B / ms / 1e3, res[v][2], res[v][3]))
    rel = np.abs(res[0][1] - res[1][1]) / np.maximum(1.0, np.abs(res[1][1]))
    print("   max rel cost difference %.1e, speed-up %.2fx" % (rel.max(), res[1][0] / res[0][0]))
