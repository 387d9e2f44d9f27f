"""Diagnostic: statuses and iteration statistics over several seeded draws per horizon (robustness of algorithm changes beyond the bench draws)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
for N, cfg, B, dt in ((20, 2, 131072, torch.float64), (50, 5, 16384, torch.float64), (8, 2, 131072, torch.float64), (20, 3, 131072, torch.float32), (28, 2, 65536, torch.float64), (40, 5, 16384, torch.float64)):
    s = BatchMPC(N=N, dtype=dt)
    tot = np.zeros(4, dtype=np.int64); its = []; mx = 0
    for k in range(1, 9):
        d = make_batch(B, N, cfg_id=cfg, seed=977 * k + N, dtype=np.float64 if dt == torch.float64 else np.float32)
        o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
        st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
        tot += np.bincount(st, minlength=4); its.append(it.mean()); mx = max(mx, int(it.max()))
    print("N=%d cfg %d %s: 8 x %d problems: status counts %s, mean iters %.3f, max %d" % (N, cfg, str(dt)[6:], B, tot, np.mean(its), mx), flush=True)
