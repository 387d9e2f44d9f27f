"""Diagnostic: fp32 kernel robustness vs cold-start barrier parameter on two draws."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 262144
for cfg in (3, 4):
    d = make_batch(B, N, cfg_id=cfg)
    dev = {k: torch.as_tensor(d[k], dtype=torch.float32, device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    for mi in (0.1, 0.3, 1.0):
        s = BatchMPC(N=N, dtype=torch.float32, mu_init=mi)
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"])
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
        bad = np.where(st != 0)[0]
        print("draw %d mu_init %.1f: status %s iters mean %.2f p99 %d max %d bad idx %s" % (cfg, mi, np.bincount(st, minlength=4), it.mean(), np.percentile(it, 99), it.max(), bad[:4]))
