"""Diagnostic: robustness of the warm start when the warm point is wrong (solution of an unrelated problem)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
for N in (8, 20):
    B = 32768
    a = make_batch(B, N, cfg_id=5); b = make_batch(B, N, cfg_id=6)
    for wm, wp in [tuple(float(x) for x in c.split(",")) for c in os.environ.get("WARM_SCAN", "1e-3,1e-2;1e-6,1e-4;1e-7,1e-5").split(";")]:
        s = BatchMPC(N=N, warm_mu=wm, warm_push=wp)
        oa = s.solve(a["z0"], a["ref"], a["v_target"], a["u_prev"], want_U=True)
        cold = s.solve(b["z0"], b["ref"], b["v_target"], b["u_prev"])
        ci = cold["iters"].float().mean().item(); cc = cold["cost"].clone()
        wu = oa["U"].clone()
        w = s.solve(b["z0"], b["ref"], b["v_target"], b["u_prev"], warm_U=wu, warm=True)
        torch.cuda.synchronize()
        st = w["status"].cpu().numpy(); it = w["iters"].cpu().numpy()
        rel = (torch.abs(w["cost"] - cc) / torch.clamp(torch.abs(cc), min=1.0)).cpu().numpy()
        # warm from the problem's own solution (ideal warm start)
        ob = s.solve(b["z0"], b["ref"], b["v_target"], b["u_prev"], want_U=True)
        w2 = s.solve(b["z0"], b["ref"], b["v_target"], b["u_prev"], warm_U=ob["U"].clone(), warm=True)
        torch.cuda.synchronize()
        print("N=%d warm_mu %g push %g | wrong warm point: status %s iters mean %.2f max %d (cold %.2f), cost differs >1e-6: %d | own solution: iters mean %.2f status %s"
              % (N, wm, wp, np.bincount(st, minlength=4), it.mean(), it.max(), ci, (rel > 1e-6).sum(), w2["iters"].float().mean().item(), np.bincount(w2["status"].cpu().numpy(), minlength=4)))
