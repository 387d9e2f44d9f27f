"""Diagnostic: find slow / non-optimal problems in a large synthetic batch on the GPU and dump them for the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 262144
d = make_batch(B, N, cfg_id=2)
s = BatchMPC(N=N)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
torch.cuda.synchronize()
it = o["iters"].cpu().numpy(); st = o["status"].cpu().numpy()
print("status counts", np.bincount(st, minlength=4), "iters mean %.2f p99 %d max %d" % (it.mean(), np.percentile(it, 99), it.max()))
print("iters histogram (>=30):", {int(k): int((it == k).sum()) for k in np.unique(it) if k >= 30})
idx = np.argsort(-it)[:24]
np.savez("gpurun_out/outliers.npz", idx=idx, z0=d["z0"][idx], ref=d["ref"][idx], v_target=d["v_target"][idx], u_prev=d["u_prev"][idx],
         iters=it[idx], status=st[idx], hard=d["hard"][idx], cost=o["cost"].cpu().numpy()[idx])
for k in idx[:12]:
    print(k, "iters", it[k], "status", st[k], "hard", d["hard"][k], "z0", np.round(d["z0"][k], 3), "vt %.2f" % d["v_target"][k], "up", np.round(d["u_prev"][k], 3))
