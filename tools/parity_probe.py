"""Diagnostic: GPU vs oracle on the bench batch -- which problems differ in cost, and are both KKT points?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
from oracle import oracle as O
N, B = 20, 4096
d = make_batch(B, N, cfg_id=2, seed=20180620 + 2)
s = BatchMPC(N=N)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True)
torch.cuda.synchronize()
g = {k: v.cpu().numpy() for k, v in o.items()}
p = O.params(N)
r = O.solve_condensed_batch(p, d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=16)
rel = np.abs(g["cost"] - r["cost"]) / np.maximum(1.0, np.abs(r["cost"]))
bad = np.where(rel > 1e-6)[0]
print("problems with cost rel diff > 1e-6:", len(bad), "max", rel.max(), "p99.9 %.2e" % np.percentile(rel, 99.9))
for b in bad[:6]:
    q = O.problem(p, d["z0"][b], d["ref"][b], d["v_target"][b], d["u_prev"][b])
    print(" b=%d gpu cost %.6f iters %d | oracle cost %.6f iters %d | |U diff| max %.3e" % (b, g["cost"][b], g["iters"][b], r["cost"][b], r["iters"][b], np.abs(g["U"][b] - r["U"][b]).max()))
    for nm, U in (("gpu", g["U"][b]), ("oracle", r["U"][b])):
        # projected-gradient style check: gradient and active set at the returned point
        gr = O.grad(p, q, U); A, bb = O.ineq(p, q); sl = bb - A @ U.ravel()
        act = sl < 1e-6
        # least-squares multipliers on the active set
        lam = np.zeros(len(bb))
        if act.any():
            lam_a, *_ = np.linalg.lstsq(A[act].T, -gr, rcond=None); lam[act] = lam_a
        res = gr + A.T @ lam
        print("    %-6s stationarity %.2e  min multiplier %.2e  active %d  min slack %.2e" % (nm, np.abs(res).max(), lam.min() if act.any() else 0.0, act.sum(), sl.min()))
