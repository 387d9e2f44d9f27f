"""Diagnostic: mean iterations of the Frenet functor in the compile-time kernel, the generic kernel and the CPU checker on the same draw."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mkz_mpc_path_follower_amd import BatchMPC
from oracle import oracle as O
from test_frenet import _cases
from mkz_mpc_path_follower_amd.synthetic import make_batch
for N in (8, 20):
    z0, kp, vt, up = _cases(400, N, seed=5)
    its = []
    for v in (0, 1):
        o = BatchMPC(N=N, dtype=torch.float64, model=1, kernel_variant=v).solve_frenet(z0, kp, vt, up); torch.cuda.synchronize()
        its.append(o["iters"].cpu().numpy())
    r = O.solve_condensed_batch(O.params(N, model=1), z0, kp, vt, up, nthreads=8)
    print("Frenet N=%d: fast %.3f generic %.3f cpu %.3f | fast==cpu on %d, generic==cpu on %d of 400" % (N, its[0].mean(), its[1].mean(), r["iters"].mean(), (its[0] == r["iters"]).sum(), (its[1] == r["iters"]).sum()))
for N in (8, 13, 20):
    d = make_batch(400, N, cfg_id=6)
    its = []
    for v in (0, 1):
        o = BatchMPC(N=N, kernel_variant=v).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
        its.append(o["iters"].cpu().numpy())
    r = O.solve_condensed_batch(O.params(N), d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
    print("Cartesian N=%d: variant0 %.3f generic %.3f cpu %.3f | v0==cpu on %d, generic==cpu on %d of 400" % (N, its[0].mean(), its[1].mean(), r["iters"].mean(), (its[0] == r["iters"]).sum(), (its[1] == r["iters"]).sum()))
