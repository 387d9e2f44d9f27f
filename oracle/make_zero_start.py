"""oracle/make_zero_start.py -- TEST INFRASTRUCTURE.  Generates tests/golden/kmpc_zero_start_N{8,20,50}.npz.

PARITY UNPINNED (the reference's Julia/JuMP/Ipopt stack cannot run here; these are not reference outputs).

Question the fixture answers: the reference starts every primal at 0 (`start=0.0`,
MKZMPCPathFollower.jl:65-72) and lets Ipopt find a local minimum of a NON-CONVEX program; the HIP kernels start from
a feed-forward guess.  How often do the two land in different local minima?  Each of >= 200 seeded synthetic
problems per horizon (the bench distribution incl. its 5 % hard stratum, mkz_mpc_path_follower_amd/synthetic.py) is
solved by
  (1) oracle/ipopt_like.py   full-space JuMP model, all primals 0 at the start, Ipopt's published algorithm + defaults
  (2) oracle/kmpc_condensed.c the algorithm of the kernels (feed-forward start)
and BOTH answers are stored (with their KKT certificates) whether or not they agree.  tests/test_certify.py compares
the GPU with (1) and counts the disagreements.
Run from the repo root:  python oracle/make_zero_start.py   (8 worker processes, ~6 minutes)
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ipopt_like as IL  # noqa: E402
from oracle import oracle as O  # noqa: E402
from mkz_mpc_path_follower_amd.synthetic import make_batch  # noqa: E402

NODE_WEIGHTS = (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0)  # mpc_cmd_pub.jl:49


def _one(a):
    N, z0, ref, vt, up = a
    r = IL.solve_problem(N, z0, ref, vt, up, weights=NODE_WEIGHTS)
    return r["status"], r["cost"], r["U"], r.get("iters", -1)


def build(N, B, cfg_id):
    p = O.params(N, NODE_WEIGHTS)
    d = make_batch(B, N, cfg_id=cfg_id)
    with Pool(8) as pool:
        res = pool.map(_one, [(N, d["z0"][i], d["ref"][i], d["v_target"][i], d["u_prev"][i]) for i in range(B)], chunksize=1)
    rc = O.solve_condensed_batch(p, d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
    st = np.array([r[0] for r in res], np.int32)
    Ji = np.array([r[1] for r in res])
    Ui = np.array([r[2] for r in res]).reshape(B, N, 2)
    rel = np.abs(Ji - rc["cost"]) / np.maximum(1.0, np.abs(Ji))
    same = (st == 0) & (rc["status"] == 0) & (rel <= 1e-6)
    print("N=%d: %d problems (%d hard); ipopt-like zero start Optimal %d; condensed Optimal %d; same minimum (1e-6 rel) %d; "
          "different minimum %d (condensed lower in %d)" % (N, B, int(d["hard"].sum()), int((st == 0).sum()), int((rc["status"] == 0).sum()),
                                                            int(same.sum()), int((~same).sum()), int(((~same) & (rc["cost"] < Ji)).sum())))
    path = os.path.join(ROOT, "tests", "golden", "kmpc_zero_start_N%d.npz" % N)
    np.savez_compressed(path, N=np.int32(N), weights=np.array(NODE_WEIGHTS), z0=d["z0"], ref=d["ref"], v_target=d["v_target"],
                        u_prev=d["u_prev"], hard=d["hard"], status_ipopt_like=st, J_ipopt_like=Ji, U_ipopt_like=Ui,
                        iters_ipopt_like=np.array([r[3] for r in res], np.int32),
                        status_condensed=rc["status"], J_condensed=rc["cost"], U_condensed=rc["U"])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or [8, 20, 50]
    for N in which:
        build(N, 208, cfg_id={8: 201, 20: 202, 50: 205}[N])
