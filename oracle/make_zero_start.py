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
Run from the repo root:  python oracle/make_zero_start.py [N ...]   (N = 8, 20: 8 worker processes, ~2 minutes; N = 50 costs
~1-2 CPU-minutes per problem in the dense numpy full-space solver -- run it niced in the background; results are checkpointed per
problem in /tmp/kmpc_zero_start_N50.pkl and `python oracle/make_zero_start.py --assemble 50` writes the fixture from what is done)
"""
import os
import sys

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):   # one BLAS thread per worker process (8 workers x 8 threads thrash)
    os.environ.setdefault(_v, "1")
import pickle
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ipopt_like as IL  # noqa: E402
from oracle import oracle as O  # noqa: E402
from mkz_mpc_path_follower_amd.synthetic import make_batch  # noqa: E402

NODE_WEIGHTS = (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0)  # mpc_cmd_pub.jl:49


def _one(a):
    i, N, z0, ref, vt, up = a
    r = IL.solve_problem(N, z0, ref, vt, up, weights=NODE_WEIGHTS)
    return i, r["status"], r["cost"], r["U"], r.get("iters", -1)


def _ckpt(N):
    return "/tmp/kmpc_zero_start_N%d.pkl" % N


def solve_all(N, B, cfg_id, workers=8):
    d = make_batch(B, N, cfg_id=cfg_id)
    done = pickle.load(open(_ckpt(N), "rb")) if os.path.exists(_ckpt(N)) else {}
    todo = [(i, N, d["z0"][i], d["ref"][i], d["v_target"][i], d["u_prev"][i]) for i in range(B) if i not in done]
    with Pool(workers) as pool:
        for k, r in enumerate(pool.imap_unordered(_one, todo, chunksize=1)):
            done[r[0]] = r[1:]
            if k % 4 == 3 or k == len(todo) - 1:
                pickle.dump(done, open(_ckpt(N) + ".tmp", "wb"))
                os.replace(_ckpt(N) + ".tmp", _ckpt(N))
                print("N=%d: %d / %d problems done" % (N, len(done), B), flush=True)
    return done


def build(N, B, cfg_id, done=None, workers=8):
    """writes the fixture from the problems that are done (all of them unless assembled from a checkpoint), in batch order"""
    p = O.params(N, NODE_WEIGHTS)
    d = make_batch(B, N, cfg_id=cfg_id)
    if done is None:
        done = solve_all(N, B, cfg_id, workers)
    idx = np.array(sorted(done))
    d = {k: v[idx] for k, v in d.items()}
    B = len(idx)
    rc = O.solve_condensed_batch(p, d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
    st = np.array([done[i][0] for i in idx], np.int32)
    Ji = np.array([done[i][1] for i in idx])
    Ui = np.array([done[i][2] for i in idx]).reshape(B, N, 2)
    rel = np.abs(Ji - rc["cost"]) / np.maximum(1.0, np.abs(Ji))
    same = (st == 0) & (rc["status"] == 0) & (rel <= 1e-6)
    print("N=%d: %d problems (%d hard); ipopt-like zero start Optimal %d; condensed Optimal %d; same minimum (1e-6 rel) %d; "
          "different minimum %d (condensed lower in %d)" % (N, B, int(d["hard"].sum()), int((st == 0).sum()), int((rc["status"] == 0).sum()),
                                                            int(same.sum()), int((~same).sum()), int(((~same) & (rc["cost"] < Ji)).sum())))
    path = os.path.join(ROOT, "tests", "golden", "kmpc_zero_start_N%d.npz" % N)
    np.savez_compressed(path, N=np.int32(N), weights=np.array(NODE_WEIGHTS), index_in_batch=idx, z0=d["z0"], ref=d["ref"], v_target=d["v_target"],
                        u_prev=d["u_prev"], hard=d["hard"], status_ipopt_like=st, J_ipopt_like=Ji, U_ipopt_like=Ui,
                        iters_ipopt_like=np.array([done[i][3] for i in idx], np.int32),
                        status_condensed=rc["status"], J_condensed=rc["cost"], U_condensed=rc["U"])
    print("wrote", path, os.path.getsize(path), "bytes")


CFG = {8: 201, 20: 202, 50: 205}

def refresh_condensed(N):
    """re-solve the stored problems with the current state of oracle/kmpc_condensed.c and rewrite only the *_condensed columns (the
    full-space zero-start columns are what costs CPU-minutes per problem and do not depend on that file)"""
    path = os.path.join(ROOT, "tests", "golden", "kmpc_zero_start_N%d.npz" % N)
    G = dict(np.load(path))
    p = O.params(N, G["weights"])
    rc = O.solve_condensed_batch(p, G["z0"], G["ref"], G["v_target"], G["u_prev"], nthreads=8)
    rel = np.abs(G["J_ipopt_like"] - rc["cost"]) / np.maximum(1.0, np.abs(G["J_ipopt_like"]))
    print("N=%d: condensed Optimal %d of %d; same minimum as the zero-start full-space solve (1e-6 rel) %d; condensed lower in %d of the others"
          % (N, int((rc["status"] == 0).sum()), len(rel), int((rel <= 1e-6).sum()), int(((rel > 1e-6) & (rc["cost"] < G["J_ipopt_like"])).sum())))
    G["status_condensed"], G["J_condensed"], G["U_condensed"] = rc["status"], rc["cost"], rc["U"]
    np.savez_compressed(path, **G)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--refresh-condensed":
        for a in sys.argv[2:]:
            refresh_condensed(int(a))
    elif len(sys.argv) > 2 and sys.argv[1] == "--assemble":
        N = int(sys.argv[2])
        build(N, 208, CFG[N], done=pickle.load(open(_ckpt(N), "rb")))
    else:
        for N in [int(a) for a in sys.argv[1:]] or [8, 20, 50]:
            build(N, 208, CFG[N], workers=int(os.environ.get("KMPC_WORKERS", "8")))
