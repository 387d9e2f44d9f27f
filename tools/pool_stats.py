"""CPU: pooled iteration statistics of the CPU port (oracle/kmpc_condensed.c = the algorithm the kernels run) over K seeded 4096-problem batches:
mean iterations, mean iteration-equivalents (iterations + 0.56 x re-factorisations) and the bootstrap expectation of the worst iteration-equivalent of a
4096-problem draw -- the score DESIGN.md section 4c ranks rule variants by (per-batch maxima of a few batches are too noisy).  Variants are selected through the
port's KMPC_X_* environment knobs:   KMPC_X_FOO=1 python tools/pool_stats.py [N] [K]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B = 4096
its, eq, bad = [], [], 0
for k in range(K):
    d = make_batch(B, N, cfg_id=1000 + k)
    r = O.solve_condensed_batch(O.params(N), d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
    its.append(r["iters"]); eq.append(r["iters"] + 0.56 * r["n_refactor"]); bad += int((r["status"] != 0).sum())
its, eq = np.concatenate(its), np.concatenate(eq)
rng = np.random.default_rng(0)
w = [eq[rng.integers(0, len(eq), B)].max() for _ in range(400)]
print("N=%d, %d x %d problems: mean iterations %.3f, mean iteration-equivalents %.3f, E[worst of %d] %.2f (+- %.2f), max %.1f, not Optimal %d"
      % (N, K, B, its.mean(), eq.mean(), B, np.mean(w), np.std(w) / np.sqrt(len(w)), eq.max(), bad))
