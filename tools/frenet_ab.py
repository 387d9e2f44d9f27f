"""A/B of two library builds on the Frenet cases of tests/test_frenet.py (per-problem iteration counts vs the CPU checker)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and not sys.argv[1].startswith("@"):
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "@" + lib])
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mkz_mpc_path_follower_amd import _lib
lib = sys.argv[1][1:]
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", lib)
from mkz_mpc_path_follower_amd import BatchMPC
from oracle import oracle as O
from test_frenet import _cases
for N, B in ((20, 600), (28, 300)):
    z0, kp, vt, up = _cases(B, N, seed=21)
    for kv in (0, 1):
        if kv == 1 and N > 24: continue
        s = BatchMPC(N=N, model=1, kernel_variant=kv)
        o = s.solve_frenet(z0, kp, vt, up, want_U=True)
        torch.cuda.synchronize()
        it = o["iters"].cpu().numpy(); st = o["status"].cpu().numpy()
        r = O.solve_condensed_batch(O.params(N, model=1), z0, kp, vt, up, nthreads=8)
        rel = np.abs(o["cost"].cpu().numpy() - r["cost"]) / np.maximum(1, np.abs(r["cost"]))
        print("%-24s N=%d variant %d: GPU iters mean %.3f (CPU %.3f), differ on %d of %d, max rel cost %.2e, bad %d | first 16 GPU %s CPU %s" % (
            lib, N, kv, it.mean(), r["iters"].mean(), (it != r["iters"]).sum(), B, rel.max(), (st != 0).sum(), it[:16], r["iters"][:16]))
