"""Diagnostic: all compile-time-horizon kernels with NaN-poisoned LDS (needs `make -C mkz_mpc_path_follower_amd/csrc poison`).
Every problem must still be Optimal with the same cost as the shipped build: a kernel that reads an LDS word before anybody wrote it fails here
on every problem, while in normal runs it fails only when the LDS's previous occupant left garbage behind (first launch after another kernel)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_poison.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
from test_frenet import _cases
bad = 0
for dt in (torch.float64, torch.float32):
    for N in (8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48, 50):
        d = make_batch(512, N, cfg_id=2)
        o = BatchMPC(N=N, dtype=dt).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
        nb = int((o["status"] != 0).sum()); bad += nb
        print("Cartesian %s N=%2d: not Optimal %d, non-finite cost %d" % (str(dt)[6:], N, nb, int((~torch.isfinite(o["cost"])).sum())))
    for N in (8, 12, 16, 20, 24, 28):
        z0, kp, vt, up = _cases(512, N, seed=5)
        o = BatchMPC(N=N, dtype=dt, model=1).solve_frenet(z0, kp, vt, up); torch.cuda.synchronize()
        nb = int((o["status"] != 0).sum()); bad += nb
        print("Frenet    %s N=%2d: not Optimal %d" % (str(dt)[6:], N, nb))
for N, dt in ((13, torch.float64), (20, torch.float64), (50, torch.float64), (20, torch.float32), (33, torch.float32)):   # generic kernel (kernel_variant 1)
    d = make_batch(256, N, cfg_id=2)
    o = BatchMPC(N=N, dtype=dt, kernel_variant=1).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
    nb = int((o["status"] != 0).sum()); bad += nb
    print("generic Cartesian %s N=%2d: not Optimal %d" % (str(dt)[6:], N, nb))
for N in (8, 20, 24):
    z0, kp, vt, up = _cases(256, N, seed=5)
    o = BatchMPC(N=N, dtype=torch.float64, model=1, kernel_variant=1).solve_frenet(z0, kp, vt, up); torch.cuda.synchronize()
    nb = int((o["status"] != 0).sum()); bad += nb
    print("generic Frenet float64 N=%2d: not Optimal %d" % (N, nb))
print("TOTAL not Optimal:", bad)
