"""What rounding alone leaves between the slack iterates and b -/+ a_f^T U at the end of a solve (needs `make -C mkz_mpc_path_follower_amd/csrc driftprobe`):
the diagnostic build writes the largest relative drift of the last iterate's slacks into the violation output.  Sets the margins of KMPC_DRIFT_TOL_F64 / _F32."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_driftprobe.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
for dt in (torch.float64, torch.float32):
    for N, B, kv in ((8, 262144, 0), (8, 512, 0), (12, 65536, 0), (20, 262144, 0), (28, 65536, 0), (36, 16384, 0), (50, 16384, 0), (13, 4096, 1), (33, 2048, 1)):
        worst, its, nopt = 0.0, 0, 0
        for seed in (2, 11):
            d = make_batch(B, N, cfg_id=seed)
            o = BatchMPC(N=N, dtype=dt, kernel_variant=kv).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
            v = o["viol"].cpu().numpy(); worst = max(worst, float(v.max())); its = max(its, int(o["iters"].max())); nopt += int((o["status"] != 0).sum())
        print("%s N=%2d B=%6d variant %d: max relative slack drift %.3e (max iterations %d, not Optimal %d)" % (str(dt)[6:], N, B, kv, worst, its, nopt), flush=True)
