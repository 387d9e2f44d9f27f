"""Child process of test_slack_guard_trips_on_a_corrupted_slack: loads the TEST build libkmpc_hip_corrupt.so (csrc/Makefile: one thread's slack iterate is pushed
1e-3 off b - a_f^T U after the second accepted step) in place of the shipped library and prints the status counts of every kernel family as JSON."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_corrupt.so")
from mkz_mpc_path_follower_amd import BatchMPC  # noqa: E402
from mkz_mpc_path_follower_amd.synthetic import make_batch  # noqa: E402

out = {}
# (label, N, B, dtype, kernel_variant): one-wave kernels, four-per-wave kernel (N = 8 from 1024 problems), four-wave kernel, generic kernel
for label, N, B, dt, kv in (("fast_f64_N20", 20, 256, torch.float64, 0), ("fast_f32_N20", 20, 256, torch.float32, 0), ("fast_f64_N8", 8, 256, torch.float64, 0),
                            ("quad_f64_N8", 8, 2048, torch.float64, 0), ("quad_f32_N8", 8, 2048, torch.float32, 0), ("dense_f64_N12", 12, 4096, torch.float64, 0),
                            ("fast_f64_N28", 28, 128, torch.float64, 0), ("wide_f64_N50", 50, 64, torch.float64, 0), ("wide_f32_N36", 36, 64, torch.float32, 0),
                            ("generic_f64_N13", 13, 64, torch.float64, 1), ("generic_f32_N20", 20, 64, torch.float32, 1)):
    d = make_batch(B, N, cfg_id=2)
    o = BatchMPC(N=N, dtype=dt, kernel_variant=kv).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
    torch.cuda.synchronize()
    st = o["status"].cpu().numpy()
    out[label] = [int((st == k).sum()) for k in range(4)]
print("CORRUPT_PROBE " + json.dumps(out))
