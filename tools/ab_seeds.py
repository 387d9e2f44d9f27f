"""A/B of library builds on the B = 4096 workload: bench batch + 8 other seeded batches (solves/s by HIP events).
usage: ab_seeds.py libA.so [libB.so ...]   (paths relative to mkz_mpc_path_follower_amd/; each build runs in its own process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].startswith("@")):
    for lib in sys.argv[1:]:
        subprocess.call([sys.executable, os.path.abspath(__file__), "@" + lib])
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from mkz_mpc_path_follower_amd import _lib
lib = sys.argv[1][1:]
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", lib)
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("AB_N", "20")); B = int(os.environ.get("AB_B", "4096"))
s = BatchMPC(N=N)
def run(d, steps=20):
    dev = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(steps): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, o
ms, o = run(make_batch(B, N, cfg_id=2))
r = []
for k in range(8):
    m, o2 = run(make_batch(B, N, cfg_id=2, seed=20180620 + 7919 * (k + 1)), 10)
    r.append(m)
print("%-28s bench batch %.4f ms (%.2f M/s) | 8 seeds: mean %.4f ms -> mean rate %.2f M/s, min %.2f max %.2f | %s" % (
    lib, ms, B / ms / 1e3, np.mean(r), np.mean(B / np.array(r)) / 1e3, B / max(r) / 1e3, B / min(r) / 1e3, " ".join("%.3f" % x for x in r)))
