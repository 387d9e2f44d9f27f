"""Mean launch time of B problems at horizon N over many seeded draws (one process; QSEEDS draws from seed QS0): the expectation the tail-bound B = 4096 headline
samples -- single draws differ by +-10 %.  usage: QN=20 QB=4096 QSEEDS=48 python tools/seed_mean.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("QN", 20)); B = int(os.environ.get("QB", 4096)); S = int(os.environ.get("QSEEDS", 48)); S0 = int(os.environ.get("QS0", 3000))
s = BatchMPC(N=N)
ms, mx, mean_it, bad = [], [], [], 0
for k in range(S):
    d = make_batch(B, N, cfg_id=2, seed=S0 + k)
    dev = {q: torch.as_tensor(d[q], device="cuda") for q in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    ms.append(best); mx.append(int(o["iters"].max().item())); mean_it.append(o["iters"].float().mean().item()); bad += int((o["status"] != 0).sum().item())
ms = np.array(ms)
print("N=%d B=%d over %d draws: mean launch %.4f ms (min %.4f max %.4f) = %.3f M solves/s (harmonic), mean of per-draw max iterations %.2f, mean iterations %.3f, not Optimal %d"
      % (N, B, S, ms.mean(), ms.min(), ms.max(), B / ms.mean() / 1e3, np.mean(mx), np.mean(mean_it), bad))
