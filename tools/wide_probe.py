"""N = 50 (WN=48 for the other compiled horizon): four-wave kernel (kernel_variant 0) vs the generic one-wave kernel (1) and the CPU oracle; timing at B = 4096"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("WN", "50"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = make_batch(B, N, cfg_id=5)
out = {}
for v in (0, 1):
    s = BatchMPC(N=N, kernel_variant=v)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True, want_X=True)
    torch.cuda.synchronize()
    out[v] = {k: x.cpu().numpy() for k, x in o.items()}
    print("variant", v, "status", np.bincount(out[v]["status"]), "iters mean %.2f max %d" % (out[v]["iters"].mean(), out[v]["iters"].max()), flush=True)
a, b = out[0], out[1]
rel = np.abs(a["cost"] - b["cost"]) / np.maximum(1.0, np.abs(b["cost"]))
print("rel cost max %.2e | dU max %.2e | dX max %.2e | viol %.2e | iters equal %d/%d" % (rel.max(), np.abs(a["U"] - b["U"]).max(), np.abs(a["X"] - b["X"]).max(), a["viol"].max(), (a["iters"] == b["iters"]).sum(), B))
if B >= 1024:
    for v in (0, 1):
        s = BatchMPC(N=N, kernel_variant=v)
        din = {k: torch.as_tensor(d[k], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
        o = None
        for _ in range(2): o = s.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=o)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): o = s.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=o)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 5 * 1e3
        print("variant %d: %.2f ms per launch, %.0f solves/s" % (v, ms, B / ms * 1e3))
