"""distribution of the independent KKT certificate (tests/certify.py) over a GPU batch: python tools/certify_probe.py N B dtype cfg [sample]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import certify as CT
from oracle import oracle as O
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B, dt, cfg = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
S = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
tdt = torch.float64 if dt == "f64" else torch.float32
d = make_batch(B, N, cfg_id=cfg)
r = {k: v.cpu().numpy() for k, v in BatchMPC(N=N, dtype=tdt).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True).items()}
idx = CT.stratified_sample(r["iters"], r["status"], S)
c = CT.certify_batch(O, O.params(N), d, r["U"].astype(np.float64), idx=idx, relax=1e-8 if dt == "f64" else 1e-5)
q = [50, 90, 99, 99.9, 100]
print("N=%d B=%d %s: status", np.bincount(r["status"]), "iters mean %.2f max %d" % (r["iters"].mean(), r["iters"].max()))
for k in ("scaled_stationarity", "scaled_complementarity", "ref_scaled_stationarity", "ref_scaled_complementarity", "violation"):
    print(k, " ".join("p%g=%.2e" % (p, v) for p, v in zip(q, np.percentile(c[k], q))))
if dt == "f64":
    w = np.argsort(-np.maximum(c["scaled_stationarity"], c["scaled_complementarity"]))[:8]
    for i in w:
        b = idx[i]
        print("problem", b, "stat %.2e comp %.2e thr %g iters %d cost %.6g hard %d" % (c["scaled_stationarity"][i], c["scaled_complementarity"][i], c["threshold"][i], r["iters"][b], r["cost"][b], d["hard"][b]))
if dt == "f32":
    r64 = {k: v.cpu().numpy() for k, v in BatchMPC(N=N, dtype=torch.float64).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]).items()}
    rel = np.abs(r["cost"] - r64["cost"]) / np.maximum(1.0, np.abs(r64["cost"]))
    print("rel cost vs fp64 GPU:", " ".join("p%g=%.2e" % (p, v) for p, v in zip(q + [99.99, 99.999], np.percentile(rel, q + [99.99, 99.999]))), "n>1e-3:", (rel > 1e-3).sum())
    w = np.argsort(-np.maximum(c["scaled_stationarity"], c["scaled_complementarity"]))[:8]
    for i in w:
        b = idx[i]
        print("problem", b, "stat %.2e comp %.2e iters %d cost32 %.6g cost64 %.6g rel %.1e" % (c["scaled_stationarity"][i], c["scaled_complementarity"][i], r["iters"][b], r["cost"][b], r64["cost"][b], rel[b]))
