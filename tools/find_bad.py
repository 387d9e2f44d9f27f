"""Diagnostic: list the non-optimal problems of a large batch (fp32 or fp64), with the fp64 kernel's answer beside them."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import _lib
if os.environ.get("KMPC_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["KMPC_LIB"])  # A/B of diagnostic builds
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N = int(os.environ.get("QN", 20)); B = int(os.environ.get("QB", 262144)); f32 = os.environ.get("QDT", "f32") == "f32"
d = make_batch(B, N, cfg_id=2, dtype=np.float32 if f32 else np.float64)
s = BatchMPC(N=N, dtype=torch.float32 if f32 else torch.float64)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
bad = np.nonzero(st != 0)[0]
print(os.environ.get("KMPC_LIB", "shipped"), "N", N, "f32" if f32 else "f64", "non-optimal:", bad, "iters", it[bad], "cost", o["cost"].cpu().numpy()[bad], "viol", o["viol"].cpu().numpy()[bad])
for b in bad[:4]:
    print(" b", b, "z0", d["z0"][b], "vt", d["v_target"][b], "up", d["u_prev"][b], "hard", d["hard"][b])
if len(bad):
    d64 = make_batch(B, N, cfg_id=2)
    o64 = BatchMPC(N=N).solve(d64["z0"][bad], d64["ref"][bad], d64["v_target"][bad], d64["u_prev"][bad])
    print(" fp64 on the same: status", o64["status"].cpu().numpy(), "iters", o64["iters"].cpu().numpy(), "cost", o64["cost"].cpu().numpy())
