"""worst-certified Optimal problems of the out-of-distribution batch on the GPU (sample as in tools/ood_sweep.py); saves them for offline analysis"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import certify as CT
from oracle import oracle as O
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_ood_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 262144
paths = [dict(np.load(os.path.join(ROOT, "tests", "golden", "path%d_decimated.npz" % k))) for k in (1, 2, 3)]
d = make_ood_batch(B, N, seed=4100 + N, paths=paths)
o = BatchMPC(N=N, dtype=torch.float64).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True); torch.cuda.synchronize()
r = {k: v.cpu().numpy() for k, v in o.items()}
p = O.params(N)
idx = CT.stratified_sample(r["iters"], r["status"], 2048)
c = CT.certify_batch(O, p, d, r["U"], idx=idx)
wr = np.maximum(c["ref_scaled_stationarity"], c["ref_scaled_complementarity"])
o_ = np.argsort(-wr)[:10]
for k in o_:
    b = idx[k]
    rc = O.solve_condensed_batch(p, d["z0"][b:b + 1], d["ref"][b:b + 1], d["v_target"][b:b + 1], d["u_prev"][b:b + 1], nthreads=1)
    print("#%d ref-scaled %.2e (stat %.2e comp %.2e thr %s) GPU %d it cost %.10g viol %.1e | port %d it cost %.10g st %d | fam %d z0 %s up %s" % (b, wr[k], c["stationarity"][k], c["complementarity"][k],
          c["threshold"][k], r["iters"][b], r["cost"][b], r["viol"][b], rc["iters"][0], rc["cost"][0], rc["status"][0], d["family"][b], d["z0"][b].tolist(), d["u_prev"][b].tolist()), flush=True)
np.savez(os.path.join(ROOT, "gpurun_out", "ood_worst_N%d.npz" % N), b=idx[o_], U=r["U"][idx[o_]], iters=r["iters"][idx[o_]], cost=r["cost"][idx[o_]])
