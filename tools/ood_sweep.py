"""Out-of-distribution sweep: >= 262 144 problems per horizon from synthetic.make_ood_batch -- heading errors N(0, 0.3), lateral errors
U(-4, 4) m, initial speeds U(0, 20) m/s independent of the reference, curvature up to 0.2 1/m, time-mode windows of the three recorded paths with their varying
spacing and the coinciding waypoints at the paths' ends.  Reports statuses, the iteration tail, the launch time, certifies a stratified sample (tests/certify.py)
and lists every non-Optimal problem with what the CPU port says about it.   usage: python tools/ood_sweep.py [B] [N ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import certify as CT
from oracle import oracle as O
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_ood_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
NS = [int(a) for a in sys.argv[2:]] or [8, 20]
paths = [dict(np.load(os.path.join(ROOT, "tests", "golden", "path%d_decimated.npz" % k))) for k in (1, 2, 3)]
for N in NS:
    for dtype in (torch.float64, torch.float32):
        d = make_ood_batch(B, N, seed=4100 + N, paths=paths)
        s = BatchMPC(N=N, dtype=dtype)
        dev = {k: torch.as_tensor(d[k], dtype=dtype, device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}   # resident inputs, as everywhere else
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], want_U=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], want_U=True); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        r = {k: v.cpu().numpy() for k, v in o.items()}
        st, it = r["status"], r["iters"]
        print("N=%d %s B=%d: statuses %s (Optimal fraction %.6f), iterations mean %.2f p50 %d p99 %d p99.9 %d max %d, launch %.2f ms = %.2f M solves/s; family B share %.2f"
              % (N, str(dtype)[6:], B, np.bincount(st, minlength=4).tolist(), (st == 0).mean(), it.mean(), np.percentile(it, 50), np.percentile(it, 99), np.percentile(it, 99.9), it.max(),
                 ms, B / ms / 1e3, d["family"].mean()), flush=True)
        for fam in (0, 1):
            m = d["family"] == fam
            print("    family %s: Optimal fraction %.6f, iterations mean %.2f max %d, max violation %.2e" % ("AB"[fam], (st[m] == 0).mean(), it[m].mean(), it[m].max(), r["viol"][m].max()), flush=True)
        p = O.params(N)
        f32 = dtype == torch.float32
        idx = CT.stratified_sample(it, st, 2048 if N <= 8 else 1024)
        idx = idx[st[idx] == 0]
        c = CT.certify_batch(O, p, d, r["U"].astype(np.float64), idx=idx, relax=1e-5 if f32 else 1e-8)
        w = np.maximum(c["scaled_stationarity"], c["scaled_complementarity"]); wr = np.maximum(c["ref_scaled_stationarity"], c["ref_scaled_complementarity"])
        print("    certified %d Optimal problems (the %d with the most iterations + uniform): STRICT-scale max %.2e p99 %.2e, REFERENCE-scale max %.2e p99 %.2e, violation max %.2e, lam_min %.1e"
              % (len(idx), len(idx) // 8, w.max(), np.percentile(w, 99), wr.max(), np.percentile(wr, 99), c["violation"].max(), c["lam_min"].min()), flush=True)
        bad = np.where(st != 0)[0]
        for b in bad[:12]:
            rc = O.solve_condensed_batch(p, d["z0"][b:b + 1], d["ref"][b:b + 1], d["v_target"][b:b + 1], d["u_prev"][b:b + 1], nthreads=1)
            print("    not Optimal #%d: GPU status %d after %d iterations (cost %.6g, viol %.1e); CPU port: status %d, %d iterations, cost %.6g; family %s z0 %s u_prev %s"
                  % (b, st[b], it[b], r["cost"][b], r["viol"][b], rc["status"][0], rc["iters"][0], rc["cost"][0], "AB"[d["family"][b]], np.round(d["z0"][b], 3).tolist(), np.round(d["u_prev"][b], 3).tolist()), flush=True)
