"""Fuzz of cost weights x horizons x precisions on in-distribution and out-of-distribution states: every weight log-uniform over four decades around the node's
defaults (zero weights included with probability 0.3 each), B problems per setting; the GPU result against the CPU port problem by problem (cost within 1e-6 / 1e-3
relative, or both certified as different local minima).   usage: python tools/fuzz_weights.py [settings] [B]      (diagnostic; uses oracle/ as the checker)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import certify as CT
    from oracle import oracle as O
    from mkz_mpc_path_follower_amd import BatchMPC
    from mkz_mpc_path_follower_amd.synthetic import make_batch, make_ood_batch
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    rng = np.random.default_rng(77)
    base = np.array([9.0, 9.0, 10.0, 1.0, 100.0, 1000.0, 1.0, 1.0])   # (C_v, C_acc, C_df default to 0: the fuzz draws them around 1)
    paths = [dict(np.load(os.path.join(ROOT, "tests", "golden", "path%d_decimated.npz" % k))) for k in (1, 2, 3)]
    tot = bad = 0
    for k in range(S):
        N = int(rng.choice([8, 12, 16, 20, 24, 28, 32, 40, 50]))
        w = base * 10.0 ** rng.uniform(-2, 2, 8)
        w[rng.uniform(size=8) < 0.3] = 0.0
        if w[:3].sum() == 0: w[0] = 9.0
        ood = k % 2 == 1
        d = make_ood_batch(B, N, seed=9000 + k, paths=paths) if ood else make_batch(B, N, cfg_id=2, seed=9000 + k)
        p = O.params(N, list(w))
        rc = O.solve_condensed_batch(p, d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
        for tdt in (torch.float64, torch.float32):
            f32 = tdt == torch.float32
            o = BatchMPC(N=N, dtype=tdt, weights=list(w)).solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True); torch.cuda.synchronize()
            r = {q: v.cpu().numpy() for q, v in o.items()}
            rel = np.abs(r["cost"] - rc["cost"]) / np.maximum(1.0, np.abs(rc["cost"]))
            both = (r["status"] == 0) & (rc["status"] == 0)
            off = np.where(both & (rel > (1e-3 if f32 else 1e-6)))[0]
            note = ""
            if len(off):
                c = CT.certify_batch(O, p, d, r["U"].astype(np.float64), idx=off, relax=1e-5 if f32 else 1e-8)
                wr = np.maximum(c["ref_scaled_stationarity"], c["ref_scaled_complementarity"])
                unexplained = int((wr > (1e-2 if f32 else 1e-6)).sum())
                note = " other-minimum %d (GPU lower on %d), uncertified %d (worst %.1e)" % (len(off), int((r["cost"][off] < rc["cost"][off]).sum()), unexplained, wr.max())
                bad += unexplained
            nb = int((r["status"] != 0).sum()); tot += B
            print("%2d N=%2d %s %s w=%s: GPU status %s iters mean %.1f max %d | port status %s | max viol %.1e%s" % (k, N, "ood" if ood else "std", str(tdt)[6:], np.array2string(w, precision=2, separator=","),
                  np.bincount(r["status"], minlength=4).tolist(), r["iters"].mean(), r["iters"].max(), np.bincount(rc["status"], minlength=4).tolist(), r["viol"].max(), note), flush=True)
            bad += int((r["status"] == 3).sum())
    print("problems", tot, "errors or uncertified", bad)


if __name__ == "__main__":
    main()
