"""Fuzz of the Frenet functor (MKZMPCPathFollowerFrenet.jl) far outside tests/test_frenet.py's road-like cases: e_y ~ U(-2.5, 2.5) m, e_psi ~ N(0, 0.3), v0 ~ U(0, 20),
curvature polynomials up to |K| ~ 0.15 1/m, previous command anywhere in the box; every compiled horizon, both precisions; GPU against the CPU port problem by
problem (cost within 1e-6 / 1e-3 relative or a lower / other minimum).   usage: python tools/fuzz_frenet.py [B]   (diagnostic; oracle/ is the checker)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle as O
    from mkz_mpc_path_follower_amd import BatchMPC
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    tot = err = 0
    for N in (8, 12, 16, 20, 24, 28):
        rng = np.random.default_rng(500 + N)
        z0 = np.stack([rng.uniform(0, 30, B), rng.uniform(-2.5, 2.5, B), rng.normal(0, 0.3, B), rng.uniform(0, 20, B)], 1)
        a, b, c, d = rng.uniform(-0.15, 0.15, B), rng.normal(0, 0.02, B), rng.normal(0, 0.02, B), rng.normal(0, 0.02, B)
        kp = np.stack([d / 60.0 ** 3, c / 60.0 ** 2, b / 60.0, a], 1)
        vt = rng.uniform(1.0, 15.0, B)
        up = np.stack([rng.uniform(-1.0, 1.0, B), rng.uniform(-0.5, 0.5, B)], 1)
        p = O.params(N, model=1)
        rc = [O.solve_condensed(p, O.problem_frenet(p, z0[i], kp[i], vt[i], up[i])) for i in range(B)]
        cst, sst, cit = np.array([r["cost"] for r in rc]), np.array([r["status"] for r in rc]), np.array([r["iters"] for r in rc])
        for tdt in (torch.float64, torch.float32):
            f32 = tdt == torch.float32
            o = BatchMPC(N=N, dtype=tdt, model=1).solve_frenet(z0, kp, vt, up); torch.cuda.synchronize()
            r = {q: v.cpu().numpy() for q, v in o.items()}
            both = (r["status"] == 0) & (sst == 0)
            rel = (r["cost"] - cst) / np.maximum(1.0, np.abs(cst))
            tolr = 1e-3 if f32 else 1e-6
            print("N=%2d %s: GPU status %s iters mean %.1f max %d | port status %s iters mean %.1f max %d | both Optimal %d: same cost %d, GPU lower %d, GPU higher %d (worst +%.1e) | max viol %.1e"
                  % (N, str(tdt)[6:], np.bincount(r["status"], minlength=4).tolist(), r["iters"].mean(), r["iters"].max(), np.bincount(sst, minlength=4).tolist(), cit.mean(), cit.max(),
                     both.sum(), (both & (np.abs(rel) <= tolr)).sum(), (both & (rel < -tolr)).sum(), (both & (rel > tolr)).sum(), max(rel[both].max(), 0.0), r["viol"].max()), flush=True)
            tot += B; err += int((r["status"] == 3).sum())
    print("problems", tot, "Error statuses", err)


if __name__ == "__main__":
    main()
