"""First-light diagnostics on the GPU box (writes gpurun_out/diag.txt); not a pytest file."""
import os, sys, time, traceback
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch, straight_line_case
from oracle import oracle as O

out = open("gpurun_out/diag.txt", "w")
def P(*a):
    s = " ".join(str(x) for x in a)
    print(s); out.write(s + "\n"); out.flush()

P(torch.cuda.get_device_name(0))
for dtype in (torch.float64, torch.float32):
    try:
        s = BatchMPC(N=8, dtype=dtype)
        lanes = np.arange(64)
        a = (lanes + 1).astype(np.float64); b = np.ones(64)
        # one-hot probes to recover the layout: A[i][k] = 1 only at lane la, B[k][j] only at lane lb
        rng = np.random.default_rng(0)
        A = rng.integers(-4, 5, (16, 4)).astype(np.float64); Bm = rng.integers(-4, 5, (4, 16)).astype(np.float64)
        d = s.debug_mfma_probe(A[lanes & 15, lanes >> 4], Bm[lanes >> 4, lanes & 15]).cpu().numpy()
        D = A @ Bm
        for name, rowf in (("f64map", lambda l, r: (l >> 4) + 4 * r), ("f32map", lambda l, r: 4 * (l >> 4) + r)):
            got = np.zeros((16, 16))
            for l in range(64):
                for r in range(4):
                    got[rowf(l, r), l & 15] = d[l, r]
            P(dtype, name, "match" if np.array_equal(got, D) else "MISMATCH")
    except Exception:
        P("probe failed", traceback.format_exc())

for N in (8, 20, 50):
    try:
        s = BatchMPC(N=N)
        B = 4
        d = make_batch(B, N, cfg_id=7, seed=1234 + N)
        rng = np.random.default_rng(5)
        U = np.stack([rng.uniform(-0.8, 0.8, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], axis=-1)
        p = O.params(N)
        for hz in (0, 1):
            H, g, J = [t.cpu().numpy() for t in s.debug_condense(d["z0"], d["ref"], d["v_target"], U, hessian=hz)]
            for b in range(B):
                q = O.problem(p, d["z0"][b], d["ref"][b], d["v_target"][b])
                Ho, go, Jo = O.condense(p, q, U[b], hessian=hz)
                P("condense N", N, "hess", hz, "b", b, "dJ", abs(J[b] - Jo), "dg", np.abs(g[b] - go).max() / max(1, np.abs(go).max()),
                  "dH", np.abs(H[b] - Ho).max() / np.abs(Ho).max())
    except Exception:
        P("condense failed", N, traceback.format_exc())

for N, B in ((8, 64), (20, 512), (50, 16)):
    try:
        s = BatchMPC(N=N)
        d = make_batch(B, N, cfg_id=2)
        t = time.time()
        o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True, want_X=True)
        torch.cuda.synchronize()
        t1 = time.time() - t
        r = {k: v.cpu().numpy() for k, v in o.items()}
        p = O.params(N)
        ro = O.solve_condensed_batch(p, d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=16)
        rel = np.abs(r["cost"] - ro["cost"]) / np.maximum(1, np.abs(ro["cost"]))
        P("solve N", N, "B", B, "time", t1, "status", np.bincount(r["status"], minlength=4), "iters gpu", r["iters"].mean(), "cpu", ro["iters"].mean(),
          "relcost max", rel.max(), "viol max", r["viol"].max(), "du0", np.abs(r["u0"] - ro["U"][:, 0]).max())
    except Exception:
        P("solve failed", N, traceback.format_exc())

# timing at the bench configuration
try:
    N, B = 20, 4096
    s = BatchMPC(N=N)
    d = make_batch(B, N, cfg_id=2)
    dev = {k: torch.as_tensor(v, device="cuda") for k, v in d.items() if k != "hard"}
    o = None
    for i in range(3):
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5):
        o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    P("bench N=20 B=4096 fp64: ms/step", ms, "solves/s", B / ms * 1e3, "iters mean", o["iters"].float().mean().item())
except Exception:
    P("bench failed", traceback.format_exc())
