"""Diagnostic: per-phase shader-cycle shares of the solver kernel (needs `make -C mkz_mpc_path_follower_amd/csrc stamps`).
Loads the KMPC_STAMPS build of the library in place of the shipped one.  Not part of the product or the tests."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_stamps.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
NAMES = ["setup", "linearize", "residual+mu", "condense", "build_K", "chol:backsub", "rhs", "predictor", "step+ftb", "eval+trial", "exit", "outputs", "chol:sweep1", "chol:schur", "chol:sweep2", "corr solve"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CFG = 5 if N == 50 else 2
for B in (1, 4096):
    s = BatchMPC(N=N)
    L = _lib.load()
    st = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
    L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
    d = make_batch(B, N, cfg_id=CFG)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
    torch.cuda.synchronize()
    it = o["iters"].double().cpu().numpy()
    c = st.cpu().numpy().astype(np.float64)
    tot = c.sum(1)
    print("B=%d N=%d mean iters %.2f, mean cycles/solve %.0f, cycles/iter %.0f" % (B, N, it.mean(), tot.mean(), (tot / it).mean()))
    for i, nm in enumerate(NAMES):
        print("   %-12s %9.0f cyc/iter  %5.1f %%" % (nm, (c[:, i] / it).mean(), 100 * c[:, i].sum() / tot.sum()))

# ---- tail analysis at the bench batch: which problems determine the launch time --------------------------------
B = 4096
s = BatchMPC(N=N)
st = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
d = make_batch(B, N, cfg_id=CFG)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
torch.cuda.synchronize()
it = o["iters"].cpu().numpy(); c = st.cpu().numpy().astype(np.float64); tot = c.sum(1)
order = np.argsort(-tot)[:8]
print("slowest problems of the B=4096 bench batch (wave lifetime in cycles; mean %.0f):" % tot.mean())
for b in order:
    nfac = c[b, 12] / max(c[order, 12].min() / 1.0, 1.0)
    print("  b=%4d iters %3d cycles %.0f (%.1fx mean)  hard=%s  condense share %.0f%%  sweep1 %.0f cyc/iter" % (b, it[b], tot[b], tot[b] / tot.mean(), d["hard"][b], 100 * c[b, 3] / tot[b], c[b, 12] / it[b]))
print("iters histogram:", np.bincount(it)[:40])
b = order[0]
print("phase cycles per iteration of the slowest problem (b=%d, %d iterations) next to the batch mean:" % (b, it[b]))
for i, nm in enumerate(NAMES):
    print("   %-12s %9.0f   %9.0f" % (nm, c[b, i] / it[b], (c[:, i] / it).mean()))
