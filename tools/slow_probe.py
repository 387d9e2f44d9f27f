"""Diagnostic: per-phase cycles of the slowest problems of one seeded batch, stamps build (KMPC_STAMPS_LIB selects the library)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", os.environ.get("KMPC_STAMPS_LIB", "libkmpc_hip_stamps.so"))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
NAMES = ["setup", "linearize", "residual+mu", "condense", "build_K", "chol:backsub", "rhs", "predictor", "step+ftb", "eval+trial", "exit", "outputs", "chol:sweep1", "chol:schur", "chol:sweep2", "corr solve"]
N = 20; B = 4096; seed = int(sys.argv[1])
s = BatchMPC(N=N); L = _lib.load()
st = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]; L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
d = make_batch(B, N, cfg_id=2, seed=seed)
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"]); torch.cuda.synchronize()
it = o["iters"].cpu().numpy(); c = st.cpu().numpy().astype(np.float64); tot = c.sum(1)
for b in np.argsort(-tot)[:4]:
    print("b=%d iters %d cycles %.0f :" % (b, it[b], tot[b]), " ".join("%s=%.0f" % (NAMES[i], c[b, i]) for i in range(16) if c[b, i] > 0))
