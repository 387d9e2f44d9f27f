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
NAMES = ["setup", "linearize", "residual+mu", "condense", "build_K", "cholesky", "rhs", "chol_solve", "step+ftb", "linesearch", "exit", "outputs"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for B in (1, 4096):
    s = BatchMPC(N=N)
    L = _lib.load()
    st = torch.zeros((B, 12), dtype=torch.int64, device="cuda")
    L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
    L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
    d = make_batch(B, N, cfg_id=2)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
    torch.cuda.synchronize()
    it = o["iters"].double().cpu().numpy()
    c = st.cpu().numpy().astype(np.float64)
    tot = c.sum(1)
    print("B=%d N=%d mean iters %.2f, mean cycles/solve %.0f, cycles/iter %.0f" % (B, N, it.mean(), tot.mean(), (tot / it).mean()))
    for i, nm in enumerate(NAMES):
        print("   %-12s %9.0f cyc/iter  %5.1f %%" % (nm, (c[:, i] / it).mean(), 100 * c[:, i].sum() / tot.sum()))
