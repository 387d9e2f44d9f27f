"""Diagnostic: per-phase cycle shares of the four-per-wave N = 8 kernel (needs `make -C mkz_mpc_path_follower_amd/csrc stamps`); the stamps
are those of the wave's row 0."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_stamps.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
NAMES = ["setup", "linearize", "residual+mu", "condense", "assemble K", "(kkt rest)", "form weights", "predictor", "step+ftb", "eval+trial", "exit", "outputs", "cholesky", "-", "-", "direction solve"]
for B, kv in ((1024, 0), (65536, 0), (1, 2), (65536, 2)):
    s = BatchMPC(N=8, kernel_variant=kv)
    L = _lib.load()
    st = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
    L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
    L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
    d = make_batch(B, 8, cfg_id=2)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
    torch.cuda.synchronize()
    c = st.cpu().numpy().astype(np.float64)
    it = o["iters"].double().cpu().numpy()
    sel = c.sum(1) > 0
    c, it = c[sel], it[sel]
    tot = c.sum(1)
    print("B=%d variant %d: %d stamped problems, mean iters %.2f, cycles/solve %.0f, cycles/iter %.0f" % (B, kv, sel.sum(), it.mean(), tot.mean(), (tot / it).mean()))
    for i, nm in enumerate(NAMES):
        if c[:, i].sum() > 0:
            print("   %-16s %9.0f cyc/iter  %5.1f %%" % (nm, (c[:, i] / it).mean(), 100 * c[:, i].sum() / tot.sum()))
