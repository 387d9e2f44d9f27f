"""Diagnostic (stamps build): final optimality error / mu / step length of one fp32 problem vs iteration cap."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_stamps.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 262144
d = make_batch(B, N, cfg_id=3)
L = _lib.load()
L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
st = torch.zeros((1, 16), dtype=torch.int64, device="cuda")
L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
for idx in (243186,):
    one = {k: d[k][idx:idx + 1] for k in ("z0", "ref", "v_target", "u_prev")}
    o32 = {k: torch.as_tensor(v, dtype=torch.float32, device="cuda") for k, v in one.items()}
    for k in list(range(1, 26)):
        r = BatchMPC(N=N, dtype=torch.float32, max_iter=k).solve(o32["z0"], o32["ref"], o32["v_target"], o32["u_prev"])
        torch.cuda.synchronize()
        v = st.cpu().numpy()[0, 13:16].view(np.float64)
        print("   cap %3d: status %d iters %3d cost %.7f err %.3e mu %.3e alpha %.3e" % (k, r["status"][0].item(), r["iters"][0].item(), r["cost"][0].item(), v[0], v[1], v[2]))
