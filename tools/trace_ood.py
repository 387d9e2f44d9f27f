"""per-iteration record of one problem of the out-of-distribution batch on the GPU (needs `make -C mkz_mpc_path_follower_amd/csrc trace`); usage: trace_ood.py N index [f32|f64] [B of the sweep, default 262144]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_trace.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_ood_batch
N, b = int(sys.argv[1]), int(sys.argv[2])
F32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
paths = [dict(np.load(os.path.join(ROOT, "tests", "golden", "path%d_decimated.npz" % k))) for k in (1, 2, 3)]
d = make_ood_batch(int(sys.argv[4]) if len(sys.argv) > 4 else 262144, N, seed=4100 + N, paths=paths)
s = BatchMPC(N=N, dtype=torch.float32 if F32 else torch.float64)
L = _lib.load()
tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda")
L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
L.kmpc_debug_set_stamps(C.c_void_p(tr.data_ptr()))
o = s.solve(d["z0"][b:b + 1], d["ref"][b:b + 1], d["v_target"][b:b + 1], d["u_prev"][b:b + 1])
torch.cuda.synchronize()
t = tr.cpu().numpy(); it = int(o["iters"][0])
print("problem", b, "status", int(o["status"][0]), "iters", it, "cost", float(o["cost"][0]), "viol", float(o["viol"][0]))
print(" it        err0         rd       comp         mu               J      alpha  ls flags(exact,indef*2,corr*4,tiny*8)")
for k in range(1, min(it, 255) + 1):
    r = t[k]
    print("%3d  %10.3e %10.3e %10.3e %10.3e %17.10f %10.3e %3d %3d" % (k, r[0], r[1], r[2], r[3], r[4], r[5], int(r[6]), int(r[7])))
print(" it         reg       hmax        dwl        dws  attempt   max-ds/s       dphi  max-dl/l")
for k in range(1, min(it, 127) + 1):
    r = t[128 + k]
    print("%3d  %10.3e %10.3e %10.3e %10.3e %3d %10.3e %10.3e %10.3e" % (k, r[0], r[1], r[2], r[3], int(r[4]), r[5], r[6], r[7]))
