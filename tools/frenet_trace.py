"""per-iteration trace of one Frenet test problem on the GPU with a KMPC_TRACE build.  usage: frenet_trace.py <lib.so> <index> [N]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", sys.argv[1])
from mkz_mpc_path_follower_amd import BatchMPC
from test_frenet import _cases
b = int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 28
z0, kp, vt, up = _cases(300, N, seed=21)
s = BatchMPC(N=N, model=1)
L = _lib.load()
tr = torch.zeros((256, 8), dtype=torch.float64, device="cuda")
L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
L.kmpc_debug_set_stamps(C.c_void_p(tr.data_ptr()))
o = s.solve_frenet(z0[b:b + 1], kp[b:b + 1], vt[b:b + 1], up[b:b + 1])
torch.cuda.synchronize()
t = tr.cpu().numpy(); it = int(o["iters"][0])
print(sys.argv[1], "problem", b, "status", int(o["status"][0]), "iters", it, "cost", float(o["cost"][0]))
print(" it        err0         rd       comp         mu               J      alpha  ls flags(exact,indef*2,corr*4,tiny*8)")
for k in range(1, min(it, 255) + 1):
    r = t[k]
    print("%3d  %10.3e %10.3e %10.3e %10.3e %15.8f %10.3e %3d %3d" % (k, r[0], r[1], r[2], r[3], r[4], r[5], int(r[6]), int(r[7])))
