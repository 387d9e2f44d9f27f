"""is the slowest problem slower under load because of contention (more cycles) or because of the clock (same cycles, more time)?
KMPC_STAMPS build: wave lifetime in shader cycles of the bench batch's slowest problem, alone and inside the full batch, next to the launch time"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "mkz_mpc_path_follower_amd", "libkmpc_hip_stamps.so")
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
N, B = 20, 4096
d = make_batch(B, N, cfg_id=2)
s = BatchMPC(N=N)
L = _lib.load()
L.kmpc_debug_set_stamps.argtypes = [C.c_void_p]
st = torch.zeros((B, 16), dtype=torch.int64, device="cuda")
L.kmpc_debug_set_stamps(C.c_void_p(st.data_ptr()))
o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"])
torch.cuda.synchronize()
it = o["iters"].cpu().numpy()
b = int(np.argmax(it))
def run(idx):
    dev = {k: torch.as_tensor(d[k][idx], device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    oo = None
    for _ in range(3): oo = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=oo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); st.zero_(); e0.record()
    oo = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=oo)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), st[:len(idx)].cpu().numpy().sum(1)
ms1, c1 = run(np.array([b]))
order = np.concatenate([[b], np.delete(np.arange(B), b)])
ms2, c2 = run(order)
print("slowest problem %d (%d iterations)" % (b, it[b]))
print("alone     : launch %.4f ms, wave lifetime %d cycles -> %.3f GHz if the wave spans the launch" % (ms1, c1[0], c1[0] / ms1 / 1e6))
print("full batch: launch %.4f ms, wave lifetime %d cycles -> %.3f GHz" % (ms2, c2[0], c2[0] / ms2 / 1e6))
print("cycles ratio %.3f, time ratio %.3f" % (c2[0] / c1[0], ms2 / ms1))
