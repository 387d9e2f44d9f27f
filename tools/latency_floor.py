"""Where the B = 1 latency goes: host wall time around launch + synchronisation of (a) a near-empty kernel through the
same C ABI (kmpc_command_batch, B = 1), (b) the N = 8 solve cold and warm-started from its own solution, and the device time of the solves from HIP events over
back-to-back launches; the slope between the two solves is the time of one iteration of a lone wave."""
import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mkz_mpc_path_follower_amd import BatchMPC, _lib
from mkz_mpc_path_follower_amd.synthetic import make_batch, straight_line_case
L = _lib.load()
dev = torch.device("cuda", 0)
def wall(fn, n=400):
    lat = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
    return np.percentile(lat[50:], 50) * 1e6, np.percentile(lat[50:], 99) * 1e6
def events(fn, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
u0 = torch.zeros((1, 2), dtype=torch.float64, device=dev); stop = torch.zeros((1,), dtype=torch.int32, device=dev); latch = torch.zeros((1,), dtype=torch.bool, device=dev)
up = torch.zeros((1, 2), dtype=torch.float64, device=dev); cmd = torch.zeros((1, 2), dtype=torch.float64, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
empty = lambda: L.kmpc_command_batch(0, 1, p(u0), p(stop), p(latch), p(up), p(cmd), st)
print("near-empty kernel through the C ABI: wall p50 %.1f us p99 %.1f us; back-to-back %.1f us per launch" % (*wall(empty), events(empty)))
for N in (8, 20):
    s = BatchMPC(N=N)
    for name, d in (("module-load problem", straight_line_case(N, v0=0.0)), ("bench problem 0", {k: v[:1] for k, v in make_batch(4, N, cfg_id=2).items()})):
        din = {k: torch.as_tensor(d[k], device=dev) for k in ("z0", "ref", "v_target", "u_prev")}
        o = s.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], want_U=True); torch.cuda.synchronize()
        it_c = int(o["iters"][0]); wu = o["U"].clone()
        oc = {}
        cold = lambda: oc.__setitem__("o", s.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=oc.get("o")))
        ow = {}
        warm = lambda: ow.__setitem__("o", s.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], warm_U=wu.clone(), warm=True, out=ow.get("o")))
        wc, ww = wall(cold), wall(warm)
        ec, ew = events(cold), events(warm)
        it_w = int(ow["o"]["iters"][0])
        print("N=%d %s: cold %d iterations: wall p50 %.1f us (p99 %.1f), back-to-back %.1f us | warm %d iterations: wall p50 %.1f us, back-to-back %.1f us | per iteration %.2f us"
              % (N, name, it_c, wc[0], wc[1], ec, it_w, ww[0], ew, (ec - ew) / max(it_c - it_w, 1)))
