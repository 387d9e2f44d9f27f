"""N = 8: four problems per wave (kmpc_quad.hip) against one problem per wave (kmpc_fast.hip, kernel_variant 2), solves/s by HIP events.
usage: python tools/quad_probe.py [fp32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
tdt = torch.float32 if "fp32" in sys.argv else torch.float64
def run(B, variant, warm=False):
    s = BatchMPC(N=8, dtype=tdt, kernel_variant=variant)
    d = make_batch(B, 8, cfg_id=2)
    dev = {k: torch.as_tensor(d[k], dtype=tdt, device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
    o = None
    for _ in range(3): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    reps = 20 if B <= 65536 else 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): o = s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, o["iters"].float().mean().item(), int(o["iters"].max().item()), int((o["status"] != 0).sum().item())
for B in (256, 512, 1024, 2048, 4096, 16384, 65536, 262144):
    # variant 0 picks the four-per-wave kernel from KMPC_QUAD_MIN_BATCH problems on; below that both lines run the one-wave kernel
    q = run(B, 0); w = run(B, 2)
    print("B=%6d  auto %.4f ms (%.2f M/s, iters %.2f max %d, bad %d) | one-wave %.4f ms (%.2f M/s, iters %.2f max %d, bad %d) | ratio %.2f" % (
        B, q[0], B / q[0] / 1e3, q[1], q[2], q[3], w[0], B / w[0] / 1e3, w[1], w[2], w[3], w[0] / q[0]), flush=True)
