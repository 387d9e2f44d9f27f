"""Diagnostic driver for counter passes: a few launches of ONE BASELINE config (resident inputs), nothing else on the GPU.
  python3 tools/launch_config.py --horizon 20 --dtype f32 --batch 262144 --cfg 3 --steps 3
Put it directly after `rocprofv3 ... --` (no wrapper: the profiler's preloaded library has already initialised the GPU)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd import BatchMPC
from mkz_mpc_path_follower_amd.synthetic import make_batch
ap = argparse.ArgumentParser()
ap.add_argument("--horizon", type=int, default=20); ap.add_argument("--dtype", default="f64"); ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--schedule", type=int, default=1); ap.add_argument("--cfg", type=int, default=2); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--seed", type=int, default=None); ap.add_argument("--packed", action="store_true", help="kmpc_solve_batch_packed on 64-B-aligned records (ABI v8)")
a = ap.parse_args()
tdt = torch.float64 if a.dtype == "f64" else torch.float32
d = make_batch(a.batch, a.horizon, cfg_id=a.cfg, seed=a.seed)
s = BatchMPC(N=a.horizon, dtype=tdt, schedule=a.schedule)
dev = {k: torch.as_tensor(d[k], dtype=tdt, device="cuda") for k in ("z0", "ref", "v_target", "u_prev")}
rec = s.pack(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"]) if a.packed else None
o = None
ev = []
for _ in range(a.steps + 1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o = s.solve_packed(rec, out=o) if a.packed else s.solve(dev["z0"], dev["ref"], dev["v_target"], dev["u_prev"], out=o); e1.record(); ev.append((e0, e1))
torch.cuda.synchronize()
ms = [x.elapsed_time(y) for x, y in ev[1:]]
print("N=%d %s B=%d%s: %.4f ms/launch, iters mean %.3f max %d, optimal %d" % (a.horizon, a.dtype, a.batch, " packed records" if a.packed else "", sum(ms) / len(ms), o["iters"].float().mean().item(),
      int(o["iters"].max().item()), int((o["status"] == 0).sum().item())))
