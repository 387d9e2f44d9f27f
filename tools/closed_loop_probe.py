"""Diagnostic: run the batched closed loop and dump the first non-Optimal solves for the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
d = np.load("tests/golden/path1_decimated.npz")
N, B, vt = 8, 512, 8.0
grt = GPSRefTrajectory(arrays=dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"]), traj_horizon=N, traj_dt=0.2)
tr = grt.get_global_trajectory_reference()
rng = np.random.default_rng(11)
idx = rng.integers(0, int(0.6 * len(tr)), B); lat = rng.normal(0, 0.5, B); psi0 = tr[idx, 3]
sim = VehicleSimulator(B, X0=tr[idx, 4] - lat * np.sin(psi0), Y0=tr[idx, 5] + lat * np.cos(psi0), Psi0=psi0 + rng.normal(0, 0.05, B))
opts = {}
if len(sys.argv) > 2: opts = dict(warm_mu=float(sys.argv[1]), warm_push=float(sys.argv[2]))
loop = ClosedLoop(grt, sim, N=N, target_vel=vt, **opts)
all_it = []
dumps = []
for k in range(120):
    z0 = sim.state[:, 0:4].clone(); up = loop.u_prev.clone(); wu = loop.warm_U.clone()
    o = loop.step()
    st = o["status"].cpu().numpy(); it = o["iters"].cpu().numpy()
    bad = np.where(st != 0)[0]
    all_it.append(it.mean())
    if k % 10 == 0 or len(bad): print("step %3d status %s iters mean %.2f max %d  v mean %.2f" % (k, np.bincount(st, minlength=4), it.mean(), it.max(), sim.state[:, 3].mean().item()))
    for b in bad[:4]:
        dumps.append(dict(k=k, b=int(b), z0=z0[b].cpu().numpy(), ref=o["ref"][b].cpu().numpy(), up=up[b].cpu().numpy(), warm=wu[b].cpu().numpy(), status=int(st[b]), iters=int(it[b]), warm_flag=k > 0))
    if len(dumps) >= 8: break
print("opts", opts, "mean iters steps 1-60: %.2f, 60-119: %.2f, bad dumps %d" % (np.mean(all_it[1:60]), np.mean(all_it[60:]), len(dumps)))
np.save("gpurun_out/closed_loop_bad.npy", np.array(dumps, dtype=object), allow_pickle=True)
