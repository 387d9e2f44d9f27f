"""Diagnostic: the closed-loop step of a 4096-vehicle fleet (bench.py closed_loop_fleet) and the device time of its pieces.  Not part of the product."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
print(json.dumps(bench.closed_loop_fleet(0), indent=1))
from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
d = np.load(os.path.join(ROOT, "tests", "golden", "path1_decimated.npz"))
grt = GPSRefTrajectory(arrays=dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"]), traj_horizon=8, traj_dt=0.2, device=0)
tr = grt.get_global_trajectory_reference()
B = 4096
idx = np.linspace(0, int(0.6 * (len(tr) - 1)), B).astype(int)
sim = VehicleSimulator(B, X0=tr[idx, 4], Y0=tr[idx, 5], Psi0=tr[idx, 3], device=0)
loop = ClosedLoop(grt, sim, N=8, target_vel=8.0)
for _ in range(5): loop.step()
def ev(f, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pose = sim.state[:, 0:3].contiguous()
print("waypoints   %.3f ms" % ev(lambda: grt.get_waypoints_batch(pose, loop.v_target)))
print("plant       %.3f ms" % ev(lambda: sim._update_vehicle_model(10)))
ref, stop = grt.get_waypoints_batch(pose, loop.v_target)
z0 = sim.state[:, 0:4].contiguous()
print("solve       %.3f ms" % ev(lambda: loop.mpc.solve(z0, ref, loop.v_target, loop.u_prev, warm_U=loop.warm_U, warm=True, out=loop.out)))
print("whole step  %.3f ms (no host synchronisation inside)" % ev(lambda: loop.step()))
t0 = time.perf_counter()
for _ in range(50): loop.step()
torch.cuda.synchronize()
print("whole step  %.3f ms wall" % ((time.perf_counter() - t0) / 50 * 1e3))
