"""usage: scenario_probe.py [steps] [horizon].  The reference's own scenario (launch/sim_path_follow.launch: path3, time mode, plant at rest at (0, 3, -1.5)) on the GPU through ClosedLoop, next to the
CPU oracle's run of the same loop (tests/scenario.py): statuses, iterations, tracking summary, largest difference between the two state histories."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenario as S
from oracle import oracle as O
from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 8   # horizon (the reference: 8)
L = S.LAUNCH
arr, lat0, lon0 = S.path_arrays()
grt = GPSRefTrajectory(arrays=arr, traj_horizon=NH, traj_dt=0.2, lat0=lat0, lon0=lon0)
B = 65
rng = np.random.default_rng(3)
X0 = np.concatenate([[L["X0"]], L["X0"] + rng.uniform(-2, 2, B - 1)]); Y0 = np.concatenate([[L["Y0"]], L["Y0"] + rng.uniform(-2, 2, B - 1)])
P0 = np.concatenate([[L["Psi0"]], L["Psi0"] + rng.uniform(-0.5, 0.5, B - 1)])
sim = VehicleSimulator(B, X0=X0, Y0=Y0, Psi0=P0)
loop = ClosedLoop(grt, sim, N=NH, target_vel=L["target_vel"], track_with_time=True)
st, cmd, status, iters, stop = [sim.state.cpu().numpy().copy()], [], [], [], []
for k in range(steps):
    o = loop.step()
    torch.cuda.synchronize()
    stop.append(loop.command_stop.cpu().numpy().copy()); cmd.append(o["cmd"].cpu().numpy().copy()); status.append(o["status"].cpu().numpy().copy())
    iters.append(o["iters"].cpu().numpy().copy()); st.append(sim.state.cpu().numpy().copy())
st, cmd, status, iters, stop = map(np.array, (st, cmd, status, iters, stop))
ro = S.oracle_closed_loop(O, steps, N=NH)
tr = grt.get_global_trajectory_reference()
for b in (0, 1, 2, 3):
    sm = S.summarize(tr, st[:, b], cmd[:, b], stop[:, b])
    nl = sm["n_live"]
    print("vehicle %d: live steps %d, statuses %s, iterations mean %.2f max %d, cross-track max after 5 s %.3f m (first %.3f), converged < 0.5 m for good after %.1f s, "
          "max first-step |dacc| %.9f |ddf| %.9f, stops at %s s, final v %.3f" % (b, nl, np.bincount(status[:nl, b]).tolist(), iters[:nl, b].mean(), iters[:nl, b].max(),
          sm["ect"][50:nl].max(), sm["ect"][0], S.summarize(tr, st[:nl, b], cmd[:nl, b], stop[:nl, b])["t_converged"], sm["max_dacc"], sm["max_ddf"], sm["t_stop"], st[-1, b, 3]))
nl = int((~ro["stop"]).sum())
print("oracle    : live steps %d, statuses %s, iterations mean %.2f" % (nl, np.bincount(ro["status"][:nl]).tolist(), ro["iters"][:nl].mean()))
n = min(nl, int((~stop[:, 0]).sum()))
dpos = np.hypot(st[:n + 1, 0, 0] - ro["state"][:n + 1, 0], st[:n + 1, 0, 1] - ro["state"][:n + 1, 1])
print("GPU vs oracle, launch vehicle: max |pos| diff %.3e m (at step %d), max |v| diff %.3e, max |cmd| diff %.3e, iteration counts equal on %d of %d steps, both stop at step %d / %d"
      % (dpos.max(), dpos.argmax(), np.abs(st[:n + 1, 0, 3] - ro["state"][:n + 1, 3]).max(), np.abs(cmd[:n, 0] - ro["cmd"][:n]).max(), int((iters[:n, 0] == ro["iters"][:n]).sum()), n,
         int(np.argmax(stop[:, 0])), int(np.argmax(ro["stop"]))))
allv = [S.summarize(tr, st[:, b], cmd[:, b], stop[:, b]) for b in range(B)]
print("all %d vehicles: non-Optimal live solves %d, worst cross-track after 10 s %.3f m, all stopped %s, all at rest %s" % (
    B, sum(int((status[:a["n_live"], b] != 0).sum()) for b, a in enumerate(allv)), max(a["ect"][100:a["n_live"]].max() for a in allv), bool(stop[-1].all()), bool((st[-1, :, 3] == 0).all())))
