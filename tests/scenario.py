"""The reference's own verification scenario (test helper; uses oracle/ -- test infrastructure).

launch/sim_path_follow.launch is the only end-to-end check the reference has: the MPC node (mpc_cmd_pub.jl) against the
dynamic-bicycle simulator (vehicle_simulator.py) on paths/path3_6_20.mat (:13), `track_using_time = True` (:8), `target_vel = 1.0` (:9), the plant started
AT REST at X0 = 0, Y0 = 3, Psi0 = -1.5 (:23-25) -- 0.4 m off the path's first point (0.0, 2.58) and 0.57 rad off its heading (-2.07), tracking the recorded,
varying speed profile (1.0 ... 9.5 m/s) for 66 s until the waypoint helper raises its stop flag.  The reference pins no numbers for it (it is watched in a
live plot); here the scenario is run twice -- by the CPU oracle alone (numpy waypoints + numpy plant + the C port of the solver: `oracle_closed_loop`) and on
the GPU through the product's ClosedLoop -- and the two runs are compared with each other and against what "follows the path" means.
"""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LAUNCH = dict(path="path3_decimated.npz", track_with_time=True, target_vel=1.0, X0=0.0, Y0=3.0, Psi0=-1.5)   # launch/sim_path_follow.launch:8,9,13,23-25
# the launch file's two sets of initial conditions with the paths it names for them (:22-30): "Good for Path 1/3" and, commented out, "Good for Path 2, RFS Gate"
# (X0 = 142, Y0 = -82, Psi0 = 2.0: 8 m and 0.6 rad off path2's first point).  t_settle: seconds after which the cross-track error must stay below 0.5 m.
VARIANTS = {"path3": dict(LAUNCH, t_settle=10.0, t_peak=0.3), "path1": dict(LAUNCH, path="path1_decimated.npz", t_settle=10.0, t_peak=0.3),
            "path2": dict(LAUNCH, path="path2_decimated.npz", X0=142.0, Y0=-82.0, Psi0=2.0, t_settle=15.0, t_peak=None)}
WEIGHTS = (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0)                                                       # mpc_cmd_pub.jl:49


def path_arrays(name=LAUNCH["path"]):
    d = np.load(os.path.join(GOLD, name))
    return dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"]), float(d["lat0"]), float(d["lon0"])


def cross_track(traj_xy, x, y):
    """distance from (x, y) [arrays] to the polyline through traj_xy [M,2] (closest point on any segment), and the index of the closest segment"""
    P, Q = traj_xy[:-1], traj_xy[1:]
    d = Q - P
    L2 = np.maximum((d ** 2).sum(1), 1e-18)
    out, seg = np.empty(len(x)), np.empty(len(x), dtype=np.int64)
    for i, (xi, yi) in enumerate(zip(x, y)):
        w = np.stack([xi - P[:, 0], yi - P[:, 1]], 1)
        s = np.clip((w * d).sum(1) / L2, 0.0, 1.0)
        e2 = ((w - s[:, None] * d) ** 2).sum(1)
        seg[i] = int(e2.argmin())
        out[i] = np.sqrt(e2[seg[i]])
    return out, seg


def oracle_closed_loop(O, steps, path=LAUNCH["path"], track_with_time=True, target_vel=1.0, X0=0.0, Y0=3.0, Psi0=-1.5, N=8, v0=0.0):
    """mpc_cmd_pub.jl:86-157 + vehicle_simulator.py for ONE vehicle, everything on the CPU from the oracle's restatements.
    -> dict of per-step arrays: state [steps+1, 8], cmd [steps, 2], status, iters, stop (latched), ref [steps, N+1, 3]"""
    from oracle import waypoints as W, vehicle_sim as V
    arr, lat0, lon0 = path_arrays(path)
    traj = W.build_trajectory(arr["t"], arr["lat"], arr["lon"], arr["psi"], lat0, lon0)
    p = O.params(N, WEIGHTS)
    des_speed = target_vel if target_vel > 0.0 else 0.0           # mpc_cmd_pub.jl:58-62
    s = V.initial_state(1, X0, Y0, Psi0)
    s[0, 3] = v0
    u_prev = np.zeros(2)
    U_prev, have_warm, command_stop = None, False, False
    log = dict(state=[s[0].copy()], cmd=[], status=[], iters=[], stop=[], ref=[], cost=[])
    for _ in range(steps):
        x, y, psi, v = s[0, 0], s[0, 1], s[0, 2], s[0, 3]
        xr, yr, pr, stop, _ci = W.get_waypoints(traj, x, y, psi, None if track_with_time else des_speed, traj_horizon=N)   # :99-112
        command_stop = command_stop or stop
        ref = np.stack([xr, yr, pr], 1)
        if not command_stop:
            q = O.problem(p, [x, y, psi, v], ref, des_speed, u_prev)
            r = O.solve_condensed(p, q, o=O.opts(warm=1) if have_warm else O.opts(), U0=U_prev)
            cmd = r["U"][0].copy()                               # (acc, d_f): published whatever the status (Q7)
            u_prev = cmd.copy()                                  # update_current_input (:140)
            U_prev, have_warm = r["U"].copy(), True
            log["status"].append(r["status"]); log["iters"].append(r["iters"]); log["cost"].append(r["cost"])
        else:
            cmd = np.array([-1.0, 0.0])                          # :148-153
            log["status"].append(-1); log["iters"].append(0); log["cost"].append(0.0)
        log["cmd"].append(cmd); log["stop"].append(command_stop); log["ref"].append(ref)
        s = V.update_vehicle_model(s, cmd[None, :], n_updates=10)    # 0.1 s of plant per 10 Hz control period
        log["state"].append(s[0].copy())
    out = {k: np.array(v) for k, v in log.items()}
    out["traj"] = traj
    return out


def summarize(traj, state, cmd, stop, dt=0.1):
    """tracking summary of a run: cross-track error per step, the time after which it stays below 0.5 m, largest first-step rate of the commands"""
    ect, seg = cross_track(traj[:, 4:6], state[:, 0], state[:, 1])
    above = np.where(ect >= 0.5)[0]
    t_conv = 0.0 if len(above) == 0 else (above[-1] + 1) * dt
    live = ~np.asarray(stop, dtype=bool)
    k = np.where(live)[0]
    dacc = np.abs(np.diff(np.concatenate([[0.0], cmd[k, 0]])))
    ddf = np.abs(np.diff(np.concatenate([[0.0], cmd[k, 1]])))
    return dict(ect=ect, seg=seg, t_converged=t_conv, max_dacc=float(dacc.max()) if len(k) else 0.0, max_ddf=float(ddf.max()) if len(k) else 0.0,
                n_live=int(live.sum()), t_stop=float(np.argmax(~live) * dt) if (~live).any() else None)
