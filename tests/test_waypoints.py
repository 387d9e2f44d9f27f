"""Waypoint generation: oracle/waypoints.py vs the reference's own path data (CPU),
and the HIP kernel (kmpc_waypoints_batch, through the C ABI) vs the oracle (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def path():
    return np.load(os.path.join(GOLD, "path1_decimated.npz"))


@pytest.fixture(scope="module")
def traj(path):
    from oracle import waypoints as W
    return W.build_trajectory(path["t"], path["lat"], path["lon"], path["psi"], float(path["lat0"]), float(path["lon0"]))


def test_projection_reproduces_the_recorded_xy(path, traj):
    """the .mat files store x, y next to lat, lon: the restated equirectangular projection
    (ref_gps_traj.py:33-52, origin launch/path_follow.launch:19-20) must reproduce them"""
    assert np.abs(traj[:, 4] - path["x"]).max() < 1e-9 and np.abs(traj[:, 5] - path["y"]).max() < 1e-9
    assert traj[0, 6] == 0.0 and np.all(np.diff(traj[:, 6]) >= 0)
    # product host code (vectorised) == oracle (scalar loop of the reference)
    from mkz_mpc_path_follower_amd.ref_traj import path_arrays
    t, lat, lon, psi, X, Y, cd = path_arrays(path["t"], path["lat"], path["lon"], path["psi"], float(path["lat0"]), float(path["lon0"]))
    assert np.abs(X - traj[:, 4]).max() < 1e-9 and np.abs(Y - traj[:, 5]).max() < 1e-9 and np.abs(cd - traj[:, 6]).max() < 1e-9


def test_oracle_waypoint_semantics(traj):
    from oracle import waypoints as W
    i0 = 500
    x0, y0, p0 = traj[i0, 4] + 0.3, traj[i0, 5] - 0.2, traj[i0, 3]
    xi, yi, pi_, stop, ci = W.get_waypoints(traj, x0, y0, p0, v_target=5.0)
    assert abs(ci - i0) <= 2 and len(xi) == 9 and not stop
    s = np.interp(np.sqrt(0) + traj[ci, 6] + 5.0 * 0.2 * np.arange(1, 10), traj[:, 6], traj[:, 6])
    assert np.allclose(np.hypot(np.diff(xi), np.diff(yi)), 1.0, atol=0.05)        # spacing v*dt, starts one step ahead (Q8)
    assert np.hypot(xi[0] - traj[ci, 4], yi[0] - traj[ci, 5]) > 0.5
    xt, yt, pt, stop, ci2 = W.get_waypoints(traj, x0, y0, p0)                     # time mode starts at the closest point
    assert ci2 == ci and xt[0] == traj[ci, 4] and yt[0] == traj[ci, 5]
    xe, ye, pe, stop, _ = W.get_waypoints(traj, traj[-3, 4], traj[-3, 5], traj[-3, 3], v_target=5.0)
    assert stop and xe[-1] == traj[-1, 4]                                         # clamped at the path end -> stop_cmd
    pw = W.fix_heading_wraparound(np.array([3.1, -3.1, -3.0]), 3.0)               # jump across +-pi gets unwrapped
    assert np.allclose(pw, [3.1, -3.1 + 2 * np.pi, -3.0 + 2 * np.pi])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["vtarget", "time"])
def test_kernel_matches_oracle(path, traj, mode):
    import torch
    from oracle import waypoints as W
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    H = 8
    g = GPSRefTrajectory(arrays=dict(t=path["t"], lat=path["lat"], lon=path["lon"], psi=path["psi"]), traj_horizon=H)
    assert np.abs(g.trajectory - traj).max() < 1e-9
    tr = g.trajectory
    rng = np.random.default_rng(7)
    B = 1500
    idx = rng.integers(0, len(tr), B)
    idx[:40] = len(tr) - 1 - np.arange(40)          # near the end of the path -> clamping / stop flag
    idx[40:60] = np.arange(20)                      # near the start
    pose = np.stack([tr[idx, 4] + rng.normal(0, 1.5, B), tr[idx, 5] + rng.normal(0, 1.5, B),
                     tr[idx, 3] + rng.normal(0, 0.3, B) + rng.choice([0, 0, 0, 2 * np.pi, -2 * np.pi], B)], axis=1)
    pose[60:70, :2] += 400.0                        # far away from the path
    vt = rng.uniform(0.5, 12.0, B) if mode == "vtarget" else None
    ref, stop, closest = g.get_waypoints_batch(pose, vt, want_closest=True)
    torch.cuda.synchronize()
    ref, stop, closest = ref.cpu().numpy(), stop.cpu().numpy(), closest.cpu().numpy()
    exact = 0
    for b in range(B):
        xi, yi, pi_, st, ci = W.get_waypoints(tr, pose[b, 0], pose[b, 1], pose[b, 2], None if vt is None else vt[b], traj_horizon=H)
        assert closest[b] == ci, (b, closest[b], ci)                        # index work: bit exact
        assert stop[b] == int(st), b
        e = max(np.abs(ref[b, :, 0] - xi).max(), np.abs(ref[b, :, 1] - yi).max(), np.abs(ref[b, :, 2] - pi_).max())
        assert e <= 1e-12, (b, e)                                            # fp64 values: np.interp's arithmetic, last ulp
        exact += e == 0.0
    assert exact >= 0.99 * B
    assert stop[:40].sum() > 0 or mode == "time"
    if mode == "vtarget":   # B = 1 mirror of the reference call
        x, y, p, s = g.get_waypoints(pose[3, 0], pose[3, 1], pose[3, 2], vt[3])
        assert np.array_equal(x, ref[3, :, 0]) and np.array_equal(p, ref[3, :, 2]) and s == bool(stop[3])


@pytest.mark.gpu
def test_non_finite_pose_is_contained(path, traj):
    """A NaN / inf pose (GPS glitch, diverged plant) must not index outside the path: np.argmin of an all-NaN / all-inf distance
    array is 0 (ref_gps_traj.py:172-173), so the kernel answers with the closest index 0 and keeps the rest of the batch intact."""
    import torch
    from oracle import waypoints as W
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    g = GPSRefTrajectory(arrays=dict(t=path["t"], lat=path["lat"], lon=path["lon"], psi=path["psi"]), traj_horizon=8)
    tr = g.trajectory
    pose = np.stack([tr[[100, 200, 300, 400, 500], 4], tr[[100, 200, 300, 400, 500], 5], tr[[100, 200, 300, 400, 500], 3]], axis=1)
    pose[1, 0] = np.nan
    pose[2, 1] = np.inf
    pose[3, 0] = -np.inf
    ref, stop, closest = g.get_waypoints_batch(pose, np.full(5, 5.0), want_closest=True)
    torch.cuda.synchronize()
    closest = closest.cpu().numpy()
    assert list(closest) == [100, 0, 0, 0, 500]
    for b in (0, 4):
        xi, yi, pi_, st, ci = W.get_waypoints(tr, pose[b, 0], pose[b, 1], pose[b, 2], 5.0, traj_horizon=8)
        assert np.abs(ref[b, :, 0].cpu().numpy() - xi).max() <= 1e-12 and ci == closest[b]
    assert torch.isfinite(ref[1:4, :, :2]).all()   # waypoints from index 0 on: finite positions


@pytest.mark.gpu
def test_waypoints_feed_the_solver_closed_loop(path):
    """state -> waypoints kernel -> MPC kernel -> command, for a small fleet, all device-resident; then the
    reference's node loop on the real path until the stop latch (mpc_cmd_pub.jl:86-157)."""
    import torch
    from mkz_mpc_path_follower_amd import BatchMPC, KinematicMPC
    from mkz_mpc_path_follower_amd.messages import StateEst
    from mkz_mpc_path_follower_amd.node import MPCNode
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    g = GPSRefTrajectory(arrays=dict(t=path["t"], lat=path["lat"], lon=path["lon"], psi=path["psi"]), traj_horizon=8)
    tr = g.trajectory
    B = 256
    idx = np.linspace(50, len(tr) - 400, B).astype(int)
    z0 = np.stack([tr[idx, 4] + 0.3, tr[idx, 5] - 0.2, tr[idx, 3] + 0.03, np.full(B, 5.0)], axis=1)
    vt = np.full(B, 5.0)
    ref, stop = g.get_waypoints_batch(z0[:, :3], vt)
    s = BatchMPC(N=8)
    o = s.solve(z0, ref, vt, np.zeros((B, 2)))
    torch.cuda.synchronize()
    assert (o["status"] == 0).all() and torch.isfinite(o["u0"]).all()
    # single vehicle, reference loop, last stretch of the path
    node = MPCNode(g.get_waypoints, lambda t, m: None, N=8, target_vel=6.0, mpc=KinematicMPC(N=8))
    i0 = len(tr) - 260
    z = np.array([tr[i0, 4], tr[i0, 5] + 0.5, tr[i0, 3], 5.0])
    n_stop = 0
    for i in range(400):
        node.state_est_callback(StateEst(x=z[0], y=z[1], psi=z[2], v=z[3]))
        cmd = node.step()
        beta = np.arctan(1.742 / 2.85 * np.tan(cmd.steer_angle_cmd))
        z = z + 0.1 * np.array([z[3] * np.cos(z[2] + beta), z[3] * np.sin(z[2] + beta), z[3] / 1.742 * np.sin(beta), cmd.accel_cmd])
        z[3] = max(z[3], 0.0)
        n_stop += node.command_stop
        if node.command_stop and z[3] == 0.0:
            break
    assert node.command_stop and n_stop > 3 and cmd.accel_cmd == -1.0
    d_end = np.hypot(z[0] - tr[-1, 4], z[1] - tr[-1, 5])
    assert d_end < 25.0
