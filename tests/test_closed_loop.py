"""batched closed-loop simulator (scripts/vehicle_simulator.py restated) and the
device-resident loop state -> waypoints -> MPC -> plant of launch/sim_path_follow.launch.

Parity of the plant kernel vs the numpy oracle: the arithmetic is the reference's, operation for operation, in fp64;
sin/cos/atan2 differ from numpy's by <= 2 ulp per call, accumulated over 100-1000 Euler sub-steps => tolerance
1e-9 absolute on positions (metres) and 1e-10 on the other states, written below.
"""
import os

import numpy as np
import torch
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _path_arrays():
    d = np.load(os.path.join(HERE, "golden", "path1_decimated.npz"))
    return dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"])


# ---------------------------------------------------------------- CPU: the oracle itself
def test_sim_oracle_straight_line_and_lag():
    """no steering: vy = wz = 0, psi constant, acc follows the lag exactly (Euler of d/dt acc = 5 (acc_des - acc)),
    vx integrates it, X/Y move along psi"""
    from oracle import vehicle_sim as V
    s = V.initial_state(1, X=0.0, Y=0.0, psi=0.3)
    s[:, 3] = 5.0
    out = V.update_vehicle_model(s, [[1.0, 0.0]], n_updates=100)  # 1 s
    acc = vx = None
    a, v, x = 0.0, 5.0, 0.0
    for _ in range(1000):
        v_new = max(0.0, v + 1e-3 * a)
        x += 1e-3 * v
        v = v_new
        a = 5.0 * (1.0 - a) * 1e-3 + a
    assert abs(out[0, 6] - a) < 1e-14 and abs(out[0, 3] - v) < 1e-12
    assert abs(out[0, 4]) == 0.0 and abs(out[0, 5]) == 0.0 and abs(out[0, 2] - 0.3) < 1e-12
    assert abs(out[0, 0] - x * np.cos(0.3)) < 1e-9 and abs(out[0, 1] - x * np.sin(0.3)) < 1e-9


def test_sim_oracle_standstill_and_wrap():
    """at rest the lateral states stay 0 and vx is floored at 0 (vehicle_simulator.py:84-92); heading stays in [-pi, pi) (:101)"""
    from oracle import vehicle_sim as V
    s = V.initial_state(2, X=0.0, Y=0.0, psi=3.1)
    out = V.update_vehicle_model(s, [[-1.0, 0.3], [-1.0, -0.3]], n_updates=20)
    assert (out[:, 3] == 0.0).all() and (out[:, 4] == 0.0).all() and (out[:, 5] == 0.0).all()
    s[:, 3] = 8.0
    out = V.update_vehicle_model(s, [[0.5, 0.2], [0.5, 0.2]], n_updates=200)
    assert (out[:, 2] >= -np.pi).all() and (out[:, 2] < np.pi).all() and out[0, 2] < 0.0  # turned left through +pi


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_sim_kernel_matches_oracle():
    import torch
    from oracle import vehicle_sim as V
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    rng = np.random.default_rng(5)
    B = 3000
    s0 = np.zeros((B, 8))
    s0[:, 0:2] = rng.uniform(-500, 500, (B, 2))
    s0[:, 2] = rng.uniform(-np.pi, np.pi, B)
    s0[:, 3] = np.where(rng.random(B) < 0.1, 0.0, rng.uniform(0, 20, B))
    s0[:, 4] = rng.normal(0, 0.2, B) * (s0[:, 3] > 0)
    s0[:, 5] = rng.normal(0, 0.1, B) * (s0[:, 3] > 0)
    s0[:, 6] = rng.uniform(-1, 1, B)
    s0[:, 7] = rng.uniform(-0.5, 0.5, B)
    cmd = np.stack([rng.uniform(-1, 1, B), rng.uniform(-0.5, 0.5, B)], 1)
    sim = VehicleSimulator(B)
    sim.state.copy_(torch.as_tensor(s0))
    sim._mpc_cmd_callback(cmd[:, 0], cmd[:, 1])
    sim._update_vehicle_model(10)                       # one 10 Hz control period = 100 sub-steps
    got = sim.state.cpu().numpy()
    exp = V.update_vehicle_model(s0, cmd, n_updates=10)
    dpsi = np.abs((got[:, 2] - exp[:, 2] + np.pi) % (2 * np.pi) - np.pi)   # a wrap at +-pi may land on either side
    mv = s0[:, 3] > 0                                  # moving vehicles: rounding-level parity (libm sin/cos/atan2 differ by <= 1-2 ulp)
    assert np.abs(got[mv, 0:2] - exp[mv, 0:2]).max() < 1e-9
    assert dpsi[mv].max() < 1e-10 and np.abs(got[mv, 3:] - exp[mv, 3:]).max() < 1e-10
    # standing starts: with the reference's (Python 2) integer division at vehicle_simulator.py:84 there is no lateral-force drag, so a
    # car at rest with a positive acceleration pulls away through 0 < vx < 0.03 m/s, where the 1 ms explicit Euler step of the
    # linear-tyre model is unstable (dt * (C_f + C_r) / (m vx) > 2): last-ulp differences of atan2 are amplified ~1e9-fold there.
    # Same model, same pose to 1e-3 m / 1e-3 rad and the same speed to 1e-2 m/s after the control period -- not rounding-level; the
    # lateral velocity and yaw rate of that stratum are what the instability amplifies (they differ by up to ~5e-2 between libm
    # implementations) and are only required to stay bounded.
    assert np.abs(got[~mv, 0:2] - exp[~mv, 0:2]).max() < 1e-3 and dpsi[~mv].max() < 1e-3
    assert np.abs(got[~mv, 3] - exp[~mv, 3]).max() < 1e-2 and np.abs(got[~mv, 6:] - exp[~mv, 6:]).max() < 1e-10
    assert np.isfinite(got).all() and np.abs(got[~mv, 4:6]).max() < 2.0
    assert (~mv).sum() > 100 and (exp[~mv, 3] > 0).sum() > 50   # the stratum is there and some of it does pull away
    assert (got[:, 3] >= 0).all()
    # a caller-written state may carry vx < 0 (the kernel itself never produces one): the slip angles of that first sub-step go through
    # atan2's own quadrant logic as np.arctan2 does in the reference (:76-77), vx is clamped to 0 (:84) and a braking car then stays at rest
    B2 = 256
    s1 = np.zeros((B2, 8))
    s1[:, 0:2] = rng.uniform(-50, 50, (B2, 2)); s1[:, 2] = rng.uniform(-np.pi, np.pi, B2)
    s1[:, 3] = -rng.uniform(0.01, 5.0, B2); s1[:, 4] = rng.normal(0, 0.2, B2); s1[:, 5] = rng.normal(0, 0.1, B2)
    s1[:, 6] = -rng.uniform(0.2, 1.0, B2); s1[:, 7] = rng.uniform(-0.5, 0.5, B2)
    cmd1 = np.stack([-rng.uniform(0.2, 1.0, B2), rng.uniform(-0.5, 0.5, B2)], 1)
    sim1 = VehicleSimulator(B2)
    sim1.state.copy_(torch.as_tensor(s1))
    sim1._mpc_cmd_callback(cmd1[:, 0], cmd1[:, 1])
    sim1._update_vehicle_model(10)
    got1 = sim1.state.cpu().numpy()
    exp1 = V.update_vehicle_model(s1, cmd1, n_updates=10)
    assert np.abs(got1[:, 0:2] - exp1[:, 0:2]).max() < 1e-9 and np.abs(got1[:, 3:] - exp1[:, 3:]).max() < 1e-10
    assert np.abs((got1[:, 2] - exp1[:, 2] + np.pi) % (2 * np.pi) - np.pi).max() < 1e-10 and (got1[:, 3] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("N", [8, 20, 50])
def test_closed_loop_tracks_the_recorded_path(N):
    """B vehicles start on / next to the recorded path (fixture of paths/path1_6_20.mat) at rest and follow it for 12 s
    at target speed 8 m/s with the reference node's protocol: tracking errors stay bounded, speed converges,
    every solve Optimal, warm-started solves need fewer iterations than the first (cold) one.  N = 8 is the reference's horizon (four problems per
    wave below 1024 vehicles: the one-wave kernel); N = 20 and 50 put the BASELINE horizons -- the one-wave and the four-wave kernel -- through the same
    warm-started loop (measured: median cross-track 0.07 m at all three, 4.3 / 4.4 / 6.8 mean iterations after the cold first solve)."""
    import torch
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
    B, vt = 512, 8.0
    grt = GPSRefTrajectory(arrays=_path_arrays(), traj_horizon=N, traj_dt=0.2)
    tr = grt.get_global_trajectory_reference()
    rng = np.random.default_rng(11)
    idx = rng.integers(0, int((0.6 if N == 8 else 0.5) * len(tr)), B)        # leave room ahead: no stop flag within 12 s (+ the horizon's look-ahead)
    lat = rng.normal(0, 0.5, B)
    psi0 = tr[idx, 3]
    sim = VehicleSimulator(B, X0=tr[idx, 4] - lat * np.sin(psi0), Y0=tr[idx, 5] + lat * np.cos(psi0), Psi0=psi0 + rng.normal(0, 0.05, B))
    loop = ClosedLoop(grt, sim, N=N, target_vel=vt)
    iters, worst = [], 0
    for k in range(120):
        o = loop.step()
        iters.append(o["iters"].float().mean().item())
        worst = max(worst, int(o["status"].max().item()))
    assert worst == 0
    assert not loop.command_stop.any().item()
    st = sim.state.cpu().numpy()
    # cross-track error to the closest recorded point, heading error, speed
    d2 = (st[:, None, 0] - tr[None, :, 4]) ** 2 + (st[:, None, 1] - tr[None, :, 5]) ** 2
    j = d2.argmin(1)
    ect = np.sqrt(d2[np.arange(B), j])
    epsi = np.abs((st[:, 2] - tr[j, 3] + np.pi) % (2 * np.pi) - np.pi)
    # (distance to the closest recorded SAMPLE: includes up to half the sample spacing of the decimated fixture, ~0.5 m at speed)
    assert np.median(ect) < 0.25 and ect.max() < 1.0 and epsi.max() < 0.15, (np.median(ect), ect.max(), epsi.max())
    # C_v = 0 in the node's weights (mpc_cmd_pub.jl:49): speed is set by the spacing of the waypoints (v_target * dt), not penalised
    assert np.abs(st[:, 3] - vt).max() < 1.5
    assert np.mean(iters[20:]) < iters[0]


@pytest.mark.gpu
def test_closed_loop_stop_latch():
    """vehicles that start near the end of the path get the stop flag from the waypoint helper; the command latches to
    accel -1.0 / steer 0.0 (mpc_cmd_pub.jl:148-153) and the cars come to rest (vx floored at 0)"""
    import torch
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
    N, B, vt = 8, 8, 6.0
    grt = GPSRefTrajectory(arrays=_path_arrays(), traj_horizon=N, traj_dt=0.2)
    tr = grt.get_global_trajectory_reference()
    i0 = len(tr) - 40
    sim = VehicleSimulator(B, X0=tr[i0, 4], Y0=tr[i0, 5], Psi0=tr[i0, 3])
    sim.state[:, 3] = vt
    loop = ClosedLoop(grt, sim, N=N, target_vel=vt)
    last = None
    for k in range(150):
        last = loop.step()
    assert loop.command_stop.all().item()
    assert torch.equal(last["cmd"], torch.tensor([[-1.0, 0.0]] * B, dtype=torch.float64, device=last["cmd"].device))
    assert (sim.state[:, 3] == 0.0).all().item()


@pytest.mark.gpu
def test_command_stage_matches_the_node_loop():
    """kmpc_command_batch = mpc_cmd_pub.jl:100-103 (stop latch), :148-153 (a latched vehicle gets accel -1 / steer 0), :140 (the published input is
    the next solve's rate-limit anchor, only on the solve branch), for a batch: against the same rules written with torch.where"""
    import ctypes as C
    from mkz_mpc_path_follower_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(3)
    B = 1000
    u0 = torch.randn((B, 2), dtype=torch.float64, generator=g).cuda()
    stop = (torch.rand((B,), generator=g) < 0.2).to(torch.int32).cuda()
    latch = (torch.rand((B,), generator=g) < 0.3).cuda()
    u_prev = torch.randn((B, 2), dtype=torch.float64, generator=g).cuda()
    cmd = torch.full((B, 2), 7.0, dtype=torch.float64, device="cuda")
    exp_latch = latch | stop.bool()
    exp_cmd = torch.where(exp_latch.unsqueeze(1), torch.tensor([-1.0, 0.0], dtype=torch.float64, device="cuda"), u0)
    exp_prev = torch.where(exp_latch.unsqueeze(1), u_prev, u0)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.kmpc_command_batch(0, B, p(u0), p(stop), p(latch), p(u_prev), p(cmd), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(latch, exp_latch) and torch.equal(cmd, exp_cmd) and torch.equal(u_prev, exp_prev)
    assert L.kmpc_command_batch(0, -1, p(u0), p(stop), p(latch), p(u_prev), p(cmd), None) < 0


@pytest.mark.gpu
def test_closed_loop_rejects_a_solver_of_the_wrong_type():
    """kmpc_command_batch reads the solver's first inputs as [B,2] doubles -- a caller-supplied fp32 / other-horizon BatchMPC is refused"""
    import torch
    from mkz_mpc_path_follower_amd import BatchMPC
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
    grt = GPSRefTrajectory(arrays=_path_arrays(), traj_horizon=8, traj_dt=0.2)
    sim = VehicleSimulator(4)
    with pytest.raises(ValueError, match="float64"):
        ClosedLoop(grt, sim, N=8, mpc=BatchMPC(N=8, dtype=torch.float32))
    with pytest.raises(ValueError, match="float64"):
        ClosedLoop(grt, sim, N=8, mpc=BatchMPC(N=12))
    ClosedLoop(grt, sim, N=8, mpc=BatchMPC(N=8))


@pytest.mark.gpu
def test_ros_adapter_drives_the_real_solver():
    """ros_node.start_mpc_node (mpc_cmd_pub.jl:158-175) with a stub rospy but the REAL KinematicMPC and GPSRefTrajectory on
    the recorded path, closed through the plant the way launch/sim_path_follow.launch wires it: the stub's Rate.sleep() advances the
    simulator by one control period with the last published MPC_cmd and delivers the next state_est.  Checks the topic order of
    :129-147, finite bounded commands, the steer-first feedback (Q6) through the first-step rate limit, and the stop latch (:148-153)."""
    import types
    import torch
    from mkz_mpc_path_follower_amd import ros_node
    from mkz_mpc_path_follower_amd.kinematic_mpc import KinematicMPC
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    N, vt = 8, 6.0
    grt = GPSRefTrajectory(arrays=_path_arrays(), traj_horizon=N, traj_dt=0.2)
    tr = grt.get_global_trajectory_reference()
    i0 = int(np.searchsorted(tr[:, 6], tr[-1, 6] - 25.0))   # 25 m before the end of the path (look-ahead 10.8 m at 6 m/s): the stop flag comes up within the run
    sim = VehicleSimulator(1, X0=tr[i0, 4], Y0=tr[i0, 5], Psi0=tr[i0, 3])
    sim.state[:, 3] = vt
    log = {"published": [], "pubs": {}, "node": None, "sleeps": 0}
    state = {"cb": None}

    def mk(name, **fields):
        def init(self):
            self.header = types.SimpleNamespace(stamp=None)
            for k, v in fields.items():
                setattr(self, k, v)
        return type(name, (), {"__init__": init})

    msgs = types.SimpleNamespace(MPC_cmd=mk("MPC_cmd", accel_cmd=0.0, steer_angle_cmd=0.0), mpc_path=mk("mpc_path", xs=[], ys=[], psis=[]),
                                 state_est=mk("state_est", x=0.0, y=0.0, psi=0.0, v=0.0))
    std_msgs = types.SimpleNamespace(Empty=mk("Empty"))

    class Pub:
        def __init__(self, topic, cls, queue_size=None):
            log["pubs"][topic] = (cls.__name__, queue_size); self.topic = topic
        def publish(self, m):
            log["published"].append((self.topic, m))
            if self.topic == "mpc_cmd":                 # vehicle_simulator.py:51-55
                sim._mpc_cmd_callback(m.accel_cmd, m.steer_angle_cmd)

    class Rate:
        def __init__(self, hz): assert hz == 10.0
        def sleep(self):                                # 0.1 s of plant, then the next state_est (vehicle_simulator.py:40-48)
            log["sleeps"] += 1
            sim._update_vehicle_model(10)
            s = sim.state_est(0)
            m = msgs.state_est(); m.x, m.y, m.psi, m.v = s.x, s.y, s.psi, s.v
            state["cb"](m)

    params = {"mat_waypoints": "unused.mat", "track_using_time": False, "target_vel": vt}
    rospy = types.SimpleNamespace(has_param=lambda k: k in params, get_param=lambda k, d=None: params.get(k, d),
                                  init_node=lambda n: log.__setitem__("node", n), Publisher=Pub,
                                  Subscriber=lambda topic, cls, cb, queue_size=None: state.__setitem__("cb", cb), Rate=Rate,
                                  is_shutdown=lambda: False, get_rostime=lambda: 0.0)
    mpc = KinematicMPC(N=N)
    assert mpc.status == "Optimal" and abs(mpc.cost - 15738.467) < 1e-2    # the module-load solve (MKZMPCPathFollower.jl:125-128)
    node = ros_node.start_mpc_node(N=N, rospy=rospy, msgs=msgs, std_msgs=std_msgs, grt=grt, mpc=mpc, max_steps=60)
    assert log["node"] == "dbw_mpc_pf" and log["sleeps"] == 60
    topics = [t for t, _ in log["published"]]
    assert topics[0] == "enable" and topics.count("enable") == 1
    body = topics[1:]
    n_solve = body.count("target_path")
    assert 5 <= n_solve < 59, n_solve                   # first pass idles (no state yet, :89-92); the latch comes up before the run ends
    assert body == ["mpc_cmd", "target_path", "mpc_path"] * n_solve + ["mpc_cmd"] * (59 - n_solve)
    assert node.command_stop
    cmds = [(m.accel_cmd, m.steer_angle_cmd) for t, m in log["published"] if t == "mpc_cmd"]
    solved, latched = np.array(cmds[:n_solve]), cmds[n_solve:]
    assert np.isfinite(solved).all() and np.abs(solved[:, 0]).max() <= 1.0 + 1e-8 and np.abs(solved[:, 1]).max() <= 0.5 + 1e-8
    assert all(c == (-1.0, 0.0) for c in latched)
    # update_current_input(df_opt, a_opt) is steer-first (Q6): were the two swapped, the anchor of the first-step rate rows (:75-76, :82-83)
    # would be wrong and consecutive commands would violate |d acc| <= 0.15, |d d_f| <= 0.05
    prev = np.vstack([[0.0, 0.0], solved[:-1]])
    assert np.abs(solved[:, 0] - prev[:, 0]).max() <= 0.15 + 1e-7 and np.abs(solved[:, 1] - prev[:, 1]).max() <= 0.05 + 1e-7
    paths = [m for t, m in log["published"] if t == "mpc_path"]
    assert all(len(p.xs) == N + 1 and len(p.ys) == N + 1 and len(p.psis) == N + 1 and np.isfinite(p.xs).all() for p in paths)
    # the prediction starts at the state the solve was given (state 0 of the horizon)
    tgt = [m for t, m in log["published"] if t == "target_path"]
    assert all(len(p.xs) == N + 1 for p in tgt)
    assert float(sim.state[0, 3]) < vt                  # the latched -1 m/s^2 is slowing the car down
