"""Host-side logic that needs no GPU: synthetic batches, sharding, the control-loop mirror of
scripts/mpc_cmd_pub.jl (with a stand-in MPC object), and the N>1 all-gather on gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mkz_mpc_path_follower_amd.dist import SolutionGather, all_gather_solutions, shard_range
from mkz_mpc_path_follower_amd.messages import MPCCmd, MPCPath, StateEst
from mkz_mpc_path_follower_amd.node import MPCNode
from mkz_mpc_path_follower_amd.synthetic import make_batch, straight_line_case


def test_synthetic_batch_is_seeded_and_feasible(oracle):
    a, b = make_batch(256, 20, cfg_id=2), make_batch(256, 20, cfg_id=2)
    for k in ("z0", "ref", "v_target", "u_prev"):
        assert np.array_equal(a[k], b[k])
    assert a["z0"].shape == (256, 4) and a["ref"].shape == (256, 21, 3) and a["u_prev"].shape == (256, 2)
    assert (a["z0"][:, 3] >= 0).all() and (a["z0"][:, 3] <= 20).all()
    # reference spacing = v_target * dt (ref_gps_traj.py:175)
    ds = np.linalg.norm(np.diff(a["ref"][:, :, :2], axis=1), axis=2)
    assert np.allclose(ds, a["v_target"][:, None] * 0.2, rtol=2e-3)
    r = oracle.solve_condensed_batch(oracle.params(20), a["z0"][:64], a["ref"][:64], a["v_target"][:64], a["u_prev"][:64], nthreads=8)
    assert (r["status"] == 0).all()
    s = straight_line_case(8)
    assert np.allclose(s["ref"][0, :, 0], 15.0 * 0.2 * np.arange(9))  # MKZMPCPathFollower.jl:36-37


def test_shard_ranges_cover_the_batch():
    for B in (1, 7, 4096, 2097152):
        for w in (1, 2, 3, 8):
            r = [shard_range(B, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


class FakeMPC:
    """records the call protocol of mpc_cmd_pub.jl:115-141"""

    def __init__(self, N=8):
        self.N, self.calls = N, []

    def update_cost(self, *w): self.calls.append(("update_cost", w))
    def update_init_cond(self, *a): self.calls.append(("update_init_cond", a))
    def update_reference(self, x, y, p, v): self.calls.append(("update_reference", (len(x), len(y), len(p), v)))
    def update_current_input(self, swa, acc): self.calls.append(("update_current_input", (swa, acc)))
    def solve_model(self): self.calls.append(("solve_model", ())); return 0.12, -0.03, "Optimal"

    def get_solver_results(self):
        n = self.N + 1
        z = np.zeros(n)
        return (z + 1, z + 2, z + 3, z + 4, z, z, z, np.zeros(self.N), np.zeros(self.N))


def test_node_loop_protocol_and_stop_latch():
    pub = []
    stop = {"v": False}

    def wp(x, y, psi, v=None):
        n = 9
        return np.arange(n) + x, np.zeros(n) + y, np.zeros(n) + psi, stop["v"]

    mpc = FakeMPC()
    node = MPCNode(wp, lambda t, m: pub.append((t, m)), target_vel=5.0, mpc=mpc)
    assert pub[0][0] == "enable"                                         # mpc_cmd_pub.jl:172
    assert mpc.calls[0] == ("update_cost", (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0))  # :49
    assert node.step() is None                                           # nothing until a state arrives (:88-91)
    node.state_est_callback(StateEst(x=1.0, y=2.0, psi=0.3, v=4.0))
    cmd = node.step()
    assert isinstance(cmd, MPCCmd) and (cmd.accel_cmd, cmd.steer_angle_cmd) == (0.12, -0.03)
    names = [c[0] for c in mpc.calls[1:]]
    assert names == ["update_init_cond", "update_reference", "solve_model", "update_current_input"]
    assert mpc.calls[1][1] == (1.0, 2.0, 0.3, 4.0)
    assert mpc.calls[2][1] == (9, 9, 9, 5.0)
    assert mpc.calls[4][1] == (-0.03, 0.12)                              # steer first (Q6, :140)
    topics = [t for t, _ in pub[1:]]
    assert topics == ["mpc_cmd", "target_path", "mpc_path"]
    path = pub[-1][1]
    assert isinstance(path, MPCPath) and path.xs[0] == 1 and path.ys[0] == 2 and path.psis[0] == 4  # res[1], res[2], res[4]
    stop["v"] = True                                                     # stop latch (:102-104, :148-153)
    cmd = node.step()
    assert (cmd.accel_cmd, cmd.steer_angle_cmd) == (-1.0, 0.0)
    stop["v"] = False
    n_solves = sum(1 for c in mpc.calls if c[0] == "solve_model")
    cmd = node.step()
    assert (cmd.accel_cmd, cmd.steer_angle_cmd) == (-1.0, 0.0)           # latched forever
    assert sum(1 for c in mpc.calls if c[0] == "solve_model") == n_solves


def _gather_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(B, rank, world)
    full = torch.arange(B * 2, dtype=torch.float64).reshape(B, 2)
    got = all_gather_solutions(full[lo:hi].clone(), B)
    q.put((rank, bool(torch.equal(got, full))))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_all_gather_of_solution_shards_gloo_world2(B):
    """N>1 path: contiguous shards, one all-gather of the [B/G, 2] (accel, steer) blocks (even and ragged)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + B) % 2000
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _pipeline_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(B, rank, world)
    g = SolutionGather(B)
    local = [torch.zeros((hi - lo, 2), dtype=torch.float64) for _ in range(2)]
    ok = True
    for step in range(5):  # the loop of bench.py: wait(slot) before the slot's source buffer is overwritten, then submit
        s = step & 1
        prev = g.wait(s)
        if step >= 2:
            ok = ok and bool(torch.equal(prev, torch.arange(B * 2, dtype=torch.float64).reshape(B, 2) + 1000.0 * (step - 2)))
        full = torch.arange(B * 2, dtype=torch.float64).reshape(B, 2) + 1000.0 * step
        local[s].copy_(full[lo:hi])
        g.submit(s, local[s])
    for step in (3, 4):
        ok = ok and bool(torch.equal(g.wait(step & 1), torch.arange(B * 2, dtype=torch.float64).reshape(B, 2) + 1000.0 * step))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_pipelined_solution_gather_gloo_world2():
    """bench.py's N>1 loop: the gather of batch k (async, double-buffered) overlaps the solve of batch k+1; every batch's
    gathered block must still be the right one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
    g = SolutionGather(6)  # single process: pass-through
    u = torch.ones((6, 2))
    g.submit(0, u)
    assert g.wait(0) is u


def test_wire_types_mirror_the_msg_files():
    """field names and order of msg/*.msg (the text of each file is three to nine `float64 name` lines) and the MD5 sums the
    genmsg rule gives for them (unverified against a ROS install)"""
    from dataclasses import fields
    from mkz_mpc_path_follower_amd import messages as M
    assert [f.name for f in fields(M.StateEst)] == ["header", "x", "y", "psi", "v", "lat", "lon", "a", "df"]     # msg/state_est.msg:1-9
    assert [f.name for f in fields(M.MPCCmd)] == ["header", "accel_cmd", "steer_angle_cmd"]                     # msg/MPC_cmd.msg:1-3
    assert [f.name for f in fields(M.MPCPath)] == ["header", "xs", "ys", "psis"]                                # msg/mpc_path.msg:1-4
    assert [f.name for f in fields(M.AccStamped)] == ["header", "accel_value"]                                  # msg/acc_stamped.msg:1-2
    assert M.ros_md5(M.MPCCmd) == "4cc9813360048599657ed07f1e3a49c7"
    assert M.ros_md5(M.StateEst) == "3fc895b2d82deff841eec10883ba25f1"
    assert M.ros_md5(M.MPCPath) == "bfa6be669d3684fb46eee30feac22a41"
    assert M.ros_md5(M.AccStamped) == "55b5fb89b56deee865ea283fe7091d26"


def test_spawn_ranks_launch_path(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its N ranks itself (dist.spawn_ranks: a torch.distributed.run child on
    127.0.0.1, before any GPU call).  Same launcher, CPU child script, gloo: world size 2 arrives, the gathered block is right."""
    import json
    import sys
    from mkz_mpc_path_follower_amd.dist import spawn_ranks
    out = str(tmp_path / "probe.json")
    rc = spawn_ranks(os.path.join(os.path.dirname(__file__), "_rank_probe.py"), 2, ["2", out], timeout=300)
    assert rc == 0
    assert json.load(open(out)) == {"n_gpus": 2, "ok": True}
    # a rank count that does not match --gpus is refused rather than silently run on one GPU
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"), "--gpus", "2"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_ros_adapter_topic_surface():
    """the rospy adapter reproduces mpc_cmd_pub.jl's node name, topics, queue sizes, 10 Hz rate, the one-shot
    `enable`, the rosparam checks and the stop latch -- checked against a stub rospy (ROS is not installed here) and a stub solver."""
    import types
    from mkz_mpc_path_follower_amd import ros_node

    log = {"pubs": {}, "subs": {}, "published": [], "rate": None, "node": None, "sleeps": 0}

    class Pub:
        def __init__(self, topic, cls, queue_size=None):
            log["pubs"][topic] = (cls.__name__, queue_size); self.topic = topic
        def publish(self, m):
            log["published"].append((self.topic, m))

    class Rate:
        def __init__(self, hz): log["rate"] = hz
        def sleep(self): log["sleeps"] += 1

    def mk(name, **fields):
        def init(self):
            self.header = types.SimpleNamespace(stamp=None)
            for k, v in fields.items():
                setattr(self, k, v)
        return type(name, (), {"__init__": init})

    params = {"mat_waypoints": "x.mat", "track_using_time": False, "target_vel": 5.0}
    state = {"cb": None}

    def Subscriber(topic, cls, cb, queue_size=None):
        log["subs"][topic] = (cls.__name__, queue_size); state["cb"] = cb

    rospy = types.SimpleNamespace(has_param=lambda k: k in params, get_param=lambda k, d=None: params.get(k, d),
                                  init_node=lambda n: log.__setitem__("node", n), Publisher=Pub, Subscriber=Subscriber, Rate=Rate,
                                  is_shutdown=lambda: False, get_rostime=lambda: 123.0)
    msgs = types.SimpleNamespace(MPC_cmd=mk("MPC_cmd", accel_cmd=0.0, steer_angle_cmd=0.0), mpc_path=mk("mpc_path", xs=[], ys=[], psis=[]),
                                 state_est=mk("state_est", x=0.0, y=0.0, psi=0.0, v=0.0))
    std_msgs = types.SimpleNamespace(Empty=mk("Empty"))

    class FakeMPC:  # the six module functions, recording the call order
        def __init__(self): self.calls = []
        def update_cost(self, *w): self.calls.append(("cost", w))
        def update_init_cond(self, *a): self.calls.append(("init", a))
        def update_reference(self, *a): self.calls.append(("ref", a[3]))
        def update_current_input(self, d, a): self.calls.append(("input", (d, a)))
        def solve_model(self): self.calls.append(("solve",)); return 0.3, -0.02, "Optimal"
        def get_solver_results(self): return ([0.0] * 9,) * 9

    stop_after = {"n": 0}

    class FakeGRT:
        def get_waypoints(self, x, y, psi, v=None):
            stop_after["n"] += 1
            return [0.0] * 9, [0.0] * 9, [0.0] * 9, stop_after["n"] >= 3

    # missing rosparam -> the reference's error text
    bad = types.SimpleNamespace(**{**rospy.__dict__, "has_param": lambda k: False})
    with pytest.raises(RuntimeError, match="No Matfile of waypoints provided!"):
        ros_node.start_mpc_node(rospy=bad, msgs=msgs, std_msgs=std_msgs, grt=FakeGRT(), mpc=FakeMPC(), max_steps=1)

    mpc = FakeMPC()
    # no state received yet: the loop idles (mpc_cmd_pub.jl:89)
    node = ros_node.start_mpc_node(rospy=rospy, msgs=msgs, std_msgs=std_msgs, grt=FakeGRT(), mpc=mpc, max_steps=2)
    assert log["node"] == "dbw_mpc_pf" and log["rate"] == 10.0 and log["sleeps"] == 2
    assert log["pubs"] == {"mpc_cmd": ("MPC_cmd", 2), "enable": ("Empty", 2), "target_path": ("mpc_path", 2), "mpc_path": ("mpc_path", 2)}
    assert log["subs"] == {"state_est": ("state_est", 2)}
    assert [t for t, _ in log["published"]] == ["enable"]
    assert mpc.calls[0] == ("cost", (9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0))
    # feed a state and run 3 more passes: 2 solves, then the stop latch
    m = msgs.state_est(); m.x, m.y, m.psi, m.v = 1.0, 2.0, 0.1, 4.0
    state["cb"](m)
    log["published"].clear()
    for _ in range(3):
        node.step()
    topics = [t for t, _ in log["published"]]
    assert topics == ["mpc_cmd", "target_path", "mpc_path"] * 2 + ["mpc_cmd"]
    first, last = log["published"][0][1], log["published"][-1][1]
    assert (first.accel_cmd, first.steer_angle_cmd) == (0.3, -0.02) and first.header.stamp is None   # never stamped, as in the reference (:129-132)
    assert (last.accel_cmd, last.steer_angle_cmd) == (-1.0, 0.0)
    assert ("input", (-0.02, 0.3)) in mpc.calls   # steer first (Q6)
    assert [c[0] for c in mpc.calls].count("solve") == 2
