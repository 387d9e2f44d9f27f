"""Plain-data mirrors of the reference's ROS message types (msg/*.msg); no ROS dependency."""
from dataclasses import dataclass, field
from typing import List


@dataclass
class Header:
    seq: int = 0
    stamp: float = 0.0
    frame_id: str = ""


@dataclass
class StateEst:
    """msg/state_est.msg:1-9"""
    header: Header = field(default_factory=Header)
    x: float = 0.0
    y: float = 0.0
    psi: float = 0.0
    v: float = 0.0
    lat: float = 0.0
    lon: float = 0.0
    a: float = 0.0
    df: float = 0.0


@dataclass
class MPCCmd:
    """msg/MPC_cmd.msg:1-3"""
    header: Header = field(default_factory=Header)
    accel_cmd: float = 0.0
    steer_angle_cmd: float = 0.0


@dataclass
class MPCPath:
    """msg/mpc_path.msg:1-4: three parallel float64[] of length N+1"""
    header: Header = field(default_factory=Header)
    xs: List[float] = field(default_factory=list)
    ys: List[float] = field(default_factory=list)
    psis: List[float] = field(default_factory=list)


@dataclass
class AccStamped:
    """msg/acc_stamped.msg:1-2 (published by the low-level controller as `filtered_accel` / `req_accel`,
    src/LowLevelController.cpp:103-104; the Gazebo-demo MPC nodes subscribe to it).  Not on the hot path: mirrored for completeness
    of the wire types of SURVEY.md 8(a) row a16."""
    header: Header = field(default_factory=Header)
    accel_value: float = 0.0


STD_MSGS_HEADER_MD5 = "2176decaecbce78abc3b96ef049fabed"  # std_msgs/Header (ROS 1)


def ros_md5(cls):
    """MD5 a ROS 1 install would compute for the message type `cls` mirrors (genmsg rule: the definition text with comments
    stripped and every embedded message type replaced by that type's MD5).  Unverified against a ROS install (none here); they
    are the sums a `rospy` subscriber would check on connection: MPC_cmd 4cc98133..., state_est 3fc895b2..., mpc_path
    bfa6be66..., acc_stamped 55b5fb89..."""
    import hashlib
    from dataclasses import fields
    lines = []
    for f in fields(cls):
        if f.type is Header or f.type == "Header":
            lines.append("%s %s" % (STD_MSGS_HEADER_MD5, f.name))
        elif f.type in (List[float], "List[float]"):
            lines.append("float64[] %s" % f.name)
        else:
            lines.append("float64 %s" % f.name)
    return hashlib.md5("\n".join(lines).encode()).hexdigest()
