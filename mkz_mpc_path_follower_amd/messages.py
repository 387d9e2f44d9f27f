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
