"""Batched mirror of scripts/vehicle_simulator.py (VehicleSimulator): the dynamic-bicycle plant the reference's
sim_path_follow.launch runs against the MPC node.  Attribute names and update methods follow the reference
(X, Y, psi, vx, vy, wz, acc, df, acc_des, df_des; `_mpc_cmd_callback`, `_update_vehicle_model`), but every
attribute is a length-B device tensor and the ODE runs on the MI355X (kmpc_sim_advance_batch).  No CPU fallback.
"""
import ctypes as C

import torch

from . import _lib
from .messages import StateEst

X0, Y0, PSI0 = -300.0, -450.0, 1.0  # vehicle_simulator.py:28-30 (rosparam defaults)


class VehicleSimulator:
    dt_model = 0.01  # :24

    def __init__(self, B=1, X0=X0, Y0=Y0, Psi0=PSI0, device=0):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("VehicleSimulator needs an MI355X; no CPU fallback")
        self.device = torch.device("cuda", device)
        self.B = int(B)
        # [B,8]: X, Y, psi, vx, vy, wz, acc, df   (:18-34; velocities and actuators start at 0)
        self.state = torch.zeros((self.B, 8), dtype=torch.float64, device=self.device)
        self.state[:, 0] = torch.as_tensor(X0, dtype=torch.float64, device=self.device)
        self.state[:, 1] = torch.as_tensor(Y0, dtype=torch.float64, device=self.device)
        self.state[:, 2] = torch.as_tensor(Psi0, dtype=torch.float64, device=self.device)
        self.cmd = torch.zeros((self.B, 2), dtype=torch.float64, device=self.device)  # acc_des, df_des (:21-22)

    # views named as in the reference
    X = property(lambda s: s.state[:, 0]); Y = property(lambda s: s.state[:, 1]); psi = property(lambda s: s.state[:, 2])
    vx = property(lambda s: s.state[:, 3]); vy = property(lambda s: s.state[:, 4]); wz = property(lambda s: s.state[:, 5])
    acc = property(lambda s: s.state[:, 6]); df = property(lambda s: s.state[:, 7])
    acc_des = property(lambda s: s.cmd[:, 0]); df_des = property(lambda s: s.cmd[:, 1])

    def _mpc_cmd_callback(self, accel_cmd, steer_angle_cmd):  # :51-55, for all vehicles
        self.cmd[:, 0] = torch.as_tensor(accel_cmd, dtype=torch.float64, device=self.device)
        self.cmd[:, 1] = torch.as_tensor(steer_angle_cmd, dtype=torch.float64, device=self.device)

    def _update_vehicle_model(self, n_updates=1):
        """n_updates passes of :58-107 (each 10 Euler sub-steps of 1 ms + the actuator lag :109-113)"""
        for t, w in ((self.state, 8), (self.cmd, 2)):  # `state` and `cmd` are plain attributes: what reaches the kernel is a raw pointer
            if not (t.dtype == torch.float64 and tuple(t.shape) == (self.B, w) and t.is_contiguous() and t.device == self.device):
                raise ValueError("state [B,8] / cmd [B,2] must stay contiguous float64 tensors on %s (write into them with copy_)" % self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.kmpc_sim_advance_batch(self.device.index, self.B, C.c_void_p(self.state.data_ptr()),
                                              C.c_void_p(self.cmd.data_ptr()), int(n_updates), stream)
        _lib.check(rc)

    def state_est(self, i=0):
        """the state_est message of vehicle i (:40-48)"""
        s = self.state[i].cpu().numpy()
        return StateEst(x=float(s[0]), y=float(s[1]), psi=float(s[2]), v=float(s[3]), a=float(s[6]), df=float(s[7]))
