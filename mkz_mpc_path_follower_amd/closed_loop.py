"""Closed loop of launch/sim_path_follow.launch for B vehicles at once, entirely on the device:
state_est (vehicle_sim.VehicleSimulator) -> waypoints (ref_traj.GPSRefTrajectory) -> MPC (solver.BatchMPC, warm-started)
-> MPC_cmd -> simulator.  One `step()` is one pass of the 10 Hz loop of mpc_cmd_pub.jl:88-153 for every vehicle,
followed by 0.1 s of plant time (10 model updates at 100 Hz, vehicle_simulator.py:24-26).

Protocol details kept from the reference node: the command is published regardless of the solver status (Q7); the
rate-limit anchor is the last *command*, not the measured actuator state (:140, Q7); the stop flag of the waypoint
helper latches and overrides the command with accel -1.0 / steer 0.0 (:100-103, :148-153), per vehicle; warm start
from the previous primal solution (JuMP keeps values, Q9).
"""
import ctypes as C

import torch

from . import _lib
from .solver import BatchMPC


class ClosedLoop:
    def __init__(self, grt, sim, N=8, target_vel=0.0, track_with_time=False, weights=(9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0),
                 mpc=None, **options):
        if grt.traj_horizon != N:
            raise ValueError("waypoint horizon %d != MPC horizon %d (Q10: the reference passes them separately)" % (grt.traj_horizon, N))
        self.grt, self.sim, self.N = grt, sim, int(N)
        self.B = sim.B
        self.mpc = mpc if mpc is not None else BatchMPC(N=N, dtype=torch.float64, device=sim.device.index, weights=weights, **options)
        # kmpc_command_batch reads the solver's first inputs as [B,2] doubles on the plant's device: a caller-supplied solver must match
        if self.mpc.dtype != torch.float64 or self.mpc.N != self.N or self.mpc.device != sim.device or sim.device.index is None:
            raise ValueError("ClosedLoop needs a float64 BatchMPC with horizon %d on %s (got %s, N=%d, %s)"
                             % (self.N, sim.device, self.mpc.dtype, self.mpc.N, self.mpc.device))
        if grt.device != sim.device:
            raise ValueError("waypoint helper on %s, plant on %s: the loop runs on one device" % (grt.device, sim.device))
        self.track_with_time = track_with_time
        self.des_speed = float(target_vel) if target_vel > 0.0 else 0.0  # mpc_cmd_pub.jl:58-62
        dev = sim.device
        self.v_target = torch.full((self.B,), self.des_speed, dtype=torch.float64, device=dev)
        self.u_prev = torch.zeros((self.B, 2), dtype=torch.float64, device=dev)       # (acc, d_f): update_current_input starts at 0
        self.warm_U = torch.zeros((self.B, self.N, 2), dtype=torch.float64, device=dev)
        self.command_stop = torch.zeros((self.B,), dtype=torch.bool, device=dev)   # the stop latch (one byte per vehicle: kmpc_command_batch's uint8)
        self._lib = _lib.load()
        self.have_warm = False
        self.out = None
        self.k = 0

    def step(self, plant_updates=10, time_solve=False):
        """time_solve=True brackets the solve with device synchronisations and returns its wall time (`solve_s`)"""
        import time
        st = self.sim.state
        pose = st[:, 0:3].contiguous()
        ref, stop = self.grt.get_waypoints_batch(pose, None if self.track_with_time else self.v_target)
        z0 = st[:, 0:4].contiguous()                                                # x, y, psi, v = vx  (state_est, :43-46 of the simulator)
        if time_solve:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        self.out = self.mpc.solve(z0, ref, self.v_target, self.u_prev, warm_U=self.warm_U, warm=self.have_warm, out=self.out)
        solve_s = None
        if time_solve:
            torch.cuda.synchronize()
            solve_s = time.perf_counter() - t0
        self.have_warm = True
        # stop latch (:100-103), command selection (:148-153) and update_current_input (:140, only on the solve branch): one kernel, straight
        # into the plant's command buffer
        cmd = self.sim.cmd
        u0 = self.out["u0"]
        assert u0.dtype == torch.float64 and u0.is_contiguous() and u0.device == cmd.device and u0.shape == (self.B, 2)
        stream = C.c_void_p(torch.cuda.current_stream(cmd.device).cuda_stream)
        _lib.check(self._lib.kmpc_command_batch(cmd.device.index, self.B, C.c_void_p(u0.data_ptr()), C.c_void_p(stop.data_ptr()),
                                                C.c_void_p(self.command_stop.data_ptr()), C.c_void_p(self.u_prev.data_ptr()),
                                                C.c_void_p(cmd.data_ptr()), stream))
        self.sim._update_vehicle_model(plant_updates)
        self.k += 1
        return dict(ref=ref, cmd=cmd, status=self.out["status"], iters=self.out["iters"], cost=self.out["cost"], solve_s=solve_s)
