"""Control-loop mirror of scripts/mpc_cmd_pub.jl (ROS-free).

`MPCNode.step()` is one iteration of pub_loop (:86-157): snapshot the latest state_est, fetch
N+1 waypoints, update the model, solve, publish MPC_cmd / target_path / mpc_path, feed the
command back as the rate-limit anchor, and latch the stop command.  Transport is injected:
`publish(topic, msg)` is any callable; a rospy adapter is a few lines.
"""
from .kinematic_mpc import KinematicMPC
from .messages import MPCCmd, MPCPath, StateEst


class MPCNode:
    def __init__(self, get_waypoints, publish, N=8, target_vel=0.0, track_with_time=False, mpc=None):
        self.kmpc = mpc if mpc is not None else KinematicMPC(N=N)
        self.kmpc.update_cost(9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0)  # mpc_cmd_pub.jl:49
        self.get_waypoints = get_waypoints   # (x, y, psi[, v_target]) -> (x_ref, y_ref, psi_ref, stop_cmd)
        self.publish = publish
        self.track_with_time = track_with_time
        self.des_speed = target_vel if target_vel > 0.0 else 0.0  # :58-62
        self.received_reference = False
        self.command_stop = False
        self.x_curr = self.y_curr = self.psi_curr = self.v_curr = 0.0
        self.publish("enable", None)  # :172

    # mpc_cmd_pub.jl:72-84 (the ref_lock flag is unnecessary: step() snapshots by value)
    def state_est_callback(self, msg: StateEst):
        self.x_curr, self.y_curr, self.psi_curr, self.v_curr = msg.x, msg.y, msg.psi, msg.v
        self.received_reference = True

    # one pass of the 10 Hz loop, mpc_cmd_pub.jl:88-153
    def step(self):
        if not self.received_reference:
            return None
        x, y, psi, v = self.x_curr, self.y_curr, self.psi_curr, self.v_curr
        if not self.track_with_time:
            x_ref, y_ref, psi_ref, stop_cmd = self.get_waypoints(x, y, psi, self.des_speed)
        else:
            x_ref, y_ref, psi_ref, stop_cmd = self.get_waypoints(x, y, psi)
        if stop_cmd:
            self.command_stop = True
        self.kmpc.update_init_cond(x, y, psi, v)
        self.kmpc.update_reference(x_ref, y_ref, psi_ref, self.des_speed)
        if not self.command_stop:
            a_opt, df_opt, is_opt = self.kmpc.solve_model()
            cmd = MPCCmd(accel_cmd=a_opt, steer_angle_cmd=df_opt)  # published regardless of status (Q7)
            self.publish("mpc_cmd", cmd)
            self.publish("target_path", MPCPath(xs=list(x_ref), ys=list(y_ref), psis=list(psi_ref)))
            self.kmpc.update_current_input(df_opt, a_opt)  # steer first (Q6)
            res = self.kmpc.get_solver_results()
            self.publish("mpc_path", MPCPath(xs=list(res[0]), ys=list(res[1]), psis=list(res[3])))
            return cmd
        cmd = MPCCmd(accel_cmd=-1.0, steer_angle_cmd=0.0)  # stop latch, :148-153
        self.publish("mpc_cmd", cmd)
        return cmd
