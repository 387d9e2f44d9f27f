"""rospy adapter: the topic I/O of scripts/mpc_cmd_pub.jl around node.MPCNode.

ROS is not part of this image, so nothing here is imported by the package; `rospy` and the reference's generated message
classes (mkz_mpc_path_follower.msg, std_msgs.msg) are looked up when `start_mpc_node()` runs -- on a ROS machine this file
is the drop-in for `rosrun mkz_mpc_path_follower mpc_cmd_pub.jl`:

    rosparams   mat_waypoints, track_using_time, target_vel (mpc_cmd_pub.jl:16-28; scripts_dir is not needed),
                lat0 / lon0 / yaw0 (ref_gps_traj.py:71-73, launch/path_follow.launch:19-21)
    node        "dbw_mpc_pf" (:159)
    subscribes  state_est (state_est, queue 2)                       (:170)
    publishes   mpc_cmd (MPC_cmd), enable (Empty), target_path and mpc_path (mpc_path), all queue 2   (:160-168)
    rate        10 Hz (:87), `enable` published once before the loop (:172)
"""
from .messages import StateEst
from .node import MPCNode


def _require_param(rospy, name, msg):
    if not rospy.has_param(name):
        raise RuntimeError(msg)  # mpc_cmd_pub.jl:16-35 call error() with these texts
    return rospy.get_param(name)


def start_mpc_node(N=8, rospy=None, msgs=None, std_msgs=None, grt=None, mpc=None, max_steps=None):
    """`rospy`, `msgs`, `std_msgs`, `grt`, `mpc` are injection points for tests; by default they are imported / built here."""
    if rospy is None:
        import rospy  # noqa: F811  (only on a ROS machine)
    if msgs is None:
        import mkz_mpc_path_follower.msg as msgs   # the reference package's generated messages
    if std_msgs is None:
        import std_msgs.msg as std_msgs
    mat_fname = _require_param(rospy, "mat_waypoints", "No Matfile of waypoints provided!")
    if not (rospy.has_param("track_using_time") and rospy.has_param("target_vel")):
        raise RuntimeError("Invalid rosparam trajectory definition: track_using_time and target_vel")
    track_with_time = rospy.get_param("track_using_time")
    target_vel = rospy.get_param("target_vel")
    if grt is None:
        from .ref_traj import GPSRefTrajectory, LAT0, LON0, YAW0
        grt = GPSRefTrajectory(mat_filename=mat_fname, traj_horizon=N,
                               lat0=rospy.get_param("lat0", LAT0), lon0=rospy.get_param("lon0", LON0), yaw0=rospy.get_param("yaw0", YAW0))

    rospy.init_node("dbw_mpc_pf")
    pubs = {"mpc_cmd": rospy.Publisher("mpc_cmd", msgs.MPC_cmd, queue_size=2),
            "enable": rospy.Publisher("enable", std_msgs.Empty, queue_size=2),
            "target_path": rospy.Publisher("target_path", msgs.mpc_path, queue_size=2),
            "mpc_path": rospy.Publisher("mpc_path", msgs.mpc_path, queue_size=2)}

    def publish(topic, m):
        if topic == "enable":
            pubs[topic].publish(std_msgs.Empty())
        elif topic == "mpc_cmd":
            out = msgs.MPC_cmd()   # the reference reads get_rostime() (:124) but never stamps the header: left default, as there
            out.accel_cmd, out.steer_angle_cmd = m.accel_cmd, m.steer_angle_cmd
            pubs[topic].publish(out)
        else:
            out = msgs.mpc_path()
            out.xs, out.ys, out.psis = list(m.xs), list(m.ys), list(m.psis)
            pubs[topic].publish(out)

    node = MPCNode(grt.get_waypoints, publish, N=N, target_vel=target_vel, track_with_time=track_with_time, mpc=mpc)
    rospy.Subscriber("state_est", msgs.state_est,
                     lambda m: node.state_est_callback(StateEst(x=m.x, y=m.y, psi=m.psi, v=m.v)), queue_size=2)
    loop_rate = rospy.Rate(10.0)
    k = 0
    while not rospy.is_shutdown() and (max_steps is None or k < max_steps):
        node.step()
        loop_rate.sleep()
        k += 1
    return node


if __name__ == "__main__":
    try:
        start_mpc_node()
    except Exception as x:  # mpc_cmd_pub.jl:177-183 prints and exits
        print(x)
