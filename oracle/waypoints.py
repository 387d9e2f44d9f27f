"""oracle/waypoints.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's waypoint helper, scripts/gps_utils/ref_gps_traj.py, function by function
(the reference file is Python 2 + rospy and cannot be imported here: SURVEY.md 8(c)).  Pinned by the reference's
own data: the recorded paths store x, y next to lat, lon, and the restated projection reproduces them
(tests/test_waypoints.py).
"""
import math

import numpy as np


def latlon_to_XY(lat0, lon0, lat1, lon1):
    """ref_gps_traj.py:33-52, scalar math as in the reference"""
    R_earth = 6371000
    delta_lat = math.radians(lat1 - lat0)
    delta_lon = math.radians(lon1 - lon0)
    lat_avg = 0.5 * (math.radians(lat1) + math.radians(lat0))
    return R_earth * delta_lon * math.cos(lat_avg), R_earth * delta_lat


def build_trajectory(tms, lats, lons, yaws, lat0, lon0):
    """ref_gps_traj.py:87-106"""
    Xs, Ys, cdists = [], [], []
    for i in range(len(lats)):
        X, Y = latlon_to_XY(lat0, lon0, lats[i], lons[i])
        if len(Xs) == 0:
            cdists.append(0.0)
        else:
            cdists.append(math.sqrt((X - Xs[-1]) ** 2 + (Y - Ys[-1]) ** 2) + cdists[-1])
        Xs.append(X)
        Ys.append(Y)
    return np.column_stack((tms, lats, lons, yaws, Xs, Ys, cdists))


def fix_heading_wraparound(psi_ref, psi_current):
    """ref_gps_traj.py:204-218"""
    check_1 = np.max(np.fabs(np.diff(psi_ref))) < np.pi
    check_2 = np.max(np.fabs(psi_ref - psi_current)) < np.pi
    if check_1 and check_2:
        return psi_ref
    psi_ref = psi_ref.copy()
    for i in range(len(psi_ref)):
        p = psi_ref[i]
        cands = np.array([p, p + 2 * np.pi, p - 2 * np.pi])
        psi_ref[i] = cands[np.argmin(np.fabs(cands - psi_current))]
    return psi_ref


def get_waypoints(traj, X_init, Y_init, yaw_init, v_target=None, traj_horizon=8, traj_dt=0.2):
    """ref_gps_traj.py:131-142 + :172-201; returns (x, y, psi, stop_cmd, closest_index)"""
    XY = traj[:, 4:6]
    diff = np.sum((XY - np.array([[X_init, Y_init]])) ** 2, axis=1)
    ci = int(np.argmin(diff))
    if v_target is not None:
        start = traj[ci, 6]
        grid = [x * traj_dt * v_target + start for x in range(1, traj_horizon + 2)]  # :175
        xp = traj[:, 6]
    else:
        start = traj[ci, 0]
        grid = [h * traj_dt + start for h in range(0, traj_horizon + 1)]  # :191
        xp = traj[:, 0]
    xi = np.interp(grid, xp, traj[:, 4])
    yi = np.interp(grid, xp, traj[:, 5])
    pr = np.interp(grid, xp, traj[:, 3])
    pi_ = fix_heading_wraparound(pr, yaw_init)
    stop = bool(xi[-1] == traj[-1, 4] and yi[-1] == traj[-1, 5])  # :182-184
    return xi, yi, pi_, stop, ci
