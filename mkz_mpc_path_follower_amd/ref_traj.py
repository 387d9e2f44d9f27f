"""Mirror of the reference's waypoint helper scripts/gps_utils/ref_gps_traj.py (GPSRefTrajectory).

Same constructor data flow (load the recorded path, project lat/lon to the local XY frame, build
the cumulative arclength -- ref_gps_traj.py:33-52, 87-106) and the same
`get_waypoints(X_init, Y_init, yaw_init, v_target=None)` call (:131-142), but the rosparams
(lat0, lon0, yaw0, is_heading_info) are constructor arguments (defaults: launch/path_follow.launch:15-21)
and the look-ahead itself runs on the MI355X for B vehicles at once (kmpc_waypoints_batch);
`get_waypoints` is the B = 1 case.  No CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

LAT0, LON0, YAW0 = 37.917929, -122.331798, 0.0  # launch/path_follow.launch:19-21


def latlon_to_XY(lat0, lon0, lat1, lon1):
    """equirectangular projection, ref_gps_traj.py:33-52 (vectorised)"""
    R_earth = 6371000.0
    delta_lat = np.radians(lat1 - lat0)
    delta_lon = np.radians(lon1 - lon0)
    lat_avg = 0.5 * (np.radians(lat1) + np.radians(lat0))
    return R_earth * delta_lon * np.cos(lat_avg), R_earth * delta_lat


def path_arrays(tms, lats, lons, yaws, lat0=LAT0, lon0=LON0):
    """columns of GPSRefTrajectory.trajectory (:106): t, lat, lon, yaw, X, Y, cdist"""
    tms, lats, lons, yaws = (np.ravel(np.asarray(a, dtype=np.float64)) for a in (tms, lats, lons, yaws))
    X, Y = latlon_to_XY(lat0, lon0, lats, lons)
    # :95-100  s_0 = 0, s_i = s_{i-1} + dist(z_i, z_{i-1})  (sequential sum, as in the reference's loop)
    step = np.sqrt((X[1:] - X[:-1]) ** 2 + (Y[1:] - Y[:-1]) ** 2)
    cd = np.zeros_like(X)
    acc = 0.0
    for i, d in enumerate(step):
        acc = d + acc
        cd[i + 1] = acc
    return tms, lats, lons, yaws, X, Y, cd


class GPSRefTrajectory:
    def __init__(self, mat_filename=None, traj_horizon=8, traj_dt=0.2, lat0=LAT0, lon0=LON0, yaw0=YAW0,
                 use_heading=False, arrays=None, device=0):
        if mat_filename is None and arrays is None:
            raise ValueError("Invalid matfile specified.")  # :68-69
        if use_heading:
            raise NotImplementedError("is_heading_info=True is read but never used by the reference (:75)")
        self.traj_horizon, self.traj_dt = int(traj_horizon), float(traj_dt)  # :77-78
        if arrays is None:
            import scipy.io as sio
            dd = sio.loadmat(mat_filename)  # :88
            arrays = dict(t=dd["t"], lat=dd["lat"], lon=dd["lon"], psi=dd["psi"])
        t, lat, lon, psi, X, Y, cd = path_arrays(arrays["t"], arrays["lat"], arrays["lon"], arrays["psi"], lat0, lon0)
        self.trajectory = np.column_stack((t, lat, lon, psi, X, Y, cd))  # :106
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GPSRefTrajectory needs an MI355X; no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        cols = [np.ascontiguousarray(self.trajectory[:, i]) for i in (0, 4, 5, 3, 6)]
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(self._lib.kmpc_path_create(int(device), len(t), dp(cols[0]), dp(cols[1]), dp(cols[2]), dp(cols[3]),
                                              dp(cols[4]), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kmpc_path_destroy(self._h)
            self._h = None

    __del__ = close

    def get_global_trajectory_reference(self): return self.trajectory  # :116-117
    def get_Xs(self): return self.trajectory[:, 4]
    def get_Ys(self): return self.trajectory[:, 5]
    def get_psis(self): return self.trajectory[:, 3]

    def get_waypoints_batch(self, pose, v_target=None, want_closest=False):
        """pose [B,3] (X_init, Y_init, yaw_init), v_target [B] or None -> ref [B,H+1,3] (device), stop [B] int32"""
        pose = torch.as_tensor(pose, dtype=torch.float64, device=self.device).contiguous()
        if pose.dim() != 2 or pose.shape[1] != 3:
            raise ValueError("pose: expected [B,3], got %s" % (tuple(pose.shape),))
        B = pose.shape[0]
        vt = None if v_target is None else torch.as_tensor(v_target, dtype=torch.float64, device=self.device).contiguous()
        if vt is not None and tuple(vt.shape) != (B,):
            raise ValueError("v_target: expected [%d], got %s" % (B, tuple(vt.shape)))
        ref = torch.empty((B, self.traj_horizon + 1, 3), dtype=torch.float64, device=self.device)
        stop = torch.empty((B,), dtype=torch.int32, device=self.device)
        closest = torch.empty((B,), dtype=torch.int32, device=self.device) if want_closest else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.kmpc_waypoints_batch(self._h, B, self.traj_horizon, self.traj_dt, p(pose), p(vt), p(ref), p(stop),
                                            p(closest), stream)
        if rc != 0:
            raise _lib.KmpcError(self._lib.kmpc_path_last_error(self._h).decode())
        return (ref, stop, closest) if want_closest else (ref, stop)

    # :131-142 -- same signature and return tuple as the reference
    def get_waypoints(self, X_init, Y_init, yaw_init, v_target=None):
        ref, stop = self.get_waypoints_batch([[X_init, Y_init, yaw_init]], None if v_target is None else [v_target])
        r = ref[0].cpu().numpy()
        return r[:, 0].copy(), r[:, 1].copy(), r[:, 2].copy(), bool(stop[0].item())
