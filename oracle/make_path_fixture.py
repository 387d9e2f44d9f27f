"""oracle/make_path_fixture.py -- TEST INFRASTRUCTURE.  Writes tests/golden/path1_decimated.npz.

A recorded path of the reference (paths/path1_6_20.mat: t, lat, lon, psi, x, y at 100 Hz) decimated 3:1 (2235
samples) so that the fixture stays small: INPUT DATA for the waypoint tests, plus the reference's own stored
x, y columns, which pin the lat/lon projection of ref_gps_traj.py:33-52.  Run from the repo root in the build
container (needs /root/reference):  python oracle/make_path_fixture.py
"""
import os

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sio.loadmat("/root/reference/paths/path1_6_20.mat")
sl = slice(0, None, 3)
out = {k: np.ravel(d[k])[sl].astype(np.float64) for k in ("t", "lat", "lon", "psi", "x", "y", "v")}
out["lat0"], out["lon0"] = np.float64(37.917929), np.float64(-122.331798)  # launch/path_follow.launch:19-20
p = os.path.join(ROOT, "tests", "golden", "path1_decimated.npz")
np.savez_compressed(p, **out)
print("wrote", p, os.path.getsize(p), "bytes,", len(out["t"]), "samples")
