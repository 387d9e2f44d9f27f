"""oracle/make_path_fixture.py -- TEST INFRASTRUCTURE.  Writes tests/golden/path{1,2,3}_decimated.npz.

The recorded paths of the reference (paths/path{1,2,3}_6_20.mat: t, lat, lon, psi, x, y, v at 100 Hz) decimated 3:1 (2235 / 2198 / 2209
samples) so that the fixtures stay small: INPUT DATA for the waypoint and closed-loop tests, plus the reference's own stored
x, y columns, which pin the lat/lon projection of ref_gps_traj.py:33-52.  path3 is the path of the reference's own verification
scenario (launch/sim_path_follow.launch:13; tests/test_scenario.py), path2 the second source of time-mode reference windows for the
out-of-distribution sweep (tools/ood_sweep.py).  Run from the repo root in the build container (needs /root/reference):
python oracle/make_path_fixture.py
"""
import os

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for k in (1, 2, 3):
    d = sio.loadmat("/root/reference/paths/path%d_6_20.mat" % k)
    sl = slice(0, None, 3)
    out = {c: np.ravel(d[c])[sl].astype(np.float64) for c in ("t", "lat", "lon", "psi", "x", "y", "v")}
    out["lat0"], out["lon0"] = np.float64(37.917929), np.float64(-122.331798)  # launch/path_follow.launch:19-20
    p = os.path.join(ROOT, "tests", "golden", "path%d_decimated.npz" % k)
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes,", len(out["t"]), "samples")
