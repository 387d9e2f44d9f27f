"""Seeded synthetic (x0, path-segment) batches for tests and bench.py.

The reference ships no benchmark inputs.  The distribution follows SURVEY.md section 8(d):
reference segments are circular arcs with the curvature / speed range of the recorded
paths (paths/*.mat: |kappa| <= 0.10 1/m, v <= 11.3 m/s), sampled the way
scripts/gps_utils/ref_gps_traj.py:172-179 samples waypoints (arclength v_target*dt*k),
initial states are perturbed poses on the arc, and a 5 % "hard" stratum holds the
launch/sim_path_follow.launch:23-25 style cases (3 m lateral offset; standing start).
"""
import numpy as np

SEED_BASE = 20180620


def make_batch(B, N, cfg_id=2, dt=0.2, hard_frac=0.05, dtype=np.float64, seed=None):
    """returns dict z0[B,4], ref[B,N+1,3], v_target[B], u_prev[B,2] (acc, steer), hard[B] (bool)"""
    rng = np.random.default_rng(SEED_BASE + cfg_id if seed is None else seed)
    kappa = rng.uniform(-0.09, 0.09, B)
    v_t = rng.uniform(2.0, 11.0, B)
    th0 = rng.uniform(-np.pi, np.pi, B)
    e_y = rng.normal(0.0, 0.5, B)
    e_psi = rng.normal(0.0, 0.05, B)
    v0 = np.clip(v_t + rng.normal(0.0, 1.0, B), 0.0, 20.0)
    acc_prev = rng.uniform(-0.5, 0.5, B)
    df_prev = np.clip(np.arctan(2.85 * kappa) + rng.normal(0.0, 0.01, B), -0.5, 0.5)
    hard = rng.uniform(size=B) < hard_frac
    kind = rng.uniform(size=B) < 0.5
    sign = np.where(rng.uniform(size=B) < 0.5, -1.0, 1.0)
    # hard stratum: standing start (v0 = 0, previous accel 0) or 3 m lateral offset
    rest = hard & kind
    far = hard & ~kind
    v0 = np.where(rest, 0.0, v0)
    acc_prev = np.where(rest, 0.0, acc_prev)
    e_y = np.where(far, 3.0 * sign, e_y)
    # keep every instance feasible: the first-step rate limit (|acc_1 - acc_prev| <= 0.15) must leave
    # room for v_2 = v0 + dt*acc_1 >= 0 (a standing car whose last command was hard braking has no
    # feasible input in the reference NLP; that case is exercised by an explicit edge-case test)
    acc_prev = np.maximum(acc_prev, -v0 / dt - 0.10)

    s = v_t[:, None] * dt * np.arange(N + 1)[None, :]
    ks = kappa[:, None] * s
    small = np.abs(kappa)[:, None] < 1e-9
    ksafe = np.where(small, 1.0, kappa[:, None])
    xr = np.where(small, s * np.cos(th0)[:, None], (np.sin(th0[:, None] + ks) - np.sin(th0)[:, None]) / ksafe)
    yr = np.where(small, s * np.sin(th0)[:, None], -(np.cos(th0[:, None] + ks) - np.cos(th0)[:, None]) / ksafe)
    pr = th0[:, None] + ks
    ref = np.stack([xr, yr, pr], axis=-1)
    z0 = np.stack([-e_y * np.sin(th0), e_y * np.cos(th0), th0 + e_psi, v0], axis=-1)
    u_prev = np.stack([acc_prev, df_prev], axis=-1)
    return dict(z0=z0.astype(dtype), ref=ref.astype(dtype), v_target=v_t.astype(dtype),
                u_prev=u_prev.astype(dtype), hard=hard)


def straight_line_case(N, v0=0.0, y0=0.0, psi0=0.0, v_ref=15.0, dt=0.2):
    """The module-load problem of MKZMPCPathFollower.jl:36-39,110-113 (BASELINE config 1 at N=8)."""
    ref = np.zeros((1, N + 1, 3))
    ref[0, :, 0] = v_ref * dt * np.arange(N + 1)
    z0 = np.array([[0.0, y0, psi0, v0]])
    return dict(z0=z0, ref=ref, v_target=np.array([v_ref]), u_prev=np.zeros((1, 2)), hard=np.array([False]))
