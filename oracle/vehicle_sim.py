"""oracle/vehicle_sim.py -- TEST INFRASTRUCTURE.  PARITY UNPINNED (the reference has no tests or recorded
trajectories for its simulator; Python 2 + rospy, cannot be imported here -- restated from the text).

numpy restatement, vectorised over B vehicles, of scripts/vehicle_simulator.py:
  _update_vehicle_model (:58-107): dynamic bicycle with a linear tyre model, forward Euler with
      disc_steps = 10 sub-steps of dt_model / 10 = 1 ms, heading wrapped to [-pi, pi) every sub-step,
      vx floored at 0, lateral states zeroed when the car stands;
  _update_low_level_control (:109-113): first-order lag (gain 5 1/s) of acc and df towards the commands,
      applied after every sub-step.
Quirk kept: the rear slip angle uses lf where lr is expected (:77); flagged in SURVEY.md section 8(f2).
State layout [B, 8]: X, Y, psi, vx, vy, wz, acc, df  (the attributes of VehicleSimulator, :18-34).
"""
import numpy as np

LF, LR, MASS, IZ = 1.152, 1.693, 1840.0, 3477.0      # :61-65
C_ALPHA_F, C_ALPHA_R = 4.0703e4, 6.4495e4            # :66-67
DT_MODEL, DISC_STEPS, KP = 0.01, 10, 5.0             # :24, :58, :112-113
X0, Y0, PSI0 = -300.0, -450.0, 1.0                   # :28-30 (rosparam defaults)


def initial_state(B=1, X=X0, Y=Y0, psi=PSI0):
    s = np.zeros((B, 8))
    s[:, 0], s[:, 1], s[:, 2] = X, Y, psi
    return s


def update_vehicle_model(state, cmd, n_updates=1, disc_steps=DISC_STEPS):
    """n_updates calls of _update_vehicle_model (each = disc_steps Euler sub-steps + lag); cmd [B,2] = (acc_des, df_des)"""
    s = np.array(state, dtype=np.float64, copy=True)
    X, Y, psi, vx, vy, wz, acc, df = (s[:, i].copy() for i in range(8))
    acc_des, df_des = np.asarray(cmd, dtype=np.float64)[:, 0], np.asarray(cmd, dtype=np.float64)[:, 1]
    deltaT = DT_MODEL / disc_steps                                                  # :69
    for _ in range(n_updates * disc_steps):
        moving = np.fabs(vx) > 1e-6                                                 # :75
        alpha_f = np.where(moving, df - np.arctan2(vy + LF * wz, vx), 0.0)          # :76
        alpha_r = np.where(moving, -np.arctan2(vy - LF * wz, vx), 0.0)              # :77 (lf, as in the reference)
        Fyf = C_ALPHA_F * alpha_f                                                   # :80
        Fyr = C_ALPHA_R * alpha_r                                                   # :81
        # :84 `acc - 1/m*Fyf*np.sin(self.df) + ...` with m = 1840 (int) under Python 2: 1/m is integer division = 0, the term vanishes
        vx_n = np.maximum(0.0, vx + deltaT * (acc - (1 // int(MASS)) * Fyf * np.sin(df) + wz * vy))  # :84
        fwd = vx_n > 1e-6                                                           # :87
        vy_n = np.where(fwd, vy + deltaT * (1.0 / MASS * (Fyf * np.cos(df) + Fyr) - wz * vx), 0.0)      # :88,91
        wz_n = np.where(fwd, wz + deltaT * (1.0 / IZ * (LF * Fyf * np.cos(df) - LR * Fyr)), 0.0)        # :89,92
        psi_n = psi + deltaT * wz                                                   # :94
        X_n = X + deltaT * (vx * np.cos(psi) - vy * np.sin(psi))                    # :95
        Y_n = Y + deltaT * (vx * np.sin(psi) + vy * np.cos(psi))                    # :96
        X, Y = X_n, Y_n
        psi = (psi_n + np.pi) % (2.0 * np.pi) - np.pi                               # :101
        vx, vy, wz = vx_n, vy_n, wz_n
        acc = KP * (acc_des - acc) * deltaT + acc                                   # :112
        df = KP * (df_des - df) * deltaT + df                                       # :113
    return np.stack([X, Y, psi, vx, vy, wz, acc, df], axis=1)
