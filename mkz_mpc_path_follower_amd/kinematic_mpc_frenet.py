"""Single-problem mirror of the reference's Julia module scripts/mpc_utils/MKZMPCPathFollowerFrenet.jl and of the
curvature-polynomial fit that feeds it (scripts/sim_path_utils/nav_msgs_path_frenet.py:44-86).

Same six functions, same argument orders: update_init_cond(s, ey, epsi, vel) (:132-138), update_reference(path, k_coeffs,
v_des) (:142-147), update_current_input(c_swa, c_acc) -- steer first (:151-154), update_cost(cey, cep, cev, cda, cdd, ca, cd)
(:158-169), solve_model() -> (acc, d_f, status) (:173-183), get_solver_results() -> (s, ey, v, epsi, K, path_ref, d_f, acc)
(:188-207, v before epsi, d_f before acc).  The solve runs on the MI355X (kmpc_solve_batch_frenet, B = 1); no CPU path.
"""
import math

import numpy as np
import torch

from ._lib import STATUS_NAMES
from .solver import BatchMPC


# ---- curvature-polynomial fit of nav_msgs_path_frenet.py:44-86 (numpy's polynomial helpers, coefficients highest degree first) --------
def fit_XY_s(x_arr, y_arr, s_arr):
    """:62-73  X(s), Y(s) cubics through the waypoints resampled every 0.5 m"""
    s_interp = np.arange(s_arr[0], s_arr[-1], 0.5)
    x_interp = np.interp(s_interp, s_arr, x_arr)
    y_interp = np.interp(s_interp, s_arr, y_arr)
    return np.polyfit(s_interp, x_interp, 3), np.polyfit(s_interp, y_interp, 3)


def compute_curvature_poly(s_interp, x_coeffs, y_coeffs):
    """:44-59  K = (x' y'' - y' x'') / (x'^2 + y'^2) on the fitted cubics, then a cubic fit of K(s), highest degree first"""
    s_interp = np.asarray(s_interp, dtype=np.float64)
    d1x, d1y = np.polyder(x_coeffs), np.polyder(y_coeffs)
    dx, dy = np.polyval(d1x, s_interp), np.polyval(d1y, s_interp)
    ddx, ddy = np.polyval(np.polyder(d1x), s_interp), np.polyval(np.polyder(d1y), s_interp)
    K_meas = (dx * ddy - dy * ddx) / (dx ** 2 + dy ** 2)
    return np.polyfit(s_interp, K_meas, 3)


def get_reference_frenet(path):
    """:76-86  path = dict(x, y, s) -> (K_coeffs, psi_start, x_interp, y_interp)"""
    x_coeffs, y_coeffs = fit_XY_s(path["x"], path["y"], path["s"])
    s_interp = np.arange(0.0, path["s"][-1], 0.25)
    x_interp = np.polyval(x_coeffs, s_interp)
    y_interp = np.polyval(y_coeffs, s_interp)
    K_coeffs = compute_curvature_poly(s_interp, x_coeffs, y_coeffs)
    psi_start = math.atan2(y_coeffs[2], x_coeffs[2])   # slopes of the two cubics at s = 0
    return K_coeffs, psi_start, x_interp, y_interp


class KinematicMPCFrenet:
    dt_control = 0.10  # MKZMPCPathFollowerFrenet.jl:28
    dt = 0.20          # :33

    def __init__(self, N=8, device=0, **options):
        self.N = int(N)  # :34
        self._mpc = BatchMPC(N=self.N, dtype=torch.float64, device=device, model=1, **options)
        self._z0 = np.zeros((1, 4))                 # s0, ey0, epsi0, v0 (:107-110)
        self._kp = np.zeros((1, 4))                 # k_coeff_ref (:38)
        self._vt = np.array([15.0])                 # v_ref (:37)
        self._up = np.zeros((1, 2))                 # (acc_current, d_f_current)
        self._U = np.zeros((self.N, 2))
        self._X = np.zeros((self.N + 1, 4))
        self._warm = None
        self.path_ref = {}                          # :36
        self.status = self.cost = self.iters = None
        self.solve_model()                          # the module solves once at load time (:125-128)

    def close(self):
        self._mpc.close()

    def update_init_cond(self, s, ey, epsi, vel):   # :132-138
        self._z0[0, :] = (s, ey, epsi, vel)

    def update_reference(self, path, k_coeffs, v_des):  # :142-147
        k = np.asarray(k_coeffs, dtype=np.float64).ravel()
        if k.shape != (4,):
            raise ValueError("k_coeffs must have 4 entries, highest degree first")
        self.path_ref = path
        self._kp[0, :] = k
        self._vt[0] = float(v_des)

    def update_current_input(self, c_swa, c_acc):   # :151-154 (steer first)
        self._up[0, :] = (c_acc, c_swa)

    def update_cost(self, cey, cep, cev, cda, cdd, ca, cd):  # :158-169
        self._mpc.update_cost(0.0, cey, cep, cev, cda, cdd, ca, cd)

    def solve_model(self):                          # :173-183
        warm = self._warm is not None
        wu = self._warm if warm else torch.zeros((1, self.N, 2), dtype=torch.float64, device=self._mpc.device)
        o = self._mpc.solve_frenet(self._z0, self._kp, self._vt, self._up, warm_U=wu, warm=warm, want_U=True, want_X=True)
        torch.cuda.synchronize(self._mpc.device)
        self._warm = o["warm_U"]                    # JuMP re-solves from the previous primal values (Q9)
        self._U = o["U"][0].cpu().numpy()
        self._X = o["X"][0].cpu().numpy()
        self.status = STATUS_NAMES[int(o["status"][0].item())]
        self.cost, self.iters = float(o["cost"][0].item()), int(o["iters"][0].item())
        return float(self._U[0, 0]), float(self._U[0, 1]), self.status

    def get_solver_results(self):                   # :188-207
        X, U = self._X, self._U
        return (X[:, 0].copy(), X[:, 1].copy(), X[:, 3].copy(), X[:, 2].copy(), self._kp[0].copy(), self.path_ref,
                U[:, 1].copy(), U[:, 0].copy())
