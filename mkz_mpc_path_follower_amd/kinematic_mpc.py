"""Single-problem mirror of the reference's Julia module scripts/mpc_utils/MKZMPCPathFollower.jl.

Same six functions, same argument orders (including the reference's quirks: update_current_input
takes steer first (:151), solve_model returns (acc, d_f, status) (:182), get_solver_results returns
v before psi and d_f before acc (:188-207)).  The solve runs on the MI355X through the C ABI's
host entry point with B = 1; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KMPC_F64, STATUS_NAMES, Config


class KinematicMPC:
    dt_control = 0.10  # MKZMPCPathFollower.jl:28
    dt = 0.20          # :33

    def __init__(self, N=8, device=0, **options):
        self.N = int(N)  # :34
        self._lib = _lib.load()
        self._cfg = Config()
        _lib.check(self._lib.kmpc_config_default(C.byref(self._cfg), self.N, KMPC_F64))
        for k, v in options.items():
            if not hasattr(self._cfg, k):
                raise TypeError("unknown option %r" % k)
            setattr(self._cfg, k, v)
        self.dt, self.dt_control = self._cfg.dt, self._cfg.dt_control
        h = C.c_void_p()
        _lib.check(self._lib.kmpc_create(C.byref(self._cfg), int(device), C.byref(h)))
        self._h = h
        n1 = self.N + 1
        # module-load defaults :36-39, :91-94, :110-113, :75, :82
        v_ref = 15.0
        self._z0 = np.zeros(4)
        self._ref = np.zeros((n1, 3))
        self._ref[:, 0] = v_ref * self.dt * np.arange(n1)
        self._vt = np.array([v_ref])
        self._up = np.zeros(2)  # (acc_current, d_f_current)
        self._U = np.zeros((self.N, 2))
        self._X = np.zeros((n1, 4))
        self._have_solution = False
        self.status = None
        self.cost = None
        self.iters = None
        # the Julia module solves once at load time (:125-128); so does this mirror
        self.solve_model()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kmpc_destroy(self._h)
            self._h = None

    __del__ = close

    # MKZMPCPathFollower.jl:132-138
    def update_init_cond(self, x, y, psi, vel):
        self._z0[:] = (x, y, psi, vel)

    # :142-147
    def update_reference(self, x_ref, y_ref, psi_ref, v_des):
        n1 = self.N + 1
        for a in (x_ref, y_ref, psi_ref):
            if len(a) != n1:
                raise ValueError("reference arrays must have length N+1 = %d" % n1)
        self._ref[:, 0] = x_ref
        self._ref[:, 1] = y_ref
        self._ref[:, 2] = psi_ref
        self._vt[0] = v_des

    # :151-154 -- NOTE steer first, as in the reference
    def update_current_input(self, c_swa, c_acc):
        self._up[:] = (c_acc, c_swa)

    # :158-169
    def update_cost(self, cx, cy, cp, cv, cda, cdd, ca, cd):
        w = (C.c_double * 8)(cx, cy, cp, cv, cda, cdd, ca, cd)
        _lib.check(self._lib.kmpc_set_cost(self._h, w), self._h)

    # :173-183
    def solve_model(self):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        u0 = np.zeros((1, 2))
        st = np.zeros(1, np.int32)
        cost = np.zeros(1)
        viol = np.zeros(1)
        iters = np.zeros(1, np.int32)
        warm = 1 if self._have_solution else 0  # JuMP re-solves from the previous primal values (Q9)
        _lib.check(self._lib.kmpc_solve_batch_host(self._h, 1, p(self._z0), p(self._ref), p(self._vt), p(self._up),
                                                   p(self._U), warm, p(u0), p(st), p(cost), p(viol), p(iters),
                                                   None, p(self._X)), self._h)
        self._have_solution = True
        self.status = STATUS_NAMES[int(st[0])]
        self.cost, self.viol, self.iters = float(cost[0]), float(viol[0]), int(iters[0])
        return float(u0[0, 0]), float(u0[0, 1]), self.status

    # :188-207 -- (x, y, v, psi, x_ref, y_ref, psi_ref, d_f_opt, acc_opt)
    def get_solver_results(self):
        X, U, R = self._X, self._U, self._ref
        return (X[:, 0].copy(), X[:, 1].copy(), X[:, 3].copy(), X[:, 2].copy(),
                R[:, 0].copy(), R[:, 1].copy(), R[:, 2].copy(), U[:, 1].copy(), U[:, 0].copy())
