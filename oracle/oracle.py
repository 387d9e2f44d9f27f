"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.

ctypes front-end to the C part of the CPU checker in this directory (kmpc_nlp.c,
kmpc_condensed.c); the full-space Ipopt-style solver is oracle/ipopt_like.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module; the shipped package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class Params(C.Structure):
    """struct kmpc_params (kmpc_nlp.h); defaults = MKZMPCPathFollower.jl:28-59."""
    _fields_ = [("N", C.c_int), ("dt", C.c_double), ("dt_control", C.c_double),
                ("L_a", C.c_double), ("L_b", C.c_double),
                ("steer_max", C.c_double), ("steer_dmax", C.c_double),
                ("a_max", C.c_double), ("a_dmax", C.c_double),
                ("v_min", C.c_double), ("v_max", C.c_double), ("C", C.c_double * 8), ("model", C.c_int)]


class Problem(C.Structure):
    _fields_ = [("z0", C.c_double * 4), ("ref", c_double_p), ("v_target", C.c_double),
                ("u_prev", C.c_double * 2), ("k_poly", C.c_double * 4)]


class Opts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol", C.c_double), ("hessian", C.c_int),
                ("mu_init", C.c_double), ("bound_relax", C.c_double), ("warm", C.c_int),
                ("warm_push", C.c_double), ("warm_mu", C.c_double), ("max_ls", C.c_int), ("mu_strategy", C.c_int), ("indef_strategy", C.c_int),
                ("start", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("n_refactor", C.c_int), ("n_ls", C.c_int), ("n_solves", C.c_int),
                ("cost", C.c_double), ("viol", C.c_double), ("kkt", C.c_double), ("mu", C.c_double)]


def build(force=False):
    """Compile libkmpc_oracle.so with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "libkmpc_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkmpc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.kmpc_params_default.argtypes = [C.POINTER(Params), C.c_int]
        L.kmpc_params_default_frenet.argtypes = [C.POINTER(Params), C.c_int]
        L.kmpc_rollout.argtypes = [C.POINTER(Params), c_double_p, c_double_p, c_double_p]
        L.kmpc_rollout_m.argtypes = [C.POINTER(Params), c_double_p, c_double_p, c_double_p, c_double_p]
        L.kmpc_stage_jac_m.argtypes = [C.POINTER(Params), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
        L.kmpc_cost.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_double_p, c_double_p]
        L.kmpc_cost.restype = C.c_double
        L.kmpc_grad.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_double_p, c_double_p, c_double_p]
        L.kmpc_ineq.argtypes = [C.POINTER(Params), C.POINTER(Problem), C.c_double, c_double_p, c_double_p]
        L.kmpc_max_violation.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_double_p]
        L.kmpc_max_violation.restype = C.c_double
        L.kmpc_certify.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_double_p, c_double_p, c_double_p]
        L.kmpc_opts_default.argtypes = [C.POINTER(Opts)]
        L.kmpc_condensed_solve.argtypes = [C.POINTER(Params), C.POINTER(Problem), C.POINTER(Opts),
                                           c_double_p, c_double_p, c_double_p, C.POINTER(Result)]
        L.kmpc_condensed_solve_batch.argtypes = [C.POINTER(Params), C.POINTER(Opts), C.c_int,
                                                 c_double_p, c_double_p, c_double_p, c_double_p,
                                                 c_double_p, c_double_p, c_int_p, c_double_p, c_double_p,
                                                 c_int_p, C.c_int]
        L.kmpc_condensed_solve_batch_stats.argtypes = [C.POINTER(Params), C.POINTER(Opts), C.c_int,
                                                       c_double_p, c_double_p, c_double_p, c_double_p,
                                                       c_double_p, c_double_p, c_int_p, c_double_p, c_double_p,
                                                       c_int_p, c_int_p, c_int_p, C.c_int]
        L.kmpc_condense.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_double_p, C.c_int,
                                    c_double_p, c_double_p, c_double_p]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(c_double_p)


def params(N=8, weights=None, model=0, **kw):
    """model 0: MKZMPCPathFollower.jl (Cartesian), 1: MKZMPCPathFollowerFrenet.jl (weights then are the full 8-vector
    (0, C_ey, C_epsi, C_ev, C_dacc, C_ddf, C_acc, C_df))"""
    p = Params()
    (lib().kmpc_params_default_frenet if model == 1 else lib().kmpc_params_default)(C.byref(p), int(N))
    if weights is not None:
        for i, w in enumerate(weights):
            p.C[i] = float(w)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def opts(**kw):
    o = Opts()
    lib().kmpc_opts_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class _Prob:
    """keeps the numpy reference array alive next to the ctypes struct"""

    def __init__(self, p, z0, ref, v_target, u_prev, k_poly=None):
        N = p.N
        if ref is None:  # Frenet model: the cost has zero references
            ref = np.zeros((N + 1, 3))
        self.ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(N + 1, 3)
        self.c = Problem()
        for i in range(4):
            self.c.z0[i] = float(z0[i])
        self.c.ref = _p(self.ref)
        self.c.v_target = float(v_target)
        self.c.u_prev[0] = float(u_prev[0])
        self.c.u_prev[1] = float(u_prev[1])
        for i in range(4):
            self.c.k_poly[i] = 0.0 if k_poly is None else float(k_poly[i])


def problem(p, z0, ref, v_target, u_prev=(0.0, 0.0), k_poly=None):
    return _Prob(p, z0, ref, v_target, u_prev, k_poly)


def problem_frenet(p, z0, k_poly, v_target, u_prev=(0.0, 0.0)):
    """z0 = (s, e_y, e_psi, v); k_poly highest degree first (MKZMPCPathFollowerFrenet.jl:132-147)"""
    return _Prob(p, z0, None, v_target, u_prev, k_poly)


def rollout(p, z0, U, k_poly=None):
    N = p.N
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(2 * N)
    z0 = np.ascontiguousarray(z0, dtype=np.float64)
    X = np.empty((N + 1, 4))
    kp = np.zeros(4) if k_poly is None else np.ascontiguousarray(k_poly, dtype=np.float64)
    lib().kmpc_rollout_m(C.byref(p), _p(kp), _p(z0), _p(U), _p(X))
    return X


def cost(p, q, U):
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(2 * p.N)
    X = rollout(p, np.array(q.c.z0[:]), U, np.array(q.c.k_poly[:]))
    return lib().kmpc_cost(C.byref(p), C.byref(q.c), _p(U), _p(X))


def grad(p, q, U):
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(2 * p.N)
    X = rollout(p, np.array(q.c.z0[:]), U, np.array(q.c.k_poly[:]))
    g = np.empty(2 * p.N)
    lib().kmpc_grad(C.byref(p), C.byref(q.c), _p(U), _p(X), _p(g))
    return g


def ineq(p, q, relax=0.0):
    n, m = 2 * p.N, 10 * p.N - 4
    A = np.empty((m, n))
    b = np.empty(m)
    lib().kmpc_ineq(C.byref(p), C.byref(q.c), float(relax), _p(A), _p(b))
    return A, b


def certify(p, q, U, lam):
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(2 * p.N)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    out = np.empty(5)
    lib().kmpc_certify(C.byref(p), C.byref(q.c), _p(U), _p(lam), _p(out))
    return dict(stationarity=out[0], violation=out[1], complementarity=out[2], lam_min=out[3], cost=out[4])


def condense(p, q, U, hessian=0):
    n = 2 * p.N
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(n)
    H = np.empty((n, n))
    g = np.empty(n)
    J = C.c_double()
    lib().kmpc_condense(C.byref(p), C.byref(q.c), _p(U), int(hessian), _p(H), _p(g), C.byref(J))
    return H, g, J.value


def solve_condensed(p, q, o=None, U0=None):
    """returns dict(U[N,2], X[N+1,4], lam[m], status, iters, cost, viol, kkt, ...)"""
    o = o or opts()
    N = p.N
    U = np.zeros(2 * N) if U0 is None else np.ascontiguousarray(U0, dtype=np.float64).reshape(2 * N).copy()
    X = np.empty((N + 1, 4))
    lam = np.empty(10 * N - 4)
    r = Result()
    lib().kmpc_condensed_solve(C.byref(p), C.byref(q.c), C.byref(o), _p(U), _p(X), _p(lam), C.byref(r))
    return dict(U=U.reshape(N, 2), X=X, lam=lam, status=r.status, iters=r.iters, n_refactor=r.n_refactor,
                n_ls=r.n_ls, n_solves=r.n_solves, cost=r.cost, viol=r.viol, kkt=r.kkt, mu=r.mu)


def solve_condensed_batch(p, z0, ref, v_target, u_prev, o=None, U0=None, nthreads=1, want_X=False):
    o = o or opts()
    N = p.N
    B = z0.shape[0]
    z0 = np.ascontiguousarray(z0, dtype=np.float64)
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    v_target = np.ascontiguousarray(v_target, dtype=np.float64)
    u_prev = np.ascontiguousarray(u_prev, dtype=np.float64)
    U = np.zeros((B, 2 * N)) if U0 is None else np.ascontiguousarray(U0, dtype=np.float64).reshape(B, 2 * N).copy()
    X = np.empty((B, N + 1, 4)) if want_X else None
    status = np.empty(B, dtype=np.int32)
    costv = np.empty(B)
    viol = np.empty(B)
    iters = np.empty(B, dtype=np.int32)
    nref, nls = (np.empty(B, dtype=np.int32) for _ in range(2))
    ip = lambda a: a.ctypes.data_as(c_int_p)
    lib().kmpc_condensed_solve_batch_stats(C.byref(p), C.byref(o), B, _p(z0), _p(ref), _p(v_target), _p(u_prev),
                                           _p(U), _p(X) if want_X else None, ip(status), _p(costv), _p(viol), ip(iters),
                                           ip(nref), ip(nls), int(nthreads))
    return dict(U=U.reshape(B, N, 2), X=X, status=status, cost=costv, viol=viol, iters=iters, n_refactor=nref, n_ls=nls)
