"""Frenet-frame model variant (scripts/mpc_utils/MKZMPCPathFollowerFrenet.jl) as a second dynamics
functor.  CPU part: the oracle's restatement of the model (finite-difference check of its Jacobians, known answers, an
independent scipy cross-solve); GPU part: the HIP functor against the oracle.  Parity unpinned (no reference tests exist)."""
import numpy as np
import pytest

FRENET_W = (0.0, 9.0, 10.0, 0.5, 100.0, 1000.0, 0.0, 0.0)  # MKZMPCPathFollowerFrenet.jl:51-59 in the 8-slot layout


def _cases(B, N, seed=3):
    """(s0, e_y0, e_psi0, v0); cubic curvature polynomials K(s) = a + b x + c x^2 + d x^3, x = s / 60 m, that stay within road-like
    curvature (|K| <~ 0.09 1/m, the range of the recorded paths) over any horizon used here; target speeds; previous inputs"""
    rng = np.random.default_rng(seed)
    z0 = np.stack([rng.uniform(0, 5, B), rng.normal(0, 0.4, B), rng.normal(0, 0.08, B), rng.uniform(2, 12, B)], 1)
    a, b, c, d = rng.uniform(-0.04, 0.04, B), rng.normal(0, 0.015, B), rng.normal(0, 0.015, B), rng.normal(0, 0.015, B)
    kp = np.stack([d / 60.0 ** 3, c / 60.0 ** 2, b / 60.0, a], 1)  # highest degree first
    vt = np.clip(z0[:, 3] + rng.normal(0, 1.0, B), 1.0, 15.0)
    up = np.stack([rng.uniform(-0.4, 0.4, B), rng.uniform(-0.05, 0.05, B)], 1)
    return z0, kp, vt, up


def test_frenet_jacobians_match_finite_differences(oracle):
    O = oracle
    import ctypes as C
    p = O.params(8, model=1)
    rng = np.random.default_rng(0)
    for _ in range(20):
        z = np.array([rng.uniform(0, 30), rng.normal(0, 0.5), rng.normal(0, 0.2), rng.uniform(1, 15)])
        u = np.array([rng.uniform(-1, 1), rng.uniform(-0.4, 0.4)])
        kp = np.array([rng.normal(0, 1e-5), rng.normal(0, 1e-4), rng.normal(0, 1e-3), rng.uniform(-0.05, 0.05)])
        A, Bm = np.empty(16), np.empty(8)
        O.lib().kmpc_stage_jac_m(C.byref(p), O._p(kp), O._p(z), O._p(u), O._p(A), O._p(Bm))
        A, Bm = A.reshape(4, 4), Bm.reshape(4, 2)

        def f(zz, uu):
            U = np.zeros((8, 2)); U[0] = uu
            return O.rollout(p, zz, U, kp)[1]
        for j in range(4):
            e = np.zeros(4); e[j] = 1e-6
            assert np.allclose((f(z + e, u) - f(z - e, u)) / 2e-6, A[:, j], atol=2e-7), j
        for j in range(2):
            e = np.zeros(2); e[j] = 1e-6
            assert np.allclose((f(z, u + e) - f(z, u - e)) / 2e-6, Bm[:, j], atol=2e-7), j


def test_frenet_known_answers(oracle):
    """on the path at the target speed the optimum is U = 0, J = 0, whatever the curvature only if the feed-forward steering
    is free -- so use K = 0; and the dynamics reduce to the reference's Euler step (:114-123) for a hand-computed step"""
    O = oracle
    N = 8
    p = O.params(N, model=1)
    r = O.solve_condensed(p, O.problem_frenet(p, [3.0, 0.0, 0.0, 7.0], [0, 0, 0, 0.0], 7.0))
    assert r["status"] == 0 and abs(r["cost"]) < 1e-12 and np.abs(r["U"]).max() < 1e-7
    z, u, kp = np.array([2.0, 0.3, 0.1, 6.0]), np.array([0.5, 0.2]), np.array([1e-4, -2e-3, 0.01, 0.03])
    U = np.zeros((N, 2)); U[0] = u
    z1 = O.rollout(p, z, U, kp)[1]
    K = kp[0] * 8 + kp[1] * 4 + kp[2] * 2 + kp[3]
    beta = np.arctan(1.742 / (1.108 + 1.742) * np.tan(0.2))
    dsdt = 6.0 * np.cos(0.1 + beta) / (1 - 0.3 * K)
    exp = [2.0 + 0.2 * dsdt, 0.3 + 0.2 * 6.0 * np.sin(0.1 + beta), 0.1 + 0.2 * (6.0 / 1.742 * np.sin(beta) - dsdt * K), 6.0 + 0.2 * 0.5]
    assert np.allclose(z1, exp, rtol=0, atol=1e-14)


def test_frenet_oracle_vs_scipy(oracle):
    """independent cross-solve of the same NLP (states eliminated) with scipy trust-constr and the analytic gradient"""
    from scipy.optimize import minimize, LinearConstraint
    O = oracle
    N = 8
    p = O.params(N, model=1)
    z0, kp, vt, up = _cases(4, N, seed=9)
    for b in range(4):
        q = O.problem_frenet(p, z0[b], kp[b], vt[b], up[b])
        r = O.solve_condensed(p, q)
        assert r["status"] == 0 and r["viol"] <= 1e-8 + 1e-12
        A, bb = O.ineq(p, q)
        res = minimize(lambda U: O.cost(p, q, U), r["U"].ravel() * 0.9, jac=lambda U: O.grad(p, q, U), method="trust-constr",
                       constraints=[LinearConstraint(A, -np.inf, bb)], options=dict(gtol=1e-10, xtol=1e-12, maxiter=3000))
        assert abs(res.fun - r["cost"]) <= 2e-6 * max(1.0, abs(r["cost"])), (res.fun, r["cost"])


def test_curvature_polynomial_fit_recovers_a_known_curvature():
    """nav_msgs_path_frenet.py:44-86 restated: waypoints on a clothoid-like path with K(s) = 0.01 + 0.002 s are fitted back to a
    polynomial whose curvature matches over the fitted range (the double cubic fit is approximate by construction)"""
    from mkz_mpc_path_follower_amd.kinematic_mpc_frenet import get_reference_frenet
    s = np.arange(0.0, 30.0, 0.05)
    K = 0.01 + 0.002 * s
    psi = 0.3 + np.concatenate([[0.0], np.cumsum(0.5 * (K[1:] + K[:-1]) * np.diff(s))])
    x = np.concatenate([[0.0], np.cumsum(np.cos(psi[:-1]) * np.diff(s))])
    y = np.concatenate([[0.0], np.cumsum(np.sin(psi[:-1]) * np.diff(s))])
    Kc, psi0, xi, yi = get_reference_frenet(dict(x=x[::20], y=y[::20], s=s[::20]))
    assert abs(psi0 - 0.3) < 0.05
    sm = np.linspace(3.0, 25.0, 12)
    assert np.abs(np.polyval(Kc, sm) - (0.01 + 0.002 * sm)).max() < 0.01
    assert np.hypot(xi - np.interp(np.arange(0.0, s[::20][-1], 0.25), s, x), yi - np.interp(np.arange(0.0, s[::20][-1], 0.25), s, y)).max() < 0.2


# ---------------------------------------------------------------- GPU: the HIP functor against the oracle
@pytest.mark.gpu
@pytest.mark.parametrize("N", [8, 12, 16, 20, 24])
def test_frenet_compile_time_and_generic_kernels_agree(N):
    """the Frenet functor lives in two kernels -- the compile-time-horizon one (kernel_variant 0, N = 8 ... 28) and the generic one
    (kernel_variant 1, N <= 24): same statuses, costs to 1e-7 relative, iteration counts within rounding effects"""
    import torch
    from mkz_mpc_path_follower_amd import BatchMPC
    z0, kp, vt, up = _cases(400, N, seed=5)
    out = []
    for v in (0, 1):
        o = BatchMPC(N=N, dtype=torch.float64, model=1, kernel_variant=v).solve_frenet(z0, kp, vt, up, want_X=True)
        torch.cuda.synchronize()
        out.append({k: t.cpu().numpy() for k, t in o.items()})
    a, b = out
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    rel = np.abs(a["cost"] - b["cost"]) / np.maximum(1.0, np.abs(b["cost"]))
    assert rel.max() <= 1e-7 and abs(a["iters"].mean() - b["iters"].mean()) < 0.5
    assert np.abs(a["X"] - b["X"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("N,B", [(8, 1500), (20, 600), (28, 300)])
def test_frenet_kernel_matches_oracle(oracle, N, B):
    """fp64 tolerances of SURVEY.md 8(c): |J - J_oracle| <= 1e-6 max(1,|J|), violation <= 1e-8, first input within 1e-6; every problem
    of the draw Optimal in both solvers (exact Hessian of the Frenet functor)."""
    import torch
    from mkz_mpc_path_follower_amd import BatchMPC
    O = oracle
    z0, kp, vt, up = _cases(B, N, seed=21)
    s = BatchMPC(N=N, dtype=torch.float64, model=1)
    o = s.solve_frenet(z0, kp, vt, up, want_U=True, want_X=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in o.items()}
    p = O.params(N, model=1)
    r = O.solve_condensed_batch(p, z0, kp, vt, up, nthreads=8, want_X=True)
    ok = (g["status"] == 0) & (r["status"] == 0)
    assert ok.all(), (np.bincount(g["status"]), np.bincount(r["status"]))
    rel = np.abs(g["cost"] - r["cost"]) / np.maximum(1.0, np.abs(r["cost"]))
    assert rel[ok].max() <= 1e-6
    assert g["viol"][ok].max() <= 1e-8 + 1e-12
    assert np.abs(g["u0"] - r["U"].reshape(B, N, 2)[:, 0, :])[ok].max() <= 1e-6
    assert np.abs(g["X"] - r["X"])[ok].max() <= 1e-5   # predicted states integrate input differences (<= 1e-6 each) over up to 28 stages
    assert abs(g["iters"][ok].mean() - r["iters"][ok].mean()) < 1.0


@pytest.mark.gpu
def test_frenet_module_api(oracle):
    """the six module functions with the reference's argument orders; the module-load solve (K = 0, on the path, v0 = 0 -> accelerate);
    results against the oracle"""
    from mkz_mpc_path_follower_amd.kinematic_mpc_frenet import KinematicMPCFrenet
    O = oracle
    m = KinematicMPCFrenet(N=8)
    assert m.status == "Optimal"                      # load-time solve: z0 = 0, v_target 15 -> acc_1 = a_dmax * dt_control
    s_, ey_, v_, epsi_, K_, path_, df_, acc_ = m.get_solver_results()
    assert abs(acc_[0] - 0.15) < 1e-6 and np.abs(df_).max() < 1e-7 and np.abs(ey_).max() < 1e-9
    m.update_init_cond(1.5, 0.4, -0.05, 6.0)
    m.update_reference({"x": [0.0], "y": [0.0]}, [1e-5, -2e-4, 1e-3, 0.03], 7.0)
    m.update_current_input(0.01, 0.2)                 # steer first
    m.update_cost(9.0, 10.0, 0.5, 100.0, 1000.0, 0.0, 0.0)
    a, d, st = m.solve_model()
    assert st == "Optimal"
    p = O.params(8, model=1)
    r = O.solve_condensed(p, O.problem_frenet(p, [1.5, 0.4, -0.05, 6.0], [1e-5, -2e-4, 1e-3, 0.03], 7.0, (0.2, 0.01)))
    assert abs(m.cost - r["cost"]) <= 1e-6 * max(1.0, r["cost"]) and abs(a - r["U"][0, 0]) < 1e-6 and abs(d - r["U"][0, 1]) < 1e-6
    res = m.get_solver_results()
    assert np.allclose(res[0], r["X"][:, 0], atol=1e-6) and np.allclose(res[2], r["X"][:, 3], atol=1e-6)   # s, then v (v before epsi)
    assert np.allclose(res[3], r["X"][:, 2], atol=1e-6) and res[5] == {"x": [0.0], "y": [0.0]}
    a2, d2, st2 = m.solve_model()                     # warm re-solve of the same problem
    assert st2 == "Optimal" and abs(a2 - a) < 1e-6
