"""oracle/ipopt_like.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.

Restatement of the reference's solve step: scripts/mpc_utils/MKZMPCPathFollower.jl:127,176
call `solve(mdl)` on a JuMP model whose solver is `IpoptSolver(print_level=0,
max_cpu_time=dt_control)` (:29).  Neither JuMP nor Ipopt is in /root/reference (third-party,
versions unpinned: README.md:15-24 says only "Julia 0.4.7", Pkg.add("JuMP"), Pkg.add("Ipopt")
-- i.e. JuMP <= 0.18 / Ipopt 3.12-era), and none of it can run in the build container, so this
file restates

  * the JuMP model in its FULL-SPACE form -- states x,y,v,psi[1..N+1] and inputs acc,d_f[1..N]
    are all variables (:65-72), dynamics and initial conditions are equality constraints
    (:110-123), rate limits are two-sided general constraints (:75-86), bounds are variable
    bounds, every primal starts at 0 (start=0.0, Q9); and
  * Ipopt's published algorithm (A. Waechter, L. T. Biegler, "On the implementation of an
    interior-point filter line-search algorithm for large-scale nonlinear programming",
    Math. Program. 106 (2006) 25-57) with Ipopt's default option values: slack reformulation of
    the general inequalities, bound_relax_factor 1e-8, bound_push = bound_frac = 1e-2,
    gradient-based objective scaling (max gradient 100), z = 1 / least-squares lambda start,
    monotone mu (mu_init 0.1, kappa_mu 0.2, theta_mu 1.5, kappa_eps 10), fraction-to-the-
    boundary tau = max(0.99, 1-mu), inertia correction of the KKT matrix, filter line search
    with second-order correction, kappa_Sigma = 1e10 dual reset, tol = 1e-8 on the scaled
    optimality error.  Not restated: the feasibility restoration phase (reported as status 3
    if it would be entered), constraint-row scaling (all constraint gradients are < 100 here),
    the acceptable-point termination, watchdog, and the max_cpu_time cap.

It is deliberately independent of oracle/kmpc_condensed.c (different formulation, different
linear algebra, different language), so agreement between the two is evidence, not tautology.
Dense numpy/scipy linear algebra: sized for N <= 50, seconds per solve.
"""
import numpy as np
import scipy.linalg as sla

OPTIMAL, ITERATION_LIMIT, INFEASIBLE, NUMERICAL_ERROR = 0, 1, 2, 3


class Model:
    """Full-space JuMP model of MKZMPCPathFollower.jl (0-based k)."""

    def __init__(self, N=8, weights=(9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0), dt=0.2,
                 dt_control=0.1, L_a=1.108, L_b=1.742, steer_max=0.5, steer_dmax=0.5, a_max=1.0,
                 a_dmax=1.5, v_min=0.0, v_max=20.0):
        self.N, self.dt, self.dtc, self.L_a, self.L_b = N, dt, dt_control, L_a, L_b
        self.steer_max, self.steer_dmax, self.a_max, self.a_dmax = steer_max, steer_dmax, a_max, a_dmax
        self.v_min, self.v_max = v_min, v_max
        self.C = np.asarray(weights, float)  # update_cost order :158-169
        self.r = L_b / (L_a + L_b)
        n1 = N + 1
        self.ix, self.iy, self.iv, self.ip = 0, n1, 2 * n1, 3 * n1
        self.ia, self.id = 4 * n1, 4 * n1 + N
        self.nv = 6 * N + 4
        self.ns = 2 * (N - 1)
        self.nt = self.nv + self.ns
        self.nc = 4 + 4 * N + self.ns

    def set_problem(self, z0, ref, v_target, u_prev):
        self.z0 = np.asarray(z0, float)
        self.ref = np.asarray(ref, float).reshape(self.N + 1, 3)
        self.vt = float(v_target)
        self.up = np.asarray(u_prev, float)  # (acc_current, d_f_current)

    def bounds(self):
        N = self.N
        xL = np.full(self.nt, -np.inf)
        xU = np.full(self.nt, np.inf)
        xL[self.iv:self.iv + N + 1] = self.v_min
        xU[self.iv:self.iv + N + 1] = self.v_max
        xL[self.ia:self.ia + N] = -self.a_max
        xU[self.ia:self.ia + N] = self.a_max
        xL[self.id:self.id + N] = -self.steer_max
        xU[self.id:self.id + N] = self.steer_max
        for kk in range(N - 1):
            for j in range(2):
                d = (self.steer_dmax if j else self.a_dmax) * (self.dtc if kk == 0 else self.dt)
                xL[self.nv + 2 * kk + j] = -d
                xU[self.nv + 2 * kk + j] = d
        return xL, xU

    def rate_expr(self, w):
        """d(w): first step relative to the current command (:76,:83), then i=2..N-1 (:77-79,:84-86)."""
        N = self.N
        a = w[self.ia:self.ia + N]
        d = w[self.id:self.id + N]
        out = np.empty(self.ns)
        out[0] = a[0] - self.up[0]
        out[1] = d[0] - self.up[1]
        for k in range(1, N - 1):
            out[2 * k] = a[k + 1] - a[k]
            out[2 * k + 1] = d[k + 1] - d[k]
        return out

    def f(self, w):
        N, C = self.N, self.C
        x = w[self.ix:self.ix + N + 1]; y = w[self.iy:self.iy + N + 1]
        v = w[self.iv:self.iv + N + 1]; p = w[self.ip:self.ip + N + 1]
        a = w[self.ia:self.ia + N]; d = w[self.id:self.id + N]
        J = np.sum(C[0] * (x[1:] - self.ref[1:, 0]) ** 2 + C[1] * (y[1:] - self.ref[1:, 1]) ** 2
                   + C[2] * (p[1:] - self.ref[1:, 2]) ** 2)
        J += C[3] * np.sum((v[1:N] - self.vt) ** 2)
        J += C[6] * np.sum(a ** 2) + C[7] * np.sum(d ** 2)
        J += C[4] * np.sum(np.diff(a) ** 2) + C[5] * np.sum(np.diff(d) ** 2)
        return J

    def grad_f(self, w):
        N, C = self.N, self.C
        g = np.zeros(self.nt)
        x = w[self.ix:self.ix + N + 1]; y = w[self.iy:self.iy + N + 1]
        v = w[self.iv:self.iv + N + 1]; p = w[self.ip:self.ip + N + 1]
        a = w[self.ia:self.ia + N]; d = w[self.id:self.id + N]
        g[self.ix + 1:self.ix + N + 1] = 2 * C[0] * (x[1:] - self.ref[1:, 0])
        g[self.iy + 1:self.iy + N + 1] = 2 * C[1] * (y[1:] - self.ref[1:, 1])
        g[self.ip + 1:self.ip + N + 1] = 2 * C[2] * (p[1:] - self.ref[1:, 2])
        g[self.iv + 1:self.iv + N] = 2 * C[3] * (v[1:N] - self.vt)
        ga = 2 * C[6] * a
        gd = 2 * C[7] * d
        da, dd = np.diff(a), np.diff(d)
        ga[:-1] -= 2 * C[4] * da; ga[1:] += 2 * C[4] * da
        gd[:-1] -= 2 * C[5] * dd; gd[1:] += 2 * C[5] * dd
        g[self.ia:self.ia + N] = ga
        g[self.id:self.id + N] = gd
        return g

    def hess_f(self):
        N, C = self.N, self.C
        H = np.zeros((self.nt, self.nt))
        for k in range(1, N + 1):
            H[self.ix + k, self.ix + k] = 2 * C[0]
            H[self.iy + k, self.iy + k] = 2 * C[1]
            H[self.ip + k, self.ip + k] = 2 * C[2]
        for k in range(1, N):
            H[self.iv + k, self.iv + k] = 2 * C[3]
        for base, cu, cd in ((self.ia, C[6], C[4]), (self.id, C[7], C[5])):
            for k in range(N):
                H[base + k, base + k] += 2 * cu
            for k in range(N - 1):
                H[base + k, base + k] += 2 * cd
                H[base + k + 1, base + k + 1] += 2 * cd
                H[base + k, base + k + 1] -= 2 * cd
                H[base + k + 1, base + k] -= 2 * cd
        return H

    def _beta(self, d):
        r = self.r
        D = np.cos(d) ** 2 + r * r * np.sin(d) ** 2
        return np.arctan(r * np.tan(d)), r / D, r * (1 - r * r) * np.sin(2 * d) / D ** 2

    def c(self, w):
        N, dt = self.N, self.dt
        x = w[self.ix:self.ix + N + 1]; y = w[self.iy:self.iy + N + 1]
        v = w[self.iv:self.iv + N + 1]; p = w[self.ip:self.ip + N + 1]
        a = w[self.ia:self.ia + N]; d = w[self.id:self.id + N]
        b, _, _ = self._beta(d)
        out = np.empty(self.nc)
        out[0:4] = [x[0] - self.z0[0], y[0] - self.z0[1], p[0] - self.z0[2], v[0] - self.z0[3]]
        out[4:4 + N] = x[1:] - (x[:-1] + dt * v[:-1] * np.cos(p[:-1] + b))          # :119
        out[4 + N:4 + 2 * N] = y[1:] - (y[:-1] + dt * v[:-1] * np.sin(p[:-1] + b))  # :120
        out[4 + 2 * N:4 + 3 * N] = p[1:] - (p[:-1] + dt * v[:-1] / self.L_b * np.sin(b))  # :121
        out[4 + 3 * N:4 + 4 * N] = v[1:] - (v[:-1] + dt * a)                         # :122
        out[4 + 4 * N:] = self.rate_expr(w) - w[self.nv:]
        return out

    def jac_c(self, w):
        N, dt, Lb = self.N, self.dt, self.L_b
        Jm = np.zeros((self.nc, self.nt))
        v = w[self.iv:self.iv + N + 1]; p = w[self.ip:self.ip + N + 1]
        d = w[self.id:self.id + N]
        b, b1, _ = self._beta(d)
        Jm[0, self.ix] = Jm[1, self.iy] = Jm[2, self.ip] = Jm[3, self.iv] = 1.0
        for k in range(N):
            cs, sn = np.cos(p[k] + b[k]), np.sin(p[k] + b[k])
            r0 = 4 + k
            Jm[r0, self.ix + k + 1] = 1; Jm[r0, self.ix + k] = -1
            Jm[r0, self.iv + k] = -dt * cs
            Jm[r0, self.ip + k] = dt * v[k] * sn
            Jm[r0, self.id + k] = dt * v[k] * sn * b1[k]
            r1 = 4 + N + k
            Jm[r1, self.iy + k + 1] = 1; Jm[r1, self.iy + k] = -1
            Jm[r1, self.iv + k] = -dt * sn
            Jm[r1, self.ip + k] = -dt * v[k] * cs
            Jm[r1, self.id + k] = -dt * v[k] * cs * b1[k]
            r2 = 4 + 2 * N + k
            Jm[r2, self.ip + k + 1] = 1; Jm[r2, self.ip + k] = -1
            Jm[r2, self.iv + k] = -dt / Lb * np.sin(b[k])
            Jm[r2, self.id + k] = -dt * v[k] / Lb * np.cos(b[k]) * b1[k]
            r3 = 4 + 3 * N + k
            Jm[r3, self.iv + k + 1] = 1; Jm[r3, self.iv + k] = -1
            Jm[r3, self.ia + k] = -dt
        rr = 4 + 4 * N
        Jm[rr, self.ia] = 1.0
        Jm[rr + 1, self.id] = 1.0
        for k in range(1, N - 1):
            Jm[rr + 2 * k, self.ia + k + 1] = 1; Jm[rr + 2 * k, self.ia + k] = -1
            Jm[rr + 2 * k + 1, self.id + k + 1] = 1; Jm[rr + 2 * k + 1, self.id + k] = -1
        for r in range(self.ns):
            Jm[rr + r, self.nv + r] = -1.0
        return Jm

    def hess_c(self, w, lam):
        """sum_j lam_j * Hessian(c_j); only the dynamics rows are nonlinear, in (psi_k, v_k, d_f_k)."""
        N, dt, Lb = self.N, self.dt, self.L_b
        W = np.zeros((self.nt, self.nt))
        v = w[self.iv:self.iv + N + 1]; p = w[self.ip:self.ip + N + 1]
        d = w[self.id:self.id + N]
        b, b1, b2 = self._beta(d)
        for k in range(N):
            lx, ly, lp = lam[4 + k], lam[4 + N + k], lam[4 + 2 * N + k]
            cs, sn, cb, sb = np.cos(p[k] + b[k]), np.sin(p[k] + b[k]), np.cos(b[k]), np.sin(b[k])
            vk = v[k]
            # c = z+ - f  ->  Hessian(c) = -Hessian(f)
            pp = -(lx * (-dt * vk * cs) + ly * (-dt * vk * sn))
            pv = -(lx * (-dt * sn) + ly * (dt * cs))
            pd = -(lx * (-dt * vk * cs * b1[k]) + ly * (-dt * vk * sn * b1[k]))
            vd = -(lx * (-dt * sn * b1[k]) + ly * (dt * cs * b1[k]) + lp * (dt / Lb * cb * b1[k]))
            dd = -(lx * (-dt * vk * (cs * b1[k] ** 2 + sn * b2[k])) + ly * (dt * vk * (-sn * b1[k] ** 2 + cs * b2[k]))
                   + lp * (dt * vk / Lb * (-sb * b1[k] ** 2 + cb * b2[k])))
            ip_, iv_, id_ = self.ip + k, self.iv + k, self.id + k
            W[ip_, ip_] += pp
            W[ip_, iv_] += pv; W[iv_, ip_] += pv
            W[ip_, id_] += pd; W[id_, ip_] += pd
            W[iv_, id_] += vd; W[id_, iv_] += vd
            W[id_, id_] += dd
        return W

    def rollout_start(self):
        """full-space point whose states are the roll-out (:115-122) of u_0 = previous command, u_k>0 = 0"""
        N, dt = self.N, self.dt
        w = np.zeros(self.nt)
        a = np.zeros(N); d = np.zeros(N)
        a[0] = np.clip(self.up[0], -self.a_max, self.a_max)
        d[0] = np.clip(self.up[1], -self.steer_max, self.steer_max)
        z = self.z0.copy()  # x, y, psi, v
        X = [z.copy()]
        for k in range(N):
            b = np.arctan(self.r * np.tan(d[k]))
            z = np.array([z[0] + dt * z[3] * np.cos(z[2] + b), z[1] + dt * z[3] * np.sin(z[2] + b),
                          z[2] + dt * z[3] / self.L_b * np.sin(b), z[3] + dt * a[k]])
            X.append(z.copy())
        X = np.array(X)
        w[self.ix:self.ix + N + 1] = X[:, 0]; w[self.iy:self.iy + N + 1] = X[:, 1]
        w[self.ip:self.ip + N + 1] = X[:, 2]; w[self.iv:self.iv + N + 1] = X[:, 3]
        w[self.ia:self.ia + N] = a; w[self.id:self.id + N] = d
        return w

    def unpack(self, w):
        N = self.N
        X = np.stack([w[self.ix:self.ix + N + 1], w[self.iy:self.iy + N + 1],
                      w[self.ip:self.ip + N + 1], w[self.iv:self.iv + N + 1]], axis=1)
        U = np.stack([w[self.ia:self.ia + N], w[self.id:self.id + N]], axis=1)
        return U, X


def _inertia(Kmat):
    """(n_pos, n_neg, n_zero) from a Bunch-Kaufman LDL^T factorisation."""
    _, D, _ = sla.ldl(Kmat, lower=True, hermitian=True)
    n = D.shape[0]
    pos = neg = zero = 0
    i = 0
    while i < n:
        if i + 1 < n and D[i + 1, i] != 0.0:
            ev = np.linalg.eigvalsh(D[i:i + 2, i:i + 2])
            i += 2
        else:
            ev = [D[i, i]]
            i += 1
        for e in ev:
            if e > 0.0 and np.isfinite(e):
                pos += 1
            elif e < 0.0 and np.isfinite(e):
                neg += 1
            else:
                zero += 1
    return pos, neg, zero


def solve(model, max_iter=3000, tol=1e-8, mu_init=0.1, bound_relax=1e-8, verbose=False):
    """Ipopt-style solve of the full-space model from the reference's cold start (all primals 0, Q9).
    Where Ipopt would enter its feasibility-restoration phase (not restated) the solve is repeated
    from a dynamically consistent start (Model.rollout_start); the result says which start was used.
    The KKT point reached does not depend on the start for these problems (checked against two
    other solvers in make_golden.py)."""
    r = _solve(model, None, max_iter, tol, mu_init, bound_relax, verbose)
    r["start"] = "zeros"
    if r["status"] == NUMERICAL_ERROR:
        r = _solve(model, model.rollout_start(), max_iter, tol, mu_init, bound_relax, verbose)
        r["start"] = "rollout"
    return r


def _solve(model, x_start, max_iter, tol, mu_init, bound_relax, verbose):
    m = model
    nt, nc = m.nt, m.nc
    xL, xU = m.bounds()
    hasL, hasU = np.isfinite(xL), np.isfinite(xU)
    # bound_relax_factor
    xL = np.where(hasL, xL - bound_relax * np.maximum(1.0, np.abs(xL)), xL)
    xU = np.where(hasU, xU + bound_relax * np.maximum(1.0, np.abs(xU)), xU)
    # starting point: JuMP start=0.0 for every variable; slacks start at d(x0)
    x = np.zeros(nt) if x_start is None else np.array(x_start, float)
    x[m.nv:] = m.rate_expr(x)
    k1 = k2 = 1e-2
    both = hasL & hasU
    with np.errstate(invalid="ignore"):
      pL = np.where(both, np.minimum(k1 * np.maximum(1.0, np.abs(xL)), k2 * (xU - xL)), k1 * np.maximum(1.0, np.abs(xL)))
      pU = np.where(both, np.minimum(k1 * np.maximum(1.0, np.abs(xU)), k2 * (xU - xL)), k1 * np.maximum(1.0, np.abs(xU)))
    with np.errstate(invalid="ignore"):
        x = np.where(hasL, np.maximum(x, xL + pL), x)
        x = np.where(hasU, np.minimum(x, xU - pU), x)

    g0 = m.grad_f(x)
    sc = min(1.0, 100.0 / max(np.abs(g0).max(), 1e-300))
    Hf = sc * m.hess_f()
    zL = np.where(hasL, 1.0, 0.0)
    zU = np.where(hasU, 1.0, 0.0)
    # least-squares multipliers
    Jc = m.jac_c(x)
    KK = np.block([[np.eye(nt), Jc.T], [Jc, np.zeros((nc, nc))]])
    rhs = -np.concatenate([sc * g0 - zL + zU, np.zeros(nc)])
    try:
        sol = np.linalg.solve(KK, rhs)
        lam = sol[nt:]
        if np.abs(lam).max() > 1e3:
            lam = np.zeros(nc)
    except np.linalg.LinAlgError:
        lam = np.zeros(nc)

    mu = mu_init
    kappa_eps, kappa_mu, theta_mu, tau_min, s_max, kappa_sigma = 10.0, 0.2, 1.5, 0.99, 100.0, 1e10
    gamma_theta, gamma_phi, eta_phi, delta_sw, s_theta, s_phi = 1e-5, 1e-8, 1e-8, 1.0, 1.1, 2.3
    kappa_soc, p_max = 0.99, 4
    dw_last = 0.0
    filt = []
    th0 = np.abs(m.c(x)).sum()
    theta_max, theta_min = 1e4 * max(1.0, th0), 1e-4 * max(1.0, th0)

    def phi(xx, mu_):
        sl = xx[hasL] - xL[hasL]
        su = xU[hasU] - xx[hasU]
        if (sl <= 0).any() or (su <= 0).any():
            return np.inf
        return sc * m.f(xx) - mu_ * (np.log(sl).sum() + np.log(su).sum())

    def errors(mu_):
        gL = sc * m.grad_f(x) + Jc.T @ lam - zL + zU
        cv = m.c(x)
        sd = max(s_max, (np.abs(lam).sum() + zL.sum() + zU.sum()) / (nc + hasL.sum() + hasU.sum())) / s_max
        scc = max(s_max, (zL.sum() + zU.sum()) / (hasL.sum() + hasU.sum())) / s_max
        comp = max(np.abs((x - xL)[hasL] * zL[hasL] - mu_).max(), np.abs((xU - x)[hasU] * zU[hasU] - mu_).max())
        return max(np.abs(gL).max() / sd, np.abs(cv).max(), comp / scc), np.abs(gL).max() / sd, np.abs(cv).max()

    status, it = ITERATION_LIMIT, 0
    for it in range(max_iter):
        Jc = m.jac_c(x)
        E0, dinf, cinf = errors(0.0)
        if verbose:
            print("it %3d f %.10g E0 %.3e dinf %.2e cinf %.2e mu %.1e" % (it, m.f(x), E0, dinf, cinf, mu))
        if E0 <= tol:
            status = OPTIMAL
            break
        while mu > tol / 10.0 and errors(mu)[0] <= kappa_eps * mu:
            mu = max(tol / 10.0, min(kappa_mu * mu, mu ** theta_mu))
            filt = []
        tau = max(tau_min, 1.0 - mu)
        sL = np.where(hasL, x - xL, 1.0)
        sU = np.where(hasU, xU - x, 1.0)
        Sigma = np.where(hasL, zL / sL, 0.0) + np.where(hasU, zU / sU, 0.0)
        W = Hf + m.hess_c(x, lam)
        gphi = sc * m.grad_f(x) - np.where(hasL, mu / sL, 0.0) + np.where(hasU, mu / sU, 0.0)
        cv = m.c(x)
        # inertia correction (Algorithm IC)
        dw, dc = 0.0, 0.0
        first_try = True
        while True:
            Kmat = np.block([[W + np.diag(Sigma) + dw * np.eye(nt), Jc.T], [Jc, -dc * np.eye(nc)]])
            pos, neg, zero = _inertia(Kmat)
            if pos == nt and neg == nc and zero == 0:
                break
            if zero > 0:
                dc = 1e-8 * mu ** 0.25
            if dw == 0.0:
                dw = 1e-4 if dw_last == 0.0 else max(1e-20, dw_last / 3.0)
            else:
                dw = dw * (100.0 if (dw_last == 0.0 and first_try) else 8.0)
                first_try = False
            if dw > 1e40:
                return dict(status=NUMERICAL_ERROR, iters=it)
        if dw > 0.0:
            dw_last = dw
        sol = np.linalg.solve(Kmat, -np.concatenate([gphi + Jc.T @ lam, cv]))
        dx, dlam = sol[:nt], sol[nt:]
        dzL = np.where(hasL, mu / sL - zL - zL / sL * dx, 0.0)
        dzU = np.where(hasU, mu / sU - zU + zU / sU * dx, 0.0)

        def ftb(dxx):
            a = 1.0
            mL = hasL & (dxx < 0)
            if mL.any():
                a = min(a, (-tau * (x - xL)[mL] / dxx[mL]).min())
            mU = hasU & (dxx > 0)
            if mU.any():
                a = min(a, (tau * (xU - x)[mU] / dxx[mU]).min())
            return a

        a_max = ftb(dx)
        a_z = 1.0
        mz = hasL & (dzL < 0)
        if mz.any():
            a_z = min(a_z, (-tau * zL[mz] / dzL[mz]).min())
        mz = hasU & (dzU < 0)
        if mz.any():
            a_z = min(a_z, (-tau * zU[mz] / dzU[mz]).min())

        theta = np.abs(cv).sum()
        ph = phi(x, mu)
        gd = gphi @ dx

        def acceptable(th_t, ph_t, alpha, g_dir):
            if not np.isfinite(ph_t) or th_t > theta_max:
                return False, False
            for (tf, pf) in filt:
                if th_t >= tf and ph_t >= pf:
                    return False, False
            switching = g_dir < 0 and alpha * (-g_dir) ** s_phi > delta_sw * theta ** s_theta
            if theta <= theta_min and switching:
                return ph_t <= ph + eta_phi * alpha * g_dir + 10 * 2.2e-16 * abs(ph), True
            ok = th_t <= (1 - gamma_theta) * theta or ph_t <= ph - gamma_phi * theta
            return ok, False

        alpha = a_max
        accepted, ftype, x_new = False, False, None
        alpha_min = 1e-14
        l = 0
        while alpha > alpha_min:
            xt = x + alpha * dx
            th_t, ph_t = np.abs(m.c(xt)).sum(), phi(xt, mu)
            ok, ft = acceptable(th_t, ph_t, alpha, gd)
            if ok:
                accepted, ftype, x_new = True, ft, xt
                break
            if l == 0 and th_t >= theta:
                # second-order correction (A-5.5 .. A-5.9)
                c_soc = alpha * cv + m.c(xt)
                th_old = theta
                for _ in range(p_max):
                    sol2 = np.linalg.solve(Kmat, -np.concatenate([gphi + Jc.T @ lam, c_soc]))
                    dxc = sol2[:nt]
                    a_soc = ftb(dxc)
                    xs = x + a_soc * dxc
                    th_s, ph_s = np.abs(m.c(xs)).sum(), phi(xs, mu)
                    ok, ft = acceptable(th_s, ph_s, alpha, gd)
                    if ok:
                        accepted, ftype, x_new = True, ft, xs
                        dlam = sol2[nt:]
                        break
                    if th_s > kappa_soc * th_old:
                        break
                    th_old = th_s
                    c_soc = a_soc * c_soc + m.c(xs)
                if accepted:
                    break
            alpha *= 0.5
            l += 1
        if not accepted:
            # Ipopt would enter feasibility restoration here; not restated.
            return dict(status=NUMERICAL_ERROR, iters=it, note="restoration phase needed")
        if not ftype:
            filt.append(((1 - gamma_theta) * theta, ph - gamma_phi * theta))
        x = x_new
        lam = lam + alpha * dlam
        zL = zL + a_z * dzL
        zU = zU + a_z * dzU
        sL = np.where(hasL, x - xL, 1.0)
        sU = np.where(hasU, xU - x, 1.0)
        zL = np.where(hasL, np.clip(zL, mu / (kappa_sigma * sL), kappa_sigma * mu / sL), 0.0)
        zU = np.where(hasU, np.clip(zU, mu / (kappa_sigma * sU), kappa_sigma * mu / sU), 0.0)

    U, X = m.unpack(x)
    return dict(status=status, iters=it, U=U, X=X, cost=m.f(x), kkt=E0, mu=mu,
                constr_viol=np.abs(m.c(x)).max(), zL=zL / sc, zU=zU / sc, lam=lam / sc, w=x)


def solve_problem(N, z0, ref, v_target, u_prev=(0.0, 0.0), weights=(9.0, 9.0, 10.0, 0.0, 100.0, 1000.0, 0.0, 0.0),
                  **kw):
    m = Model(N=N, weights=weights)
    m.set_problem(z0, ref, v_target, u_prev)
    return solve(m, **kw)
