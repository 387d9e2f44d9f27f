"""Independent certification of solver outputs (test helper; uses oracle/ -- test infrastructure).

A returned input sequence U is certified WITHOUT trusting anything the solver reports.  The gradient of the state-eliminated
objective comes from oracle/kmpc_nlp.c (costate sweep), the inequality rows from kmpc_ineq().  A certificate is a multiplier
vector lam >= 0; (U, lam) goes through oracle/kmpc_nlp.c::kmpc_certify, which returns

    stationarity     || grad J(U) + A^T lam ||_inf
    violation        max_i (A U - b)_i            (UNRELAXED bounds of MKZMPCPathFollower.jl:65-86)
    complementarity  max_i lam_i * slack_i
    lam_min

Candidate certificates: for each activity threshold t in 1e-6 ... 1, inf the non-negative least-squares multipliers on the rows
whose slack is below t,  lam = argmin_{lam >= 0} ||grad J + A_act^T lam||_2  (zero elsewhere) -- and the same with the remaining
rows filled with mu/slack_i (mu = median complementarity product of the active rows: what a point on the central path carries).
An interior-point solution with complementarity ~1e-8 has no sharp active set (multipliers decay geometrically along chains
of rate limits down to ~1e-6 at slacks ~1e-2), so the threshold trades stationarity against complementarity; the certificate
reported is the candidate with the smallest max(stationarity, complementarity).  Where none of these is good (best worse than AUG_IF_ABOVE) two more are
tried: multipliers on all rows from the least-squares trade-off ||grad J + A^T lam||^2 + rho^2 ||diag(slack) lam||^2, rho = 0.1, 1 (round 4).

Scaling.  Ipopt tests optimality on a SCALED problem (Waechter & Biegler 2006, eq. 5-6; option nlp_scaling_max_gradient = 100): the
objective is multiplied by sc = min(1, 100 / |grad f(x_start)|_inf), fixed at the starting point, and the residuals are divided by
s_d = max(100, |lam|_1 / m) / 100.  Two scalings are reported, both computed from the problem and U alone:
  scaled_*      STRICT: sc from the gradient at the returned point, 1 / max(1, |grad J(U)|_inf / 100) -- at a solution the gradient is
                what the active multipliers balance, typically 100-1000x smaller than at any starting point, so this is 1-3 decades
                harsher than the test any Ipopt run applies;
  ref_scaled_*  REFERENCE: sc from the gradient at the reference's own starting point, all inputs zero (`start=0.0`,
                MKZMPCPathFollower.jl:65-72; Q9), 1 / max(1, |grad J(0)|_inf / 100) -- the scale on which the reference's
                `tol = 1e-8` is stated.
"""
import os
import sys

import numpy as np
from scipy.optimize import nnls

THRESHOLDS = (1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0, np.inf)
AUG_IF_ABOVE_F32 = 1e-3   # ... for fp32 solutions (callers pass bound_relax 1e-5): their certificates sit at 1e-4 ... 1e-2 whatever the multipliers
AUG_IF_ABOVE = 2e-8   # the all-rows least-squares candidates are tried while the best active-set certificate is worse than this (STRICT scale)
WIDE_FROM, WIDE_IF_ABOVE = 1.0, 2e-7   # thresholds >= 1 (NNLS over most rows: tens of ms at N = 50) only while the best certificate is worse than this


def _nnls(M, rhs):
    return nnls(M, rhs, maxiter=50 * M.shape[1] + 200)[0]


def certify_one(O, p, z0, ref, v_target, u_prev, U, relax=1e-8):
    return certify_problem(O, p, O.problem(p, z0, ref, v_target, u_prev), U, relax)


def certify_problem(O, p, q, U, relax=1e-8):
    """q: an oracle problem object (O.problem for the Cartesian model, O.problem_frenet for the Frenet functor)"""
    U = np.asarray(U, dtype=np.float64).reshape(-1)
    A, b = O.ineq(p, q, relax=relax)
    g = O.grad(p, q, U)
    slack = b - A @ U
    osc = max(1.0, np.abs(g).max() / 100.0)
    m = len(b)
    best = None
    n_prev = -1
    for thr in THRESHOLDS:
        act = slack <= thr * np.maximum(1.0, np.abs(b))
        if best is not None and ((thr >= WIDE_FROM and best[0] <= WIDE_IF_ABOVE) or int(act.sum()) == n_prev):
            continue
        n_prev = int(act.sum())
        cands = [np.zeros(m)]
        if act.any():
            cands[0][act] = _nnls(A[act].T, -g)
            pr = (cands[0] * slack)[act & (cands[0] > 0) & (slack > 0)]   # (a rounded fp32 iterate may sit a hair outside a relaxed bound)
            if len(pr) and (~act).any() and slack[~act].min() > 0:
                lam = np.zeros(m)
                lam[~act] = np.median(pr) / slack[~act]
                lam[act] = _nnls(A[act].T, -(g + A[~act].T @ lam[~act]))
                cands.append(lam)
        for lam in cands:
            sd = max(100.0, lam.sum() / m) / 100.0
            stat = np.abs(g + A.T @ lam).max() / (osc * sd)
            comp = (lam * np.maximum(slack, 0.0)).max() / (osc * sd)
            if best is None or max(stat, comp) < best[0]:
                best = (max(stat, comp), lam, osc * sd, thr)
    # Round 4: one more family of candidates, tried while the best one is worse than AUG_IF_ABOVE -- multipliers on ALL rows that trade stationarity against
    # complementarity in the least-squares sense,  lam = argmin_{lam >= 0} ||grad J + A^T lam||^2 + rho^2 ||diag(slack) lam||^2  (one NNLS with m extra
    # rows).  An interior-point solution carries multipliers ~mu/slack on every row; where rows with slacks 1e-4 ... 1e-2 still matter to the gradient
    # balance (hard braking into a saturated steering ramp: found by the out-of-distribution sweep) no active-set threshold gives a
    # good certificate although one exists (the solver's own multipliers; the LP that minimises max(stationarity, complementarity) confirms the optimum).
    if best[0] > (AUG_IF_ABOVE if relax <= 1e-7 else AUG_IF_ABOVE_F32):
        sp = np.maximum(slack, 0.0)
        # all rows up to N = 20 (m <= 196); beyond, the rows within 0.3 of their bound (the others' multipliers ~mu / slack no longer matter; keeps the NNLS small)
        near = slack <= (np.inf if m <= 200 else 0.3) * np.maximum(1.0, np.abs(b))
        if not near.any():
            near[:] = True
        for rho in (0.1, 1.0):
            lam = np.zeros(m)
            lam[near] = _nnls(np.vstack([A[near].T, rho * np.diag(sp[near])]), np.concatenate([-g, np.zeros(int(near.sum()))]))
            sd = max(100.0, lam.sum() / m) / 100.0
            stat = np.abs(g + A.T @ lam).max() / (osc * sd)
            comp = (lam * sp).max() / (osc * sd)
            if max(stat, comp) < best[0]:
                best = (max(stat, comp), lam, osc * sd, -rho)   # (reported as a negative "threshold")
    _, lam, scale, thr = best
    c = O.certify(p, q, U, lam)   # the independent C evaluation of (U, lam)
    c["scaled_stationarity"] = c["stationarity"] / scale
    c["scaled_complementarity"] = c["complementarity"] / scale
    ref_scale = scale / osc * max(1.0, np.abs(O.grad(p, q, np.zeros_like(U))).max() / 100.0)
    c["ref_scaled_stationarity"] = c["stationarity"] / ref_scale
    c["ref_scaled_complementarity"] = c["complementarity"] / ref_scale
    c["threshold"] = thr
    return c


KEYS = ("scaled_stationarity", "scaled_complementarity", "ref_scaled_stationarity", "ref_scaled_complementarity", "stationarity", "violation", "lam_min", "complementarity", "cost", "threshold")


_POOL = None


def _pool():
    """worker processes for large certification jobs (spawned, never forked: the test process may hold a GPU context; the workers touch only numpy and oracle/)"""
    global _POOL
    if _POOL is None:
        import atexit
        import multiprocessing as mp
        n = int(os.environ.get("KMPC_CERT_WORKERS", min(8, os.cpu_count() or 1)))
        keep = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        os.environ.update({k: "1" for k in keep})  # one BLAS thread per worker (inherited at spawn)
        # the workers import THIS module by name; they must not re-run the parent's main script (a spawned child re-imports `__main__` from its file: an
        # unguarded tool script would repeat its GPU work in every worker -- the GPU box allows six processes on the card), so the file is hidden meanwhile
        main = sys.modules.get("__main__")
        hidden = {a: getattr(main, a) for a in ("__file__", "__spec__") if getattr(main, a, None) is not None}
        for a in hidden:
            setattr(main, a, None)
        try:
            _POOL = mp.get_context("spawn").Pool(n) if n > 1 else False
        finally:
            for a, v in hidden.items():
                setattr(main, a, v)
        for k, v in keep.items():
            os.environ.pop(k) if v is None else os.environ.__setitem__(k, v)
        if _POOL:
            atexit.register(_POOL.terminate)
    return _POOL


def _certify_chunk(a):
    N, weights, model, z0, ref, vt, up, U, kw = a
    from oracle import oracle as O
    p = O.params(N, weights, model=model)
    rows = []
    for i in range(len(U)):
        c = certify_one(O, p, z0[i], ref[i], vt[i], up[i], U[i], **kw)
        rows.append([c[k] for k in KEYS])
    return rows


def certify_batch(O, p, d, U, idx=None, **kw):
    """-> dict of arrays over the certified problems (idx = indices into the batch, default all); jobs of 256 problems and more are spread over worker processes"""
    idx = np.arange(len(U)) if idx is None else np.asarray(idx)
    pool = _pool() if len(idx) >= 256 else None
    if pool:
        chunks = np.array_split(idx, max(1, min(len(idx) // 16, 8 * pool._processes)))
        args = [(int(p.N), [float(p.C[i]) for i in range(8)], int(p.model), d["z0"][c], d["ref"][c], d["v_target"][c], d["u_prev"][c], np.asarray(U)[c], kw) for c in chunks if len(c)]
        rows = [r for part in pool.map(_certify_chunk, args) for r in part]
        return {k: np.array([r[j] for r in rows]) for j, k in enumerate(KEYS)}
    out = {k: [] for k in KEYS}
    for i in idx:
        c = certify_one(O, p, d["z0"][i], d["ref"][i], d["v_target"][i], d["u_prev"][i], U[i], **kw)
        for k in KEYS:
            out[k].append(c[k])
    return {k: np.array(v) for k, v in out.items()}


def stratified_sample(iters, status, n, seed=0):
    """indices of a sample of size <= n: non-Optimal problems (up to n/4), the n/8 problems with the most iterations, the rest uniform"""
    B = len(iters)
    if B <= n:
        return np.arange(B)
    rng = np.random.default_rng(seed)
    bad = np.where(status != 0)[0][: n // 4]
    top = np.argsort(-iters, kind="stable")[: n // 8]
    pick = set(bad.tolist()) | set(top.tolist())
    for i in rng.permutation(B):
        if len(pick) >= n:
            break
        pick.add(int(i))
    return np.array(sorted(pick))


def second_order_check(O, p, q, U, act_tol=1e-6, h=1e-5):
    """Second-order NECESSARY condition of a local minimum at U: the Hessian of the state-eliminated objective (every constraint of the
    condensed program is linear in U, so it is the Lagrangian Hessian) is positive semi-definite on the null space of the active rows.
    The Hessian is a central finite difference of oracle/kmpc_nlp.c's costate gradient -- independent of the condensing code of
    oracle/kmpc_condensed.c and of the kernels.  Returns (smallest eigenvalue of Z^T H Z divided by max(1, max |H|), dim Z)."""
    U = np.asarray(U, dtype=np.float64).reshape(-1)
    n = len(U)
    A, b = O.ineq(p, q, relax=1e-8)
    act = (b - A @ U) <= act_tol * np.maximum(1.0, np.abs(b))
    H = np.empty((n, n))
    for j in range(n):
        e = np.zeros(n)
        e[j] = h
        H[:, j] = (O.grad(p, q, U + e) - O.grad(p, q, U - e)) / (2.0 * h)
    H = 0.5 * (H + H.T)
    if act.any():
        _, sv, Vt = np.linalg.svd(A[act], full_matrices=True)
        rank = int((sv > 1e-10 * max(1.0, sv.max())).sum())
        Z = Vt[rank:].T
    else:
        Z = np.eye(n)
    if Z.shape[1] == 0:
        return 0.0, 0
    ev = np.linalg.eigvalsh(Z.T @ H @ Z)
    return float(ev.min() / max(1.0, np.abs(H).max())), int(Z.shape[1])
