"""GPU unit tests of the HIP building blocks against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from mkz_mpc_path_follower_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu


def _solver(N, dtype=torch.float64, **kw):
    from mkz_mpc_path_follower_amd import BatchMPC
    return BatchMPC(N=N, dtype=dtype, **kw)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_mfma_16x16x4_layout(dtype):
    """D = A(16x4) B(4x16): operand lane maps A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
    C/D col = l&15, row = (l>>4)+4r for f64 and 4(l>>4)+r for f32 (asymmetric integer data)."""
    s = _solver(8, dtype)
    rng = np.random.default_rng(0)
    A = rng.integers(-4, 5, (16, 4)).astype(np.float64)
    Bm = rng.integers(-4, 5, (4, 16)).astype(np.float64)
    lanes = np.arange(64)
    a = A[lanes & 15, lanes >> 4]
    b = Bm[lanes >> 4, lanes & 15]
    d = s.debug_mfma_probe(a, b).cpu().numpy().astype(np.float64)
    D = A @ Bm
    got = np.zeros((16, 16))
    for l in range(64):
        for r in range(4):
            row = (l >> 4) + 4 * r if dtype == torch.float64 else 4 * (l >> 4) + r
            got[row, l & 15] = d[l, r]
    assert np.array_equal(got, D), "MFMA C/D layout differs:\nexpected\n%s\ngot\n%s" % (D, got)


@pytest.mark.parametrize("N", [8, 20, 50])
@pytest.mark.parametrize("hessian", [0, 1])
def test_condense_matches_oracle(oracle, N, hessian):
    """Roll-out, costate gradient and MFMA condensing == oracle/kmpc_condensed.c::kmpc_condense."""
    O = oracle
    B = 6
    d = make_batch(B, N, cfg_id=7, seed=1234 + N)
    rng = np.random.default_rng(5)
    U = np.stack([rng.uniform(-0.8, 0.8, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], axis=-1)
    s = _solver(N)
    H, g, J = s.debug_condense(d["z0"], d["ref"], d["v_target"], U, hessian=hessian)
    H, g, J = H.cpu().numpy(), g.cpu().numpy(), J.cpu().numpy()
    p = O.params(N)
    for b in range(B):
        q = O.problem(p, d["z0"][b], d["ref"][b], d["v_target"][b])
        Ho, go, Jo = O.condense(p, q, U[b], hessian=hessian)
        scale = np.abs(Ho).max()
        assert abs(J[b] - Jo) <= 1e-10 * max(1.0, abs(Jo))
        assert np.abs(g[b] - go).max() <= 1e-9 * max(1.0, np.abs(go).max())
        assert np.abs(H[b] - Ho).max() <= 1e-10 * scale, (b, np.abs(H[b] - Ho).max(), scale)


def _forms_matrix(N, dt=0.2):
    """the 5N-2 two-sided linear forms a_f^T U of the state-eliminated problem (boxes, rate forms, speed prefix sums), in the kernels' order"""
    n, R, nf = 2 * N, 2 * (N - 1), 5 * N - 2
    A = np.zeros((nf, n))
    for f in range(n):
        A[f, f] = 1.0
    for r in range(R):
        if r < 2:
            A[n + r, r] = 1.0                      # first-step rows: u_0 against the previous command (MKZMPCPathFollower.jl:76,83)
        else:
            A[n + r, r + 2] = 1.0                  # u_{k+1} - u_k for k = 1 .. N-2 (:77-79,84-86; the k = 0 pair is free, Q1)
            A[n + r, r] = -1.0
    for k in range(N):
        A[n + R + k, 0:2 * k + 1:2] = dt           # v_{k+1} - v_0 = dt * sum_{j<=k} acc_j (:122)
    return A


@pytest.mark.parametrize("N,dtype", [(8, torch.float64), (20, torch.float64), (28, torch.float64), (50, torch.float64), (48, torch.float64), (44, torch.float64), (40, torch.float32), (36, torch.float64), (32, torch.float32),
                                     (8, torch.float32), (20, torch.float32), (28, torch.float32)])
@pytest.mark.parametrize("hessian", [0, 1])
def test_kkt_pipeline_of_the_solve_kernels(oracle, N, dtype, hessian):
    """The code BENCH times, block by block (kmpc_debug_kkt dispatches the same FastSolver / WideSolver members the solve uses):
    condensing (adjoint recursion in the compile-time-horizon kernels, matrix cores in the generic one) + in-register KKT assembly == sc*H + A^T W A + reg*I built from the oracle's Hessian in numpy;
    blocked Cholesky + block-LDL^T substitutions == numpy's solve of that system.
    fp64: |K - K_ref| <= 1e-10 max|K|, gradient 1e-9, solve residual <= 1e-9 |rhs|;  fp32: 2e-5 / 1e-4 / 2e-3."""
    O = oracle
    B = 6
    f64 = dtype == torch.float64
    d = make_batch(B, N, cfg_id=7, seed=4321 + N)
    rng = np.random.default_rng(11 + N)
    U = np.stack([rng.uniform(-0.8, 0.8, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], axis=-1)
    nf, n = 5 * N - 2, 2 * N
    w = 10.0 ** rng.uniform(-2, 6 if f64 else 3, (B, nf))
    bb = rng.normal(0, 1, (B, n))
    sc, reg = 0.37, 2.5
    s = _solver(N, dtype)
    K, g, x, ok = s.debug_kkt(d["z0"], d["ref"], d["v_target"], d["u_prev"], U, w, bb, sc=sc, reg=reg, hessian=hessian)
    K, g, x, ok = K.double().cpu().numpy(), g.double().cpu().numpy(), x.double().cpu().numpy(), ok.cpu().numpy()
    A = _forms_matrix(N)
    p = O.params(N)
    tK, tg, tr = (1e-10, 1e-9, 1e-9) if f64 else (2e-5, 1e-4, 2e-3)
    for b in range(B):
        q = O.problem(p, d["z0"][b], d["ref"][b], d["v_target"][b], d["u_prev"][b])
        Ho, go, _ = O.condense(p, q, U[b], hessian=hessian)
        wb = w[b].astype(np.float32).astype(np.float64) if not f64 else w[b]
        Kr = sc * Ho + A.T @ (wb[:, None] * A) + reg * np.eye(n)
        assert np.abs(K[b] - Kr).max() <= tK * np.abs(Kr).max(), (b, np.abs(K[b] - Kr).max(), np.abs(Kr).max())
        assert np.abs(g[b] - go).max() <= tg * max(1.0, np.abs(go).max())
        if np.linalg.eigvalsh(Kr).min() > 0:
            assert ok[b] == 1
            rhs = bb[b] - sc * go
            assert np.abs(Kr @ x[b] - rhs).max() <= tr * max(1.0, np.abs(rhs).max()) * (1.0 if f64 else np.sqrt(np.linalg.cond(Kr))), b
            xr = np.linalg.solve(Kr, rhs)
            assert np.abs(x[b] - xr).max() <= (1e-6 if f64 else 2e-2) * max(1e-12, np.abs(xr).max()), b
    # an indefinite matrix is reported, not factored: exact Hessian, no barrier weights, no shift, inputs far from any minimum
    if hessian == 1 and f64:
        K2, g2, x2, ok2 = s.debug_kkt(d["z0"], d["ref"], d["v_target"], d["u_prev"], U, np.zeros((B, nf)), bb, sc=1.0, reg=0.0, hessian=1)
        K2, ok2 = K2.cpu().numpy(), ok2.cpu().numpy()
        for b in range(B):
            pd = np.linalg.eigvalsh(K2[b]).min() > 1e-9 * np.abs(K2[b]).max()
            nd = np.linalg.eigvalsh(K2[b]).min() < -1e-9 * np.abs(K2[b]).max()
            assert (ok2[b] == 1) if pd else ((ok2[b] == 0) if nd else True)
