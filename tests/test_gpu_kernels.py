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
