"""The four-problems-per-wave kernel at the reference's own horizon N = 8 (csrc/kmpc_quad.hip; MKZMPCPathFollower.jl:34): one problem per
16-lane DPP row, the shared interior-point code of kmpc_ipm.h with per-row control flow.  Dispatched for batches of >= 1024 problems
(kmpc_config.kernel_variant 0); kernel_variant 2 keeps the one-wave-per-problem kernel at every batch size and is what it is compared with."""
import os

import numpy as np
import pytest

import certify as CT
from mkz_mpc_path_follower_amd.synthetic import make_batch, straight_line_case

pytestmark = pytest.mark.gpu


def _solve(d, dtype=None, **kw):
    import torch
    from mkz_mpc_path_follower_amd import BatchMPC
    s = BatchMPC(N=8, dtype=dtype or torch.float64, **kw)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True, want_X=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items()}


@pytest.mark.parametrize("B", [1024, 4099, 8192])
def test_quad_matches_the_oracle_and_the_one_wave_kernel(oracle, B):
    """batch sizes incl. one that is not a multiple of four (the last wave repeats the last problem in its spare rows): every problem
    Optimal, fp64 parity tolerances against the CPU checker, and against the one-wave kernel (same algorithm, other summation trees)
    the same statuses, costs to 1e-9 and iteration counts equal on >= 99.5 % of the problems"""
    O = oracle
    d = make_batch(B, 8, cfg_id=2, seed=4242 + B)
    q = _solve(d)                       # >= 1024 problems: four per wave
    w = _solve(d, kernel_variant=2)     # one wave per problem
    assert (q["status"] == 0).all() and (w["status"] == 0).all(), (np.bincount(q["status"]), np.bincount(w["status"]))
    rel = np.abs(q["cost"] - w["cost"]) / np.maximum(1.0, np.abs(w["cost"]))
    assert rel.max() <= 1e-9, rel.max()
    same = q["iters"] == w["iters"]
    assert same.mean() >= 0.995, (~same).sum()
    # same path -> same iterate; the few that stop an iteration apart (a threshold of the iteration's rules straddled by rounding) agree as far as the
    # tolerance pins the inputs: the cost is flat in the last acceleration (only the rate cost ties it), which then differs by up to ~1e-5
    dU, dX = np.abs(q["U"] - w["U"]).reshape(B, -1).max(1), np.abs(q["X"] - w["X"]).reshape(B, -1).max(1)
    assert dU[same].max() <= 1e-6 and dX[same].max() <= 1e-6
    assert (~same).sum() == 0 or (dU[~same].max() <= 1e-4 and dX[~same].max() <= 1e-4)
    assert q["viol"].max() <= 1e-8 + 1e-12
    ro = O.solve_condensed_batch(O.params(8), d["z0"], d["ref"], d["v_target"], d["u_prev"], nthreads=8)
    relo = np.abs(q["cost"] - ro["cost"]) / np.maximum(1.0, np.abs(ro["cost"]))
    assert relo.max() <= 1e-6 and np.abs(q["u0"] - ro["U"][:, 0]).max() <= 1e-6


def test_quad_outputs_are_certified_kkt_points(oracle):
    """every one of 4096 returned U is a certified KKT point of the reference's NLP (independent NNLS multipliers, tests/certify.py)"""
    O = oracle
    d = make_batch(4096, 8, cfg_id=2)
    q = _solve(d)
    assert (q["status"] == 0).all()
    c = CT.certify_batch(O, O.params(8), d, q["U"])
    assert c["scaled_stationarity"].max() <= 1e-6 and c["scaled_complementarity"].max() <= 1e-6 and c["violation"].max() <= 1e-8 + 1e-12
    assert c["ref_scaled_stationarity"].max() <= 1e-7 and c["ref_scaled_complementarity"].max() <= 1e-7


def test_quad_known_answers_and_row_independence():
    """the module-load problem (J* = 15738.467, acc_1 = 0.15) in every row of many waves next to other problems: a row's result does not
    depend on its neighbours, and rows with very different iteration counts (an on-path problem next to a 3 m offset) coexist"""
    k = straight_line_case(8)
    d = make_batch(2048, 8, cfg_id=7)
    for key in ("z0", "ref", "v_target", "u_prev"):
        d[key][::5] = k[key][0]                       # every fifth problem: all four row positions occur
    q = _solve(d)
    assert (q["status"] == 0).all()
    assert np.abs(q["cost"][::5] - 15738.467188588813).max() <= 1e-6 and np.abs(q["u0"][::5, 0] - 0.15).max() <= 1e-7 and np.abs(q["u0"][::5, 1]).max() <= 1e-9
    assert np.ptp(q["cost"][::5]) <= 1e-9 * 15738.0      # (identical inputs, any row, any neighbours)
    w = _solve(d, kernel_variant=2)
    assert np.abs(q["cost"] - w["cost"]).max() <= 1e-9 * np.maximum(1.0, np.abs(w["cost"])).max()


def test_quad_status_containment():
    """infeasible (v0 outside the speed bounds, Q5) and NaN problems sit in rows next to healthy ones: statuses per problem, finite bounded
    commands everywhere, the healthy rows unaffected"""
    d = make_batch(1024, 8, cfg_id=9)
    ref = _solve(d)
    bad = d["z0"].copy()
    bad[3::7, 3] = 25.0                                # infeasible speed
    bad[5::11, 0] = np.nan                             # NaN pose
    dd = dict(d, z0=bad)
    q = _solve(dd)
    inf_, nan_ = np.zeros(1024, bool), np.zeros(1024, bool)
    inf_[3::7] = True; nan_[5::11] = True
    assert (q["status"][inf_ & ~nan_] == 2).all()
    assert (q["status"][nan_] != 0).all()
    ok = ~(inf_ | nan_)
    assert (q["status"][ok] == 0).all() and np.abs(q["cost"][ok] - ref["cost"][ok]).max() <= 1e-9 * np.abs(ref["cost"][ok]).max()
    assert np.isfinite(q["u0"]).all() and np.abs(q["u0"][:, 0]).max() <= 1.0 + 1e-8 and np.abs(q["u0"][:, 1]).max() <= 0.5 + 1e-8


def test_quad_fp32_and_warm_start(oracle):
    import torch
    d = make_batch(4096, 8, cfg_id=2)
    q32 = _solve(d, dtype=torch.float32)
    q64 = _solve(d)
    assert (q32["status"] == 0).all()
    rel = np.abs(q32["cost"] - q64["cost"]) / np.maximum(1.0, np.abs(q64["cost"]))
    assert np.percentile(rel, 99) <= 1e-3 and q32["viol"].max() <= 1e-4
    # warm start from the solution: far fewer iterations, same minimum
    from mkz_mpc_path_follower_amd import BatchMPC
    s = BatchMPC(N=8)
    o = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], want_U=True)
    it_cold = o["iters"].float().mean().item()
    warm = o["U"].clone()
    o2 = s.solve(d["z0"], d["ref"], d["v_target"], d["u_prev"], warm_U=warm, warm=True)
    torch.cuda.synchronize()
    assert (o2["status"] == 0).all().item() and o2["iters"].float().mean().item() < 0.7 * it_cold
    assert (torch.abs(o2["cost"] - o["cost"]) <= 1e-6 * torch.clamp(o["cost"].abs(), min=1.0)).all().item()


def test_quad_full_size_properties():
    """262 144 problems (the large-batch size of the BASELINE configs at the reference's horizon): all Optimal, same statuses and costs
    (1e-9) as the one-wave kernel, mirror symmetry of the model (y, psi -> -y, -psi flips the steering and keeps the cost)"""
    B = 262144
    d = make_batch(B, 8, cfg_id=3)
    q = _solve(d)
    w = _solve(d, kernel_variant=2)
    assert (q["status"] == 0).all() and (w["status"] == 0).all()
    rel = np.abs(q["cost"] - w["cost"]) / np.maximum(1.0, np.abs(w["cost"]))
    assert (rel <= 1e-9).mean() >= 0.9999 and rel.max() <= 5e-2      # (a handful of non-convex paths may split on rounding: other local minima)
    m = dict(z0=d["z0"] * np.array([1, -1, -1, 1.0]), ref=d["ref"] * np.array([1, -1, -1.0]), v_target=d["v_target"], u_prev=d["u_prev"] * np.array([1, -1.0]))
    qm = _solve(m)
    relm = np.abs(qm["cost"] - q["cost"]) / np.maximum(1.0, np.abs(q["cost"]))
    assert (relm <= 1e-9).mean() >= 0.999 and np.abs(qm["u0"][relm <= 1e-9] * np.array([1, -1.0]) - q["u0"][relm <= 1e-9]).max() <= 1e-6
