"""bench.py -- MPC solves/s (batch, N=20 kinematic bicycle) on N GPUs of one node.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (kmpc_solve_batch) over one batch of synthetic (x0, path
segment) problems already resident in HBM, followed -- for N > 1 -- by the all-gather of the
(accel, steer) blocks (RCCL; asynchronous and double-buffered, so the exchange of batch k overlaps the
solve of batch k+1 -- every gather completes inside the timed region).  Workload at every N: BASELINE.json configs[1], batch = 4096 problems per
GPU, horizon 20, fp64 (weak scaling: the batch is sharded, per-GPU work is fixed).  A launch of a few thousand problems takes as long as its
slowest problem, so its time depends on the draw: every rank cycles through the SAME K_DRAWS seeded batches (rank r solves draw (step + r) % K),
so `value` is a mean over draws and every rank does the same total work whatever N is.
`--batch 262144 --gpus 8` is BASELINE.json configs[3] (2 097 152 problems over 8 GPUs) and labels itself so.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = fp64 MFMA (v_mfma_f64_16x16x4) spec rate, SURVEY.md section 7.2
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32-input MFMA peak


def algorithmic_bytes_per_solve(N, es):
    # in z0(4) + ref 3(N+1) + v_t(1) + u_prev(2); out u0(2) + int32 status
    return (3 * N + 10) * es + 2 * es + 4


def algorithmic_flops_per_iteration(N):
    # linearise 150N + sensitivities 32N^2 + structured condensing sum 16k^2 + one IPM step
    n, m = 2 * N, 10 * N - 4
    f_cond = sum(16 * k * k for k in range(1, N + 1))
    return 150 * N + 32 * N * N + f_cond + (n ** 3 / 3.0 + 4 * n * n + 10 * m)


def executed_flops_per_iteration(N, kernel=""):
    # what the kernels execute since round 2: condensing by the O(N^2) adjoint recursion -- per stage and live column 5 FMAs to recover G_s,
    # 4 for the two Hessian rows, 2 for the second-order row, 5 for A^T p, 10 for W G (52 flops), columns 0..2s+1 live at stage s -- plus the
    # suffix scans of the terminal sensitivities (~40 N); linearisation and the IPM step as in SURVEY.md 8(d).  This is the numerator of
    # `roofline.achieved` / `frac` (round 3); the SURVEY formula (above) is kept beside it as `frac_survey_model` so that rounds compare.
    # Round 3: the recursion runs as two halves at the same time (stages >= M in the column's own lane, stages < M of the columns < 2M in
    # spare lanes; same 52 flops per live column and stage) and is stitched by one rank-4 product: 8 flops per entry of the 2M x 2M lower triangle.
    n, m = 2 * N, 10 * N - 4
    M = min(N // 2, (64 - n) // 2) if N <= 28 else N // 2   # kmpc_fast.hip / kmpc_wide.hip: MSPLIT
    if "quad" in kernel or "frenet" in kernel:
        M = 0                                                  # the four-per-wave kernel and the Frenet functor do not split the recursion (no stitch product)
    return 150 * N + 52 * N * (N + 1) + 40 * N + 8 * M * (2 * M + 1) + (n ** 3 / 3.0 + 4 * n * n + 10 * m)


def counter_profile(kernel):
    """committed counter passes of one kernel (profiles/r*_pmc_*.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE and the SQ passes, each with the
    command that produced it), newest round first, or None.  PMC counters need their own rocprofv3 passes (tools/pmc_quick.sh); a bench run does
    not collect them -- the figures are the committed ones of the same kernel build and the JSON line says so."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            with open(f) as fh:
                j = json.load(fh)
            k = j.get("kernels", {}).get(kernel)
            if k:
                return dict(k, file="profiles/" + os.path.basename(f))
        except Exception:
            continue
    return None


def measured_traffic_bytes(kernel="kmpc_solve_fast_kernel<double,20>"):
    """(HBM bytes per dispatch, source) of the headline kernel from the committed PMC passes, or (None, None)"""
    k = counter_profile(kernel)
    if k and k.get("hbm_bytes_per_dispatch"):
        return float(k["hbm_bytes_per_dispatch"]), "%s (B = %s): committed rocprofv3 --pmc passes, not measured in this run" % (k["file"], k.get("batch"))
    for name in ("r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                j = json.load(f)
                return float(j["runs"][j["current"]]["hbm_bytes_per_dispatch"]), "profiles/%s (%s): committed rocprofv3 --pmc passes, not measured in this run" % (name, j["current"])
        except Exception:
            continue
    return None, None


def time_launches(solver, din, steps, warmup):
    """mean HIP-event duration (ms) of `steps` launches over resident inputs, after `warmup` untimed ones; -> (ms, last outputs)"""
    out = None
    for _ in range(warmup):
        out = solver.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for e0, e1 in ev:
        e0.record()
        out = solver.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=out)
        e1.record()
    torch.cuda.synchronize()
    return float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])), out


def multi_seed(solver, N, B, tdt, dev, seeds=8, steps=10, warmup=2):
    """The launch time of a few thousand problems is set by its slowest problems, so it depends on the draw: the same workload
    on `seeds` OTHER seeded batches (the headline batch is not among them)."""
    from mkz_mpc_path_follower_amd.synthetic import make_batch
    rates, its, opt = [], [], []
    for s in range(seeds):
        d = make_batch(B, N, cfg_id=2, seed=20180620 + 7919 * (s + 1))
        din = {k: torch.as_tensor(d[k], dtype=tdt, device=dev) for k in ("z0", "ref", "v_target", "u_prev")}
        ms, out = time_launches(solver, din, steps, warmup)
        rates.append(B / (ms * 1e-3)); its.append(out["iters"].float().mean().item())
        opt.append(int((out["status"] == 0).sum().item()) / B)
    return {"seeds": seeds, "launches_per_seed": steps, "mean": float(np.mean(rates)), "min": float(np.min(rates)), "max": float(np.max(rates)),
            "per_seed": [float(r) for r in rates], "unit": "solves/s", "mean_iterations": float(np.mean(its)),
            "optimal_fraction": float(np.mean(opt)), "optimal_fraction_min": float(np.min(opt)), "all_optimal": bool(min(opt) == 1.0)}


def two_in_flight(N, B, tdt, din, local, steps=40, warmup=6):
    """Not the headline: the same 4096-problem batches with TWO launches in flight (two handles, two streams, alternating), the way a
    stream of independent batches would be fed.  A single launch of this size takes as long as its slowest problem while most of the
    chip idles in the tail; the next batch's problems fill that tail.  Whole-job solves/s over `steps` batches."""
    from mkz_mpc_path_follower_amd import BatchMPC
    sv = [BatchMPC(N=N, dtype=tdt, device=local) for _ in range(2)]
    st = [torch.cuda.Stream(device=local) for _ in range(2)]
    outs = [None, None]
    torch.cuda.synchronize()
    def run(k):
        t0 = time.perf_counter()
        for i in range(k):
            j = i & 1
            with torch.cuda.stream(st[j]):   # stream-ordered per handle: launch i waits for launch i-2 (same stream, same buffers)
                outs[j] = sv[j].solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=outs[j])
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(warmup)
    el = run(steps)
    ok = all(int((o["status"] == 0).sum().item()) == B for o in outs)
    return {"batches_in_flight": 2, "steps": steps, "ms_per_batch": el / steps * 1e3, "solves_per_s": B * steps / el, "all_optimal": ok,
            "note": "extra key, not the headline: consecutive independent batches overlapped on two streams"}


def other_config(N, B, dtype, cfg_id, dev, local, steps=5, warmup=2, packed=False):
    """untimed-headline extra key: one of the other BASELINE configs on this GPU (kernel time by HIP events over resident inputs);
    packed = True: the same problems through kmpc_solve_batch_packed (ABI v8: one 64-B-aligned record per problem)"""
    from mkz_mpc_path_follower_amd import BatchMPC
    from mkz_mpc_path_follower_amd.synthetic import make_batch
    tdt = torch.float64 if dtype == "f64" else torch.float32
    d = make_batch(B, N, cfg_id=cfg_id)
    din = {k: torch.as_tensor(d[k], dtype=tdt, device=dev) for k in ("z0", "ref", "v_target", "u_prev")}
    solver = BatchMPC(N=N, dtype=tdt, device=local)
    if packed:
        rec = solver.pack(din["z0"], din["ref"], din["v_target"], din["u_prev"])

        class _P:   # time_launches calls solver.solve(z0, ref, v_target, u_prev, out=...)
            @staticmethod
            def solve(z0, ref, vt, up, out=None):
                return solver.solve_packed(rec, out=out)
        ms, out = time_launches(_P, din, steps, warmup)
    else:
        ms, out = time_launches(solver, din, steps, warmup)
    iters = out["iters"].float().mean().item()
    peak = FP64_PEAK_TFLOPS if dtype == "f64" else FP32_PEAK_TFLOPS
    tf = executed_flops_per_iteration(N, kernel_name(N, dtype, B)) * iters * B / (ms * 1e-3) / 1e12
    tf_model = algorithmic_flops_per_iteration(N) * iters * B / (ms * 1e-3) / 1e12
    es = 8 if dtype == "f64" else 4
    r = {"workload": "batch=%d, N=%d, %s, 1 GPU, seeded synthetic (cfg_id %d)%s" % (B, N, dtype, cfg_id, ", packed records (kmpc_solve_batch_packed)" if packed else ""), "solves_per_s": B / (ms * 1e-3),
         "kernel": kernel_name(N, dtype, B), "kernel_ms": ms, "launches": steps, "mean_iterations": iters, "max_iterations": int(out["iters"].max().item()),
         "optimal_fraction": float((out["status"] == 0).float().mean().item()),
         "achieved_tflops": tf, "peak_tflops": peak, "frac_of_peak": tf / peak, "frac_survey_model": tf_model / peak,
         "flops_note": "frac_of_peak prices the flops the kernel executes (O(N^2) adjoint condensing); frac_survey_model the SURVEY 8(d) formula",
         "hbm_gbs_algorithmic": algorithmic_bytes_per_solve(N, es) * B / (ms * 1e-3) / 1e9}
    k = counter_profile(r["kernel"] + (" packed records" if packed else ""))
    if k and k.get("hbm_bytes_per_dispatch") and k.get("batch") == B:
        r["hbm_bytes_per_dispatch_counters"] = float(k["hbm_bytes_per_dispatch"])
        r["hbm_traffic_over_algorithmic"] = float(k["hbm_bytes_per_dispatch"]) / (algorithmic_bytes_per_solve(N, es) * B)
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same kernel at the same batch (committed; corrected as MI355X_MICROARCH.md prescribes)
        r["hbm_gbs_counters"] = float(k["hbm_bytes_per_dispatch"]) / (ms * 1e-3) / 1e9
        r["hbm_frac_of_peak_counters"] = r["hbm_gbs_counters"] / HBM_PEAK_GBS
        r["hbm_counters_source"] = "%s: committed rocprofv3 --pmc passes (bytes per dispatch), divided by this run's kernel time" % k["file"]
    return r


def closed_loop_latency(device, steps=150):
    """SURVEY.md 8(d) config 1 continued: the reference's own horizon (N = 8) in a 10 Hz receding-horizon loop against the
    simulated plant on the recorded path (tests/golden/path1_decimated.npz), warm-started, one vehicle: per-solve latency."""
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
    f = os.path.join(ROOT, "tests", "golden", "path1_decimated.npz")
    if not os.path.exists(f):
        return None
    d = np.load(f)
    grt = GPSRefTrajectory(arrays=dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"]), traj_horizon=8, traj_dt=0.2, device=device)
    tr = grt.get_global_trajectory_reference()
    sim = VehicleSimulator(1, X0=tr[0, 4], Y0=tr[0, 5], Psi0=tr[0, 3], device=device)
    loop = ClosedLoop(grt, sim, N=8, target_vel=8.0)
    lat, its, worst = [], [], 0
    for k in range(steps):
        o = loop.step(time_solve=True)
        lat.append(o["solve_s"]); its.append(int(o["iters"][0].item())); worst = max(worst, int(o["status"][0].item()))
    w = np.array(lat[10:]) * 1e6
    return {"workload": "N=8, B=1, warm start, 10 Hz closed loop on path1 at 8 m/s, %d steps" % steps, "p50_latency_us": float(np.percentile(w, 50)),
            "p99_latency_us": float(np.percentile(w, 99)), "first_solve_us": float(lat[0] * 1e6), "mean_iterations": float(np.mean(its[10:])),
            "worst_status": worst}


def dropin_host_path_latency(device, steps=400):
    """The drop-in B = 1 path exactly as the reference's node drives it (mpc_cmd_pub.jl:115-141): HOST arrays through the module API mirror
    (KinematicMPC.update_init_cond / update_reference / solve_model / update_current_input / get_solver_results -> kmpc_solve_batch_host), N = 8, warm-started,
    a car accelerating along a straight reference.  Wall time of the whole control step, Python included.  Round 4: the host entry point runs small batches on
    pinned, device-mapped host memory (no copy launches) and waits on a completion counter in that memory (174 -> 57 us at N = 8)."""
    from mkz_mpc_path_follower_amd import KinematicMPC
    k = KinematicMPC(N=8, device=device)
    k.update_cost(9, 9, 10, 0, 100, 1000, 0, 0)     # mpc_cmd_pub.jl:49
    N, lat, its, worst, x = 8, [], [], 0, 0.0
    for i in range(steps):
        v = min(15.0, 0.1 * i)
        xr = x + v * 0.2 * np.arange(N + 1) + 0.3
        t0 = time.perf_counter()
        k.update_init_cond(x, 0.05 * np.sin(0.05 * i), 0.01, v)
        k.update_reference(xr, np.zeros(N + 1), np.zeros(N + 1), v)
        a, dsteer, st = k.solve_model()
        k.update_current_input(dsteer, a)
        k.get_solver_results()
        lat.append(time.perf_counter() - t0); its.append(k.iters); worst = max(worst, 0 if st == "Optimal" else 1)
        x += v * 0.1
    k.close()
    w = np.array(lat[50:]) * 1e6
    return {"workload": "N=8, B=1, host arrays through the module-API mirror (kmpc_solve_batch_host), warm start, %d control steps" % steps,
            "p50_step_us": float(np.percentile(w, 50)), "p99_step_us": float(np.percentile(w, 99)), "mean_iterations": float(np.mean(its[50:])), "worst_status": worst,
            "note": "whole control step incl. Python; a near-empty kernel costs 18 us of launch + synchronisation on this stack and one iteration of a lone fp64 wave "
                    "7-9 us (tools/latency_floor.py, DESIGN.md section 7)"}


def closed_loop_fleet(device, B=4096, steps=40):
    """The same 10 Hz loop for a FLEET: B vehicles spread along the recorded path, each with its own warm start -- waypoint look-ahead, the N = 8
    solve (the reference's horizon) and the plant, all on the device.  Extra key: vehicle control steps per second and the solve's share."""
    from mkz_mpc_path_follower_amd.ref_traj import GPSRefTrajectory
    from mkz_mpc_path_follower_amd.vehicle_sim import VehicleSimulator
    from mkz_mpc_path_follower_amd.closed_loop import ClosedLoop
    f = os.path.join(ROOT, "tests", "golden", "path1_decimated.npz")
    if not os.path.exists(f):
        return None
    d = np.load(f)
    grt = GPSRefTrajectory(arrays=dict(t=d["t"], lat=d["lat"], lon=d["lon"], psi=d["psi"]), traj_horizon=8, traj_dt=0.2, device=device)
    tr = grt.get_global_trajectory_reference()
    idx = np.linspace(0, int(0.6 * (len(tr) - 1)), B).astype(int)     # start poses along the first 60 % of the path
    rng = np.random.default_rng(20180620)
    off = rng.normal(0.0, 0.3, B)                                      # lateral offsets
    X0 = tr[idx, 4] - off * np.sin(tr[idx, 3]); Y0 = tr[idx, 5] + off * np.cos(tr[idx, 3])
    sim = VehicleSimulator(B, X0=X0, Y0=Y0, Psi0=tr[idx, 3], device=device)
    loop = ClosedLoop(grt, sim, N=8, target_vel=8.0)
    solve_s, its, worst = [], [], 0
    for k in range(5):
        loop.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):          # free-running: nothing waits for the host inside the loop
        o = loop.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for k in range(10):             # the solve alone, bracketed by device synchronisations
        o = loop.step(time_solve=True)
        solve_s.append(o["solve_s"]); its.append(float(o["iters"].float().mean().item())); worst = max(worst, int(o["status"].max().item()))
    return {"workload": "N=8, %d vehicles on path1 at 8 m/s, warm start, %d loop steps (waypoints + solve + 0.1 s of plant per step)" % (B, steps),
            "vehicle_steps_per_s": B * steps / el, "ms_per_loop_step": el / steps * 1e3, "solve_ms_p50": float(np.percentile(solve_s, 50) * 1e3),
            "mean_iterations": float(np.mean(its)), "worst_status": worst}


def parity_sample(N, d, out, ro, tol=1e-6):
    """GPU vs the CPU port on the sampled problems.  The NLP is non-convex: on a few problems per thousand whose Hessian is
    indefinite along the way, rounding-level differences between the two implementations end in different local minima; those
    are counted separately and both solutions are checked to be KKT points (least-squares multipliers on the active set)."""
    from oracle import oracle as O
    S = ro["cost"].shape[0]
    gc = out["cost"][:S].double().cpu().numpy()
    rel = np.abs(gc - ro["cost"]) / np.maximum(1.0, np.abs(ro["cost"]))
    within = rel <= tol
    info = {"n": int(S), "tol": tol, "within_tol": int(within.sum()), "max_rel_cost_err_within_tol": float(rel[within].max()) if within.any() else None,
            "max_viol": float(out["viol"].max().item()), "other_local_minimum": 0, "unexplained": 0}
    bad = np.where(~within)[0]
    if len(bad) and "U" in out:
        p = O.params(N)
        gU = out["U"][:S].double().cpu().numpy()
        for b in bad[:32]:
            q = O.problem(p, d["z0"][b], d["ref"][b], d["v_target"][b], d["u_prev"][b])
            A, bb = O.ineq(p, q)
            kkt = []
            for U in (gU[b], ro["U"][b]):
                gr = O.grad(p, q, U)
                act = (bb - A @ U.ravel()) < 1e-6
                lam = np.linalg.lstsq(A[act].T, -gr, rcond=None)[0] if act.any() else np.zeros(0)
                r_ = gr + (A[act].T @ lam if act.any() else 0.0)
                kkt.append(np.abs(r_).max() <= 1e-4 * max(1.0, np.abs(gr).max()) and (lam.min() if act.any() else 0.0) >= -1e-4 * max(1.0, np.abs(gr).max()))
            info["other_local_minimum" if all(kkt) else "unexplained"] += 1
        info["max_rel_cost_err_other_minimum"] = float(rel[bad].max())
    return info


def kernel_name(N, dtype, B=4096):
    """the kernel kmpc_solve_batch dispatches (csrc/kmpc_api.hip: solve_dev; kmpc_fast_available / kmpc_wide_available / launch_fast_n)"""
    t = "double" if dtype == "f64" else "float"
    if N == 8 and B >= 1024:               # KMPC_QUAD_MIN_BATCH: four problems per wave at the reference's own horizon (kmpc_quad.hip)
        return "kmpc_solve_quad_kernel<%s>" % t
    if N in (8, 12, 16, 20, 24, 28):       # kmpc_fast_available: compile-time horizons with 2N + 1 <= 64
        if N <= 12 and B > 2048:
            return "kmpc_solve_fast_dense_kernel<%s,%d>" % (t, N)
        return "kmpc_solve_fast_kernel<%s,%d>" % (t, N)
    if N in (32, 36, 40, 44, 48, 50):      # kmpc_wide_available: four waves per problem, both element types
        return "kmpc_solve_wide_kernel<%s,%d>" % (t, N)
    return "kmpc_solve_kernel<%s>" % t      # generic kernel (runtime horizon)


def host_cores():
    """threads worth starting: the affinity mask, capped by the cgroup CPU quota when there is one (a 1-GPU box exposes 256 logical
    CPUs but grants a share of them)"""
    n = len(os.sched_getaffinity(0))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            if f.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(np.ceil(float(t[0]) / float(t[1])))))
            else:
                q = float(t[0]); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(np.ceil(q / per))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(N, d, budget_s=15.0):
    """oracle/ CPU port (same algorithm, scalar C, pthreads over problems) on a bounded sample."""
    from oracle import oracle as O
    cores = host_cores()
    p = O.params(N)
    pilot = min(4 * cores, d["z0"].shape[0])
    t = time.perf_counter()
    O.solve_condensed_batch(p, d["z0"][:pilot], d["ref"][:pilot], d["v_target"][:pilot], d["u_prev"][:pilot], nthreads=cores)
    rate = pilot / max(time.perf_counter() - t, 1e-9)
    S = int(min(d["z0"].shape[0], max(pilot, rate * budget_s)))
    el, reps = 0.0, 0
    while el < 3.0 and reps < 64:  # the sample is a fraction of a second on a many-core host: repeat it for a stable figure
        t = time.perf_counter()
        r = O.solve_condensed_batch(p, d["z0"][:S], d["ref"][:S], d["v_target"][:S], d["u_prev"][:S], nthreads=cores)
        el += time.perf_counter() - t
        reps += 1
    return dict(value=S * reps / el, unit="solves/s", cores=cores, kind="port",
                sample="first %d of the %d-problem GPU batch x %d repeats, oracle/kmpc_condensed.c (scalar fp64 C, %d pthreads), "
                       "mean %.1f iterations" % (S, d["z0"].shape[0], reps, cores, float(r["iters"].mean()))), r


def cpu_baseline_config1(n_cold=1000, n_warm=1000):
    """BASELINE.md section 3 run C1: the CPU port (oracle/kmpc_condensed.c), ONE thread, BASELINE configs[0] -- the reference's module-load
    problem (N = 8, z0 = 0, straight reference at 15 m/s, u_prev = 0, node weights; MKZMPCPathFollower.jl:36-39,110-113,127) solved cold
    n_cold times, and its 10 Hz receding-horizon continuation (state advanced by the model's own Euler step over dt_control with the first
    input, reference re-anchored at the car as the time-mode waypoint helper does, warm start from the previous primal solution as JuMP
    keeps it, Q9).  Per-solve wall time around the C call; the reference's only figure is its 0.1 s cap (MKZMPCPathFollower.jl:29)."""
    from oracle import oracle as O
    N = 8
    p = O.params(N)
    ref = np.zeros((N + 1, 3)); ref[:, 0] = 15.0 * 0.2 * np.arange(N + 1)
    q = O.problem(p, np.zeros(4), ref, 15.0, (0.0, 0.0))
    o = O.opts()
    cold, r = [], None
    for _ in range(n_cold + 20):
        t = time.perf_counter()
        r = O.solve_condensed(p, q, o)
        cold.append(time.perf_counter() - t)
    cold = np.array(cold[20:]) * 1e6
    cold_cost, cold_iters, cold_status = r["cost"], r["iters"], r["status"]
    ow = O.opts(warm=1)
    z = np.zeros(4); up = np.zeros(2); U = r["U"].copy()
    rr = p.L_b / (p.L_a + p.L_b)
    warm, its, worst = [], [], 0
    for k in range(n_warm):
        a, d = U[0]
        beta = np.arctan(rr * np.tan(d))                                   # MKZMPCPathFollower.jl:115-122 over dt_control
        z = z + p.dt_control * np.array([z[3] * np.cos(z[2] + beta), z[3] * np.sin(z[2] + beta), z[3] / p.L_b * np.sin(beta), a])
        up = np.array([a, d])
        refk = np.zeros((N + 1, 3)); refk[:, 0] = z[0] + 15.0 * 0.2 * np.arange(N + 1)
        qk = O.problem(p, z, refk, 15.0, up)
        t = time.perf_counter()
        r = O.solve_condensed(p, qk, ow, U0=U)
        warm.append(time.perf_counter() - t)
        U = r["U"]; its.append(r["iters"]); worst = max(worst, r["status"])
    warm = np.array(warm) * 1e6
    return {"workload": "BASELINE configs[0]: single N=8 problem (module-load problem of MKZMPCPathFollower.jl), CPU port, 1 thread",
            "kind": "port", "cores": 1, "cold": {"solves": n_cold, "p50_us": float(np.percentile(cold, 50)), "p99_us": float(np.percentile(cold, 99)),
                                                   "cost": cold_cost, "iterations": cold_iters, "status": cold_status},
            "warm_10hz": {"solves": n_warm, "p50_us": float(np.percentile(warm, 50)), "p99_us": float(np.percentile(warm, 99)),
                          "mean_iterations": float(np.mean(its)), "worst_status": worst, "final_speed": float(z[3])},
            "reference_cap_us": 1e5, "note": "the reference caps one Ipopt solve at 0.1 s CPU (max_cpu_time = dt_control); its own timing is unavailable (no julia)"}


def gpu_config1_latency(local, n=300):
    """the same module-load problem on the GPU, B = 1, cold, device-resident inputs: host clock around launch + sync"""
    from mkz_mpc_path_follower_amd import BatchMPC
    from mkz_mpc_path_follower_amd.synthetic import straight_line_case
    d = straight_line_case(8)
    sv = BatchMPC(N=8, device=local)
    din = {k: torch.as_tensor(d[k], dtype=torch.float64, device=sv.device) for k in ("z0", "ref", "v_target", "u_prev")}
    o, lat = None, []
    for i in range(n + 20):
        torch.cuda.synchronize()
        t = time.perf_counter()
        o = sv.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], out=o)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t)
    w = np.array(lat[20:]) * 1e6
    return {"workload": "BASELINE configs[0] on the GPU: module-load problem, N=8, B=1, cold, resident inputs", "p50_us": float(np.percentile(w, 50)),
            "p99_us": float(np.percentile(w, 99)), "cost": float(o["cost"][0].item()), "iterations": int(o["iters"][0].item()), "status": int(o["status"][0].item())}


K_DRAWS = 4   # seeded batches every rank cycles through (draw 0 is the batch rounds 1-2 quoted alone)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="problems per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="skip the multi-seed and other-config extra keys")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the multi-rank control flow on a one-GPU box)")
    ap.add_argument("--clock-ramp", type=int, default=100, help="untimed launches BEFORE the W warm-up steps that bring the GPU's clocks up from idle (0 = none)")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the asynchronous RCCL all-gather of the N > 1 loop also at --gpus 1 (1-rank nccl group): prices the collective's overhead on one GPU "
                         "and puts the RCCL path under test before multi-GPU hardware shows up (tests/test_bench_cli.py)")
    a = ap.parse_args()
    # dmabuf IPC only on this pool (RCCL / cross-process tensors fail with hipIpcGetMemHandle otherwise): already exported on the boxes; set before the
    # first HIP call in case a launcher dropped it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torch.distributed.run child) BEFORE any
        # GPU call in this process -- a process that has touched the GPU must never be replaced or forked from
        from mkz_mpc_path_follower_amd.dist import spawn_ranks
        if a.backend == "nccl" and torch.cuda.device_count() < a.gpus:  # device_count() does not initialise the GPU on this image
            sys.exit("bench.py: --gpus %d but %d GPU(s) visible (use --backend gloo to rehearse the multi-rank control flow on fewer)"
                     % (a.gpus, torch.cuda.device_count()))
        sys.exit(spawn_ranks(os.path.abspath(__file__), a.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)" % (a.gpus, world, a.gpus))
    if a.backend == "gloo":
        local = local % torch.cuda.device_count()  # rehearsal: the ranks share the box's GPU(s)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    forced = a.force_collective and world == 1
    if forced:   # a 1-rank group of its own: rendezvous on a free local port
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == a.gpus

    from mkz_mpc_path_follower_amd import BatchMPC
    from mkz_mpc_path_follower_amd.dist import SolutionGather
    from mkz_mpc_path_follower_amd.synthetic import make_batch

    N, Bl = a.horizon, a.batch
    tdt = torch.float64 if a.dtype == "f64" else torch.float32
    es = 8 if a.dtype == "f64" else 4
    B = Bl * world
    # every rank generates the same K seeded draws locally (no scatter) and solves draw (step + rank) % K at each step: equal total work on
    # every rank and at every N, different problems in flight on different GPUs at any one time
    K = K_DRAWS
    draws = [make_batch(Bl, N, cfg_id=2, seed=20180620 + 2 + 7919 * j) for j in range(K)]
    d = draws[0]
    dins = [{k: torch.as_tensor(dj[k], dtype=tdt, device=dev) for k in ("z0", "ref", "v_target", "u_prev")} for dj in draws]
    din = dins[0]
    solver = BatchMPC(N=N, dtype=tdt, device=local)
    # N > 1: the all-gather of batch k's (accel, steer) block runs asynchronously on the collective's stream while later batches are
    # solved; K output slots alternate (a rank may run up to K steps ahead of the slowest one -- over K steps every rank has done the same
    # work), and a slot's gather is waited for before a solve may overwrite the buffer it reads.
    # Every step's gather completes inside the timed region (final waits + synchronize below).
    gather = SolutionGather(B, slots=K, force_collective=forced)
    outs = [None] * K

    # (host-side set-up of the timed loop comes BEFORE the warm-up, so that nothing but the synchronisation sits between the last warm-up launch and t0)
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(a.steps)]
    wait_host = 0.0
    # The GPU has idled (and clocked down) while the host generated the inputs: bring it to its sustained state before the W warm-up steps with
    # untimed launches of the same solve into scratch outputs (reported as config.clock_ramp_launches; --clock-ramp 0 turns it off).  With W = 5 and
    # nothing else, the first timed kernels ran 3 % slower than the same kernels 30 ms later (0.478 vs 0.461 ms).
    ramp_out = None
    for i in range(a.clock_ramp):
        dj = dins[i % K]
        ramp_out = solver.solve(dj["z0"], dj["ref"], dj["v_target"], dj["u_prev"], out=ramp_out)
    for i in range(a.warmup):
        s = i % K
        gather.wait(s)
        dj = dins[(i + rank) % K]
        outs[s] = solver.solve(dj["z0"], dj["ref"], dj["v_target"], dj["u_prev"], out=outs[s])
        gather.submit(s, outs[s]["u0"])
    for s in range(K):
        gather.wait(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        s = i % K
        ev[i][0].record()
        tw = time.perf_counter()
        gather.wait(s)
        wait_host += time.perf_counter() - tw
        dj = dins[(a.warmup + i + rank) % K]
        ev[i][1].record()   # same stream the kernel is launched on (torch's current stream is handed to the C ABI)
        outs[s] = solver.solve(dj["z0"], dj["ref"], dj["v_target"], dj["u_prev"], out=outs[s])
        ev[i][2].record()
        gather.submit(s, outs[s]["u0"])
    for s in range(K):
        gather.wait(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # untimed: K more steps through the same loop, every gathered [B, 2] block compared with the blocks the ranks solved (this rank's own block exactly;
    # with N > 1 also the block every other rank reports for the same draw -- all ranks solve the same K draws, rotated, and the kernels are deterministic)
    gather_checked, gather_ok = 0, True
    if world > 1 or forced:
        own = {}
        for i in range(K):
            s = i % K
            dj = dins[(i + rank) % K]
            outs[s] = solver.solve(dj["z0"], dj["ref"], dj["v_target"], dj["u_prev"], out=outs[s])
            gather.submit(s, outs[s]["u0"])
            own[(i + rank) % K] = outs[s]["u0"].clone()
        for i in range(K):
            s = i % K
            g = gather.wait(s)
            torch.cuda.synchronize()
            for r in range(world):
                blk = g[r * Bl:(r + 1) * Bl]
                gather_ok = gather_ok and bool(torch.equal(blk, own[(i + r) % K]))
                gather_checked += 1
    kern = np.array([e[1].elapsed_time(e[2]) for e in ev])
    kern_ms = float(kern.mean())
    wait_stream_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    # iteration counts / statuses of the K draws (the last launch of each is still in its slot when steps >= K)
    live = [o for o in outs if o is not None]
    iters = float(np.mean([o["iters"].float().mean().item() for o in live]))
    n_opt = int(sum(int((o["status"] == 0).sum().item()) for o in live))
    n_tot = Bl * len(live)
    out = outs[(a.steps - 1) % K]
    per_draw = [None] * K
    for i in range(a.steps):
        j = (a.warmup + i + rank) % K
        per_draw[j] = (per_draw[j] or []) + [kern[i]]
    per_draw_ms = [float(np.mean(x)) if x else None for x in per_draw]
    # per-rank diagnostics for the N > 1 line: what each rank's kernel took, how long its solve stream stood waiting for a gather, its iterations
    mine = torch.tensor([kern_ms, float(kern.min()), float(kern.max()), wait_stream_ms, wait_host / a.steps * 1e3, iters], dtype=torch.float64,
                        device=dev if a.backend == "nccl" else "cpu")
    if world > 1:
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
    else:
        allr = mine.cpu().numpy()[None]

    res = None
    if rank == 0:
        value = B * a.steps / el
        peak = FP64_PEAK_TFLOPS if a.dtype == "f64" else FP32_PEAK_TFLOPS
        kname = kernel_name(N, a.dtype, Bl)
        flops_exec = executed_flops_per_iteration(N) * iters * Bl       # what one launch executes
        flops_model = algorithmic_flops_per_iteration(N) * iters * Bl   # SURVEY.md 8(d) formula (O(N^3) structured condensing)
        byts = algorithmic_bytes_per_solve(N, es) * Bl
        ach_tf = flops_exec / (kern_ms * 1e-3) / 1e12
        ach_gbs = byts / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic_bytes(kname) if (world == 1 and Bl == 4096 and N == 20 and a.dtype == "f64") else (None, None)
        cfg_label = ("BASELINE configs[3]: batch=%d sharded across %d GPUs (%d problems per GPU), N=%d, %s, all-gather of (accel, steer)" % (B, world, Bl, N, a.dtype)
                     if (world == 8 and Bl == 262144 and N == 20) else
                     "BASELINE configs[1]: batch=%d problems per GPU, N=%d, %s, one wavefront per problem" % (Bl, N, a.dtype))
        res = {
            "metric": "MPC solves/sec (batch, N=20 bicycle)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": cfg_label + "; every rank cycles through the same %d seeded draws (rank r solves draw (step + r) %% %d)" % (K, K),
                       "batch_per_gpu": Bl, "global_batch": B, "horizon": N, "parallelism": "shard%d" % world, "draws": K,
                       "draw_seeds": [20180620 + 2 + 7919 * j for j in range(K)], "kernel_ms_per_draw": per_draw_ms,
                       "mean_iterations": iters, "optimal_fraction": n_opt / n_tot, "clock_ramp_launches": a.clock_ramp,
                       "collective": ("none (one rank)" if (world == 1 and not forced) else
                                      "%s all_gather_into_tensor(async_op=True) of the [%d, 2] (accel, steer) block per step, %d slots%s"
                                      % ("RCCL" if a.backend == "nccl" else "gloo", Bl, K, ", forced in a 1-rank group" if forced else "")),
                       "gather_blocks_checked": gather_checked, "gather_blocks_equal_to_the_ranks_solutions": gather_ok},
            # the path is compute/latency-bound.  The SQ counters (profiles/r*_sq_counters.json) say the kernel is bound by
            # fp64 VALU ISSUE while the chip is full and by single-wave latency in the tail, not by the matrix cores; the denominator is the
            # fp64 vector = fp64 MFMA peak (78.6 TFLOP/s), the numerator the flops the kernel EXECUTES per launch (the SURVEY 8(d) model,
            # whose O(N^3) condensing term the kernels no longer run, is kept beside it as frac_survey_model so that rounds compare)
            "roofline": {"bound": "valu", "bound_note": "fp64 VALU issue while the chip is full, single-wave latency in the tail; MFMA pipes ~5 %% busy "
                                                        "(%s); priced against the fp64 vector = MFMA peak" % ((counter_profile(kname) or {}).get("file", "profiles/")),
                         "achieved": ach_tf, "peak": peak, "unit": "TFLOP/s", "frac": ach_tf / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "kernel_ms": kern_ms, "launches_timed": a.steps,
                         "flops_per_solve": executed_flops_per_iteration(N) * iters,
                         "achieved_survey_model": flops_model / (kern_ms * 1e-3) / 1e12, "frac_survey_model": flops_model / (kern_ms * 1e-3) / 1e12 / peak,
                         "survey_model_flops_per_solve": algorithmic_flops_per_iteration(N) * iters,
                         "flops_note": "achieved / frac price the arithmetic the kernel issues (condensing by the O(N^2) adjoint recursion: 52 N (N+1) "
                                       "flops per iteration); frac_survey_model prices SURVEY 8(d)'s formula (sum 16 k^2 + 32 N^2 for that stage)"},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": byts,
                             "bytes_per_solve": algorithmic_bytes_per_solve(N, es)},
            # one row per rank: with N > 1 a step runs at the pace of the ranks it is coupled to through the gathers; these say whether a
            # slow line is a slow kernel (draw), a rank waiting for others' gathers, or the collective itself
            "ranks": {"kernel_ms_mean": [float(x) for x in allr[:, 0]], "kernel_ms_min": [float(x) for x in allr[:, 1]],
                      "kernel_ms_max": [float(x) for x in allr[:, 2]], "gather_wait_stream_ms": [float(x) for x in allr[:, 3]],
                      "gather_wait_host_ms": [float(x) for x in allr[:, 4]], "mean_iterations": [float(x) for x in allr[:, 5]],
                      "summary": {"kernel_ms": {"min": float(allr[:, 0].min()), "mean": float(allr[:, 0].mean()), "max": float(allr[:, 0].max())},
                                  "gather_wait_stream_ms": {"min": float(allr[:, 3].min()), "mean": float(allr[:, 3].mean()), "max": float(allr[:, 3].max())}},
                      "note": "gather_wait_*: time per step the solve stream (device events) / the host stood in SolutionGather.wait before a launch"},
        }
        if world == 1:
            # p50 per-solve latency: B = 1 through the same entry point, host clock around launch + sync
            one = {k: v[:1].contiguous() for k, v in din.items()}
            o1 = None
            lat = []
            for i in range(60):
                torch.cuda.synchronize()
                t = time.perf_counter()
                o1 = solver.solve(one["z0"], one["ref"], one["v_target"], one["u_prev"], out=o1)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t)
            res["p50_latency_us_B1"] = float(np.percentile(lat[10:], 50) * 1e6)
            res["closed_loop_N8"] = closed_loop_latency(local)
            res["config1_gpu_B1_cold"] = gpu_config1_latency(local)
            res["dropin_host_path_N8"] = dropin_host_path_latency(local)
            if Bl == 4096 and N == 20 and a.dtype == "f64" and not a.quick:
                # the headline cycles through K draws; the same workload over 8 OTHER seeded batches (launch time = slowest problem of the draw)
                res["closed_loop_fleet_N8_B4096"] = closed_loop_fleet(local)
                res["multi_seed"] = multi_seed(solver, N, Bl, tdt, dev)
                res["two_batches_in_flight"] = two_in_flight(N, Bl, tdt, din, local)
                # the other single-GPU BASELINE configs, untimed-headline extra keys: configs[2] (fp32, B = 262144) and configs[4] (N = 50)
                res["config3_fp32_B262144"] = other_config(20, 262144, "f32", 3, dev, local, steps=5, warmup=2)
                # ... and through the packed-record entry point (ABI v8): same kernel, same results, whole-line fetches behind the start-order permutation
                res["config3_fp32_B262144_packed"] = other_config(20, 262144, "f32", 3, dev, local, steps=5, warmup=2, packed=True)
                res["config5_N50_B4096"] = other_config(50, 4096, "f64", 5, dev, local, steps=5, warmup=2)
                # the shard one GPU of configs[3] (2 097 152 problems over 8 GPUs) gets, fp64
                res["config4_shard_fp64_B262144"] = other_config(20, 262144, "f64", 4, dev, local, steps=3, warmup=1)
                # the reference's own horizon (N = 8, configs[0]'s model) at the large batch: four problems per wave (kmpc_quad.hip)
                res["N8_fp64_B262144"] = other_config(8, 262144, "f64", 3, dev, local, steps=3, warmup=1)
            # the reference's own Julia/Ipopt path is timed only if it is already installed (never fetched)
            import shutil
            res["reference_julia_ipopt_baseline"] = ("julia found at %s: not run (the reference's files do not travel)" % shutil.which("julia")) if shutil.which("julia") else "unavailable (no julia on this host)"
            if not a.no_cpu_baseline:
                cb, ro = cpu_baseline(N, d)
                res["cpu_baseline"] = cb
                res["cpu_baseline_config1"] = cpu_baseline_config1()   # BASELINE.md section 3, run C1 (beside closed_loop_N8 / config1_gpu_B1_cold)
                out = solver.solve(din["z0"], din["ref"], din["v_target"], din["u_prev"], want_U=True)  # untimed, with the input trajectories
                torch.cuda.synchronize()
                # parity spot check of the timed batch against the CPU port on the sampled problems
                res["parity_sample"] = parity_sample(N, d, out, ro)
        print(json.dumps(res), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
