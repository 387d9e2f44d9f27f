"""bench.py as the driver calls it (GPU box): the one-GPU line carries every key the contract and VERDICT r1 ask for, and
`--gpus 2` without a launcher starts two ranks itself (gloo rehearsal on the one GPU of the box: two ranks share it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    d = _run(["--steps", "5", "--warmup", "2"])
    assert d["n_gpus"] == 1 and d["unit"] == "solves/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["batch_per_gpu"] == 4096 and d["config"]["horizon"] == 20 and d["config"]["optimal_fraction"] == 1.0
    rf = d["roofline"]
    assert rf["bound"] == "valu" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["kernel"] == "kmpc_solve_fast_kernel<double,20>"
    # `frac` prices executed flops; the SURVEY 8(d) model figure rides beside it and is the larger one
    assert rf["frac_survey_model"] > rf["frac"] > 0 and abs(rf["achieved"] * 1e12 * rf["kernel_ms"] * 1e-3 / 4096 - rf["flops_per_solve"]) < 1e-6 * rf["flops_per_solve"]
    assert d["config"]["draws"] == 4 and len(d["config"]["kernel_ms_per_draw"]) == 4
    rk = d["ranks"]
    assert len(rk["kernel_ms_mean"]) == 1 and abs(rk["kernel_ms_mean"][0] - rf["kernel_ms"]) < 1e-9 and rk["gather_wait_stream_ms"][0] < 0.05
    assert "not measured in this run" in rf["traffic_source"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and rf["kernel_ms"] <= d["ms_per_step"] * 1.02
    ms = d["multi_seed"]
    assert ms["seeds"] >= 8 and ms["min"] <= ms["mean"] <= ms["max"] and ms["all_optimal"] and ms["optimal_fraction"] == 1.0 and len(ms["per_seed"]) == ms["seeds"]
    tb = d["two_batches_in_flight"]
    assert tb["batches_in_flight"] == 2 and tb["all_optimal"] and tb["solves_per_s"] > d["value"]   # the next batch fills the tail
    for key, kern_peak in (("config3_fp32_B262144", 157.3), ("config5_N50_B4096", 78.6), ("config4_shard_fp64_B262144", 78.6), ("N8_fp64_B262144", 78.6)):
        c = d[key]
        assert c["optimal_fraction"] == 1.0 and c["peak_tflops"] == kern_peak and 0 < c["frac_of_peak"] < c["frac_survey_model"] < 1 and c["solves_per_s"] > 0
    pk = d["config3_fp32_B262144_packed"]   # ABI v8: same kernel on packed records -- same iterations, no slower (measured equal), less counter traffic where committed
    assert pk["optimal_fraction"] == 1.0 and pk["mean_iterations"] == d["config3_fp32_B262144"]["mean_iterations"] and pk["solves_per_s"] >= 0.93 * d["config3_fp32_B262144"]["solves_per_s"]
    assert d["config3_fp32_B262144"]["kernel"] == "kmpc_solve_fast_kernel<float,20>" and d["config5_N50_B4096"]["kernel"] == "kmpc_solve_wide_kernel<double,50>"
    assert d["config5_N50_B4096"]["solves_per_s"] >= 4e5
    assert d["N8_fp64_B262144"]["kernel"] == "kmpc_solve_quad_kernel<double>" and d["N8_fp64_B262144"]["solves_per_s"] >= 4e7
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    c1 = d["cpu_baseline_config1"]   # BASELINE.md section 3, run C1: CPU port, 1 thread, the module-load problem cold + its 10 Hz continuation
    assert c1["cores"] == 1 and c1["cold"]["solves"] >= 1000 and c1["warm_10hz"]["solves"] >= 1000 and c1["cold"]["status"] == 0 and c1["warm_10hz"]["worst_status"] == 0
    assert abs(c1["cold"]["cost"] - 15738.467) < 1e-2 and c1["cold"]["p50_us"] <= c1["cold"]["p99_us"] < c1["reference_cap_us"]
    g1 = d["config1_gpu_B1_cold"]
    assert g1["status"] == 0 and abs(g1["cost"] - 15738.467) < 1e-2 and g1["iterations"] == c1["cold"]["iterations"]
    hp = d["dropin_host_path_N8"]   # the module-API mirror on host arrays: no copy launches since round 4 (174 us before)
    assert hp["worst_status"] == 0 and hp["p50_step_us"] < 120 and hp["p50_step_us"] <= hp["p99_step_us"]
    assert d["parity_sample"]["unexplained"] == 0 and d["parity_sample"]["within_tol"] >= 4096 - 8
    assert "unavailable" in d["reference_julia_ipopt_baseline"] or "julia found" in d["reference_julia_ipopt_baseline"]


def test_bench_launches_its_own_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--quick", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "shard2"
    assert abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    # the N > 1 line says what every rank's kernel took, how long it waited for gathers, and its iteration counts
    rk = d["ranks"]
    for k in ("kernel_ms_mean", "kernel_ms_min", "kernel_ms_max", "gather_wait_stream_ms", "gather_wait_host_ms", "mean_iterations"):
        assert len(rk[k]) == 2 and all(v >= 0 for v in rk[k]), k
    assert rk["summary"]["kernel_ms"]["min"] <= rk["summary"]["kernel_ms"]["mean"] <= rk["summary"]["kernel_ms"]["max"]
    assert d["config"]["draws"] == 4 and "rank r solves draw" in d["config"]["workload"]
    assert d["config"]["gather_blocks_checked"] == 8 and d["config"]["gather_blocks_equal_to_the_ranks_solutions"] and d["config"]["collective"].startswith("gloo")


def test_bench_forced_collective_runs_rccl_in_a_one_rank_group():
    """the RCCL path of the N > 1 loop under test before multi-GPU hardware shows up.  A fresh child process (no GPU call before the
    process group exists) runs `bench.py --gpus 1 --force-collective`: init_process_group("nccl", world_size=1, device_id=...), the K-slot asynchronous
    all_gather_into_tensor of every step's (accel, steer) block on the collective's stream, every gathered block compared with the step's own solution,
    and the per-rank wait times reported.  Its JSON line is kept under profiles/ (copied there from gpurun_out/ by hand: the test writes gpurun_out/)."""
    d = _run(["--gpus", "1", "--force-collective", "--steps", "8", "--warmup", "2", "--quick", "--no-cpu-baseline"])
    c = d["config"]
    assert d["n_gpus"] == 1 and c["collective"].startswith("RCCL all_gather_into_tensor(async_op=True)") and "forced in a 1-rank group" in c["collective"]
    assert c["gather_blocks_checked"] == 4 and c["gather_blocks_equal_to_the_ranks_solutions"] is True and c["optimal_fraction"] == 1.0
    rk = d["ranks"]
    assert len(rk["gather_wait_stream_ms"]) == 1 and rk["gather_wait_stream_ms"][0] >= 0 and rk["gather_wait_host_ms"][0] >= 0
    # the collective is overlapped: a step costs at most a few per cent more than its kernel (measured: 0.47 vs 0.46 ms)
    assert d["ms_per_step"] <= 1.25 * rk["kernel_ms_mean"][0] + 0.05
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl_one_rank_bench.json"), "w") as f:
        json.dump(d, f)
