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
    assert "not measured in this run" in rf["traffic_source"]
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"] and rf["kernel_ms"] <= d["ms_per_step"] * 1.02
    ms = d["multi_seed"]
    assert ms["seeds"] >= 8 and ms["min"] <= ms["mean"] <= ms["max"] and ms["all_optimal"]
    tb = d["two_batches_in_flight"]
    assert tb["batches_in_flight"] == 2 and tb["all_optimal"] and tb["solves_per_s"] > d["value"]   # the next batch fills the tail
    for key, kern_peak in (("config3_fp32_B262144", 157.3), ("config5_N50_B4096", 78.6), ("config4_shard_fp64_B262144", 78.6)):
        c = d[key]
        assert c["optimal_fraction"] == 1.0 and c["peak_tflops"] == kern_peak and 0 < c["frac_of_peak"] < 1 and c["solves_per_s"] > 0
    assert d["config5_N50_B4096"]["solves_per_s"] >= 4e5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert d["parity_sample"]["unexplained"] == 0 and d["parity_sample"]["within_tol"] >= 4096 - 8
    assert "unavailable" in d["reference_julia_ipopt_baseline"] or "julia found" in d["reference_julia_ipopt_baseline"]


def test_bench_launches_its_own_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--quick", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["config"]["parallelism"] == "shard2"
    assert abs(d["value"] - 8192 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
