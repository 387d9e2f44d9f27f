"""gpurun_out/pmc_<tag>/ (tools/pmc_configs.sh) -> gpurun_out/pmc_<tag>/counters.json, the file bench.py's counter_profile() reads once it
is copied to profiles/<tag>_pmc_counters.json.  HBM bytes per dispatch = 1024 * (2 * FETCH_SIZE_KB + WRITE_SIZE_KB): FETCH_SIZE tallies
128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md; calibrated in round 1/2 with tools/calib/, profiles/r2_pmc_traffic.json)."""
import glob, json, sqlite3, sys
tag = sys.argv[1]
root = "gpurun_out/pmc_%s" % tag
spec = {"headline": ("kmpc_solve_fast_kernel<double,20>", "%kmpc_solve_fast_kernel<double, 20>%", 4096, 64, "python3 bench.py --quick --steps 8 --warmup 4 --no-cpu-baseline"),
        "config3": ("kmpc_solve_fast_kernel<float,20>", "%kmpc_solve_fast_kernel<float, 20>%", 262144, 64, "python3 tools/launch_config.py --horizon 20 --dtype f32 --batch 262144 --cfg 3 --steps 3"),
        "config3_packed": ("kmpc_solve_fast_kernel<float,20> packed records", "%kmpc_solve_fast_kernel<float, 20>%", 262144, 64, "python3 tools/launch_config.py --horizon 20 --dtype f32 --batch 262144 --cfg 3 --steps 3 --packed"),
        "headline_packed": ("kmpc_solve_fast_kernel<double,20> packed records", "%kmpc_solve_fast_kernel<double, 20>%", 4096, 64, "python3 tools/launch_config.py --horizon 20 --dtype f64 --batch 4096 --cfg 2 --steps 8 --packed"),
        "config5": ("kmpc_solve_wide_kernel<double,50>", "%kmpc_solve_wide_kernel<double, 50>%", 4096, 256, "python3 tools/launch_config.py --horizon 50 --dtype f64 --batch 4096 --cfg 5 --steps 3")}
out = {"note": "rocprofv3 --pmc passes (separate runs per counter group, no tracing), averaged over the dispatches of the named kernel at the named grid; "
               "SQ_* summed over the chip per dispatch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)",
       "correction": "hbm_bytes_per_dispatch = 1024 * (2 * FETCH_SIZE_KB + WRITE_SIZE_KB)", "tag": tag, "kernels": {}}
for name, (kern, like, B, wg, cmd) in spec.items():
    k = {"batch": B, "command": cmd}
    for sub in ("fetch", "write", "sqa", "sqb"):
        dbs = glob.glob("%s/%s/%s/**/*_results.db" % (root, name, sub), recursive=True)
        if not dbs:
            continue
        c = sqlite3.connect(dbs[0])
        for cname, val, n, scr, lds in c.execute("select counter_name, avg(value), count(*), max(scratch_size), max(lds_block_size) from counters_collection "
                                                 "where kernel_name like ? and grid_size=? group by counter_name", (like, B * wg)):
            k[cname] = val; k["dispatches_averaged"] = n; k["scratch_bytes_per_lane"] = scr; k["lds_bytes_per_workgroup"] = lds
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["hbm_bytes_per_dispatch"] = 1024.0 * (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"])
    if "SQ_INSTS_VALU" in k:
        k["per_solve"] = {x: k[x] / B for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES") if x in k}
    out["kernels"][kern] = k
json.dump(out, open("%s/counters.json" % root, "w"), indent=1)
print(json.dumps(out, indent=1))
