"""Turn the rocprofv3 (rocpd sqlite) output of tools/profile_round.sh into the committed summaries under profiles/:
   <tag>_kernel_stats.csv  per (kernel, grid) calls / average / min / max duration from --kernel-trace
   <tag>_bench.json        the bench line of the same command without the profiler
   <round>_pmc_traffic.json FETCH_SIZE / WRITE_SIZE per dispatch of the B=4096 solve kernel (separate --pmc passes)
Usage: python tools/profile_digest.py r1_v5 "note about the build" """
import csv, glob, json, os, shutil, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def db(sub):
    return sqlite3.connect(glob.glob(os.path.join(src, sub, "*_results.db"))[0])


rows = db("trace").execute(
    "select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(scratch_size), max(lds_size) "
    "from kernels group by name, grid_x order by sum(duration) desc").fetchall()
with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "GridX", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "VGPRs", "ScratchBytesPerLane", "LDSBytes"])
    for r in rows:
        w.writerow(r)
shutil.copy(os.path.join(src, "bench_plain.json"), os.path.join(ROOT, "profiles", tag + "_bench.json"))
if glob.glob(os.path.join(src, "trace_full", "*_results.db")):   # the full default bench: multi-seed batches and the other BASELINE configs
    rows_all = db("trace_full").execute(
        "select name, grid_x, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(scratch_size), max(lds_size) "
        "from kernels where name like '%kmpc_%' group by name, grid_x order by sum(duration) desc").fetchall()
    with open(os.path.join(ROOT, "profiles", tag + "_kernel_stats_all_configs.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "GridX", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "VGPRs", "ScratchBytesPerLane", "LDSBytes"])
        for r in rows_all:
            w.writerow(r)

out = {}
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    r = db(sub).execute(
        "select avg(value), count(*), max(scratch_size), max(lds_block_size) from counters_collection "
        "where counter_name = ? and kernel_name like '%kmpc_solve_fast_kernel<double, 20>%' and grid_size = ?", (ctr, 4096 * 64)).fetchone()
    out[ctr + "_KB_per_dispatch"] = r[0]
    out["dispatches_averaged"] = r[1]
    out["scratch_bytes_per_lane"] = r[2]
    out["lds_bytes_per_wave"] = r[3]
# FETCH_SIZE reports half of the bytes read on gfx950 (calibrated on 8 B and 24 B per lane reads: tools/calib/run.sh); WRITE_SIZE is exact
out["hbm_bytes_per_dispatch_uncorrected"] = 1024.0 * (out["FETCH_SIZE_KB_per_dispatch"] + out["WRITE_SIZE_KB_per_dispatch"])
out["hbm_bytes_per_dispatch"] = 1024.0 * (2 * out["FETCH_SIZE_KB_per_dispatch"] + out["WRITE_SIZE_KB_per_dispatch"])
if note:
    out["build"] = note
rnd = tag.split("_")[0]   # r1 / r2 ...: one traffic file per round
pj = os.path.join(ROOT, "profiles", rnd + "_pmc_traffic.json")
if os.path.exists(pj):
    j = json.load(open(pj))
else:  # a new round starts from the calibration note of the previous one
    prev = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
    j = {k: v for k, v in prev.items() if k not in ("runs", "current")}
    j["runs"] = {}
j["runs"][tag] = out
j["current"] = tag
json.dump(j, open(pj, "w"), indent=1)
st = glob.glob(os.path.join(src, "trace_csv", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(ROOT, "profiles", tag + "_rocprofv3_kernel_stats.csv"))
for r in rows[:6]:
    print(r)
print(json.dumps(out, indent=1))
