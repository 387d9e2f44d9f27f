"""rocprofv3 --kernel-trace --output-format csv  ->  per (kernel, grid) calls / average / min / max duration (the tool's own --stats summary
averages a kernel over all its grids: the B = 4096 launches of the bench and its B = 1 latency launches share a kernel name).
usage: python tools/trace_digest.py <dir with *_kernel_trace.csv> <out.csv>"""
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
agg = {}
with open(f, newline="") as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        if "kmpc_" not in name:
            continue
        grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
        wg = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = (name, grid, wg)
        a = agg.setdefault(k, [0, 0, 1 << 62, 0, r.get("VGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size")])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
with open(dst, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "GridX", "WorkgroupX", "Workgroups", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "VGPRs", "ScratchBytesPerLane", "LDSBytes"])
    for (name, grid, wg), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, grid, wg, grid // max(wg, 1), a[0], a[1], a[1] / a[0], a[2], a[3], a[4], a[5], a[6]])
print(open(dst).read()[:3000])
