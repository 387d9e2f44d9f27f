"""Per-kernel register / scratch / LDS / occupancy table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py mkz_mpc_path_follower_amd/csrc/kmpc_fast.hip [substring ...]"""
import re, subprocess, sys
src = sys.argv[1]
filt = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on",
       "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null", src]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for l in err.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass-analysis", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-64s %5s %5s %9s %9s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPRspill", "VGPRspill", "scratch", "LDS", "occ"))
for r, d in zip(rows, names):
    d = re.sub(r"\(KP.*", "", d).replace("void ", "")
    if filt and not any(f in d for f in filt):
        continue
    print("%-64s %5s %5s %9s %9s %8s %7s %4s" % (d[:64], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                   r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
