#!/bin/bash
# gpurun -- bash tools/calib/run.sh   -> prints counter bytes / known bytes per kernel
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/traffic_calib tools/calib/traffic_calib.hip
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f -- $OUT/traffic_calib > $OUT/f.log 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w -- $OUT/traffic_calib > $OUT/w.log 2> $OUT/w.err
python3 - <<'PY'
import sqlite3, glob
known = {"write8": 2**31, "read8": 2**31, "read24": (2**28 // 3) * 24}
for sub, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    c = sqlite3.connect(glob.glob("gpurun_out/calib/%s/*_results.db" % sub)[0])
    for name, val in c.execute("select kernel_name, value from counters_collection where counter_name=?", (ctr,)):
        k = [x for x in known if x in name]
        if k: print("%-10s %-8s counter %.1f MB / known %.1f MB = %.3f" % (ctr, k[0], val * 1024 / 1e6, known[k[0]] / 1e6, val * 1024 / known[k[0]]))
PY
rm -f $OUT/traffic_calib
