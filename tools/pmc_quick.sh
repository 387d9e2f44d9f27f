#!/bin/bash
# quick check: bench line + WRITE_SIZE/FETCH_SIZE per B=4096 dispatch (two separate PMC passes)
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['p50_latency_us_B1'])"
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/w.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/f.err
python3 - <<'PY'
import sqlite3, glob
for sub, ctr in (("w","WRITE_SIZE"),("f","FETCH_SIZE")):
    c = sqlite3.connect(glob.glob("gpurun_out/pmcq/%s/*_results.db" % sub)[0])
    r = c.execute("select avg(value), count(*), max(scratch_size) from counters_collection where counter_name=? and kernel_name like '%kmpc_solve_fast_kernel<double, 20>%' and grid_size=262144", (ctr,)).fetchone()
    print(ctr, "KB/dispatch %.0f over %d dispatches, scratch %s B/lane" % r)
PY
