#!/bin/bash
# instruction-cache / LDS-conflict counter pass of the short bench (diagnostic)
set -eo pipefail
export TMPDIR=/tmp
OUT=gpurun_out/sqi; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/a -o a -- python3 bench.py --quick --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err
python3 - <<'PY'
import sqlite3, glob, json
c = sqlite3.connect(glob.glob("gpurun_out/sqi/a/*_results.db")[0])
out = {}
for name, val, n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%kmpc_solve_fast_kernel<double, 20>%' and grid_size=262144 group by counter_name"):
    out[name] = val
json.dump(out, open("gpurun_out/sqi/icache.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/a
