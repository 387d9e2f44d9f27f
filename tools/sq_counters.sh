#!/bin/bash
# SQ counter passes (<= 8 SQ slots each) of the short bench; digests into gpurun_out/sq_<tag>/sq_counters.json
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-r2_v1}
OUT=gpurun_out/sq_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/a -o a -- python3 bench.py --quick --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 -d $OUT/b -o b -- python3 bench.py --quick --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/b.err
python3 - "$TAG" <<'PY'
import sqlite3, glob, json, sys
tag = sys.argv[1]
out = {}
for sub in ("a", "b"):
    c = sqlite3.connect(glob.glob("gpurun_out/sq_%s/%s/*_results.db" % (tag, sub))[0])
    for name, val, n in c.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%kmpc_solve_fast_kernel<double, 20>%' and grid_size=262144 group by counter_name"):
        out[name] = val
out["note"] = "per B=4096 dispatch of kmpc_solve_fast_kernel<double,20>, summed over the chip; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)"
json.dump(out, open("gpurun_out/sq_%s/sq_counters.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
