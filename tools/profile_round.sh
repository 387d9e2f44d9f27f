#!/bin/bash
# Collect the judged profiles on the GPU box (run via gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) of a short bench, no tracing (MI355X_MICROARCH.md HBM recipe)
# Raw output lands in gpurun_out/prof_<tag>/ ; tools/profile_digest.py turns it into profiles/<tag>_*.
set -eo pipefail
TAG=${1:-r2_v1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
# headline workload only (--quick: no multi-seed / other-config launches, which share the headline kernel's name and grid)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --quick --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
# the tool's own per-kernel summary of the same command, as CSV
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_csv -o trace -- python3 bench.py --quick --no-cpu-baseline > $OUT/bench_traced_csv.json 2> $OUT/trace_csv.err
# the full default bench (multi-seed batches, configs 3 / 4 / 5): the other kernels
rocprofv3 --kernel-trace --stats -d $OUT/trace_full -o trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_traced_full.json 2> $OUT/trace_full.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 bench.py --quick --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 bench.py --quick --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
