#!/bin/bash
# Round-4 profile collection on the GPU box (run via gpurun from the repo root): everything lands under gpurun_out/prof_<tag>/ as small
# text / json files (the raw rocprofv3 databases are deleted: gpurun copies back at most 64 MiB).
#   1. rocprofv3 --kernel-trace --stats of the headline command (bench.py --quick) and of the full default bench (all kernels)
#   2. counter passes of the three priced kernels (tools/pmc_configs.sh): FETCH_SIZE, WRITE_SIZE, two SQ groups
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-r4_v3}
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
echo "bench done"; tail -c 300 $OUT/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --quick --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_headline.csv \;
python3 tools/trace_digest.py $OUT/trace $OUT/kernel_stats_headline.csv > /dev/null
rm -rf $OUT/trace
echo "trace headline done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -o trace -- python3 bench.py --no-cpu-baseline > $OUT/bench_traced_full.json 2> $OUT/trace_full.err
find $OUT/trace_full -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_all_configs.csv \;
python3 tools/trace_digest.py $OUT/trace_full $OUT/kernel_stats_all_configs.csv > /dev/null
rm -rf $OUT/trace_full
echo "trace full done"
bash tools/pmc_configs.sh $TAG > $OUT/pmc.log 2>&1
cp gpurun_out/pmc_$TAG/counters.json $OUT/pmc_counters.json
echo "pmc done"
ls -la $OUT
