#!/bin/bash
# Counter evidence for the three kernels the bench line prices, run on the GPU box from the repo root:
#   headline  kmpc_solve_fast_kernel<double,20>  B = 4096     (bench.py --quick)
#   config 3  kmpc_solve_fast_kernel<float,20>   B = 262144   (tools/launch_config.py)
#   config 5  kmpc_solve_wide_kernel<double,50>  B = 4096
# per kernel: --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, no tracing: MI355X_MICROARCH.md HBM recipe) and two SQ passes.
# The program sits directly after `rocprofv3 ... --`.  Digest: gpurun_out/pmc_<tag>/counters.json  ->  profiles/<tag>_pmc_counters.json
set -eo pipefail
export TMPDIR=/tmp
TAG=${1:-r4_v3}
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
SQB="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run() {  # name, command...
  local name=$1; shift
  rocprofv3 --pmc FETCH_SIZE -d $OUT/$name/fetch -o c -- "$@" > $OUT/$name.fetch.out 2> $OUT/$name.fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/$name/write -o c -- "$@" > $OUT/$name.write.out 2> $OUT/$name.write.err
  rocprofv3 --pmc $SQA -d $OUT/$name/sqa -o c -- "$@" > $OUT/$name.sqa.out 2> $OUT/$name.sqa.err
  rocprofv3 --pmc $SQB -d $OUT/$name/sqb -o c -- "$@" > $OUT/$name.sqb.out 2> $OUT/$name.sqb.err
  echo "$name done"; tail -1 $OUT/$name.fetch.out || true
}
run headline python3 bench.py --quick --steps 8 --warmup 4 --no-cpu-baseline
run config3 python3 tools/launch_config.py --horizon 20 --dtype f32 --batch 262144 --cfg 3 --steps 3
run config5 python3 tools/launch_config.py --horizon 50 --dtype f64 --batch 4096 --cfg 5 --steps 3
run config3_packed python3 tools/launch_config.py --horizon 20 --dtype f32 --batch 262144 --cfg 3 --steps 3 --packed
run headline_packed python3 tools/launch_config.py --horizon 20 --dtype f64 --batch 4096 --cfg 2 --steps 8 --packed
python3 tools/pmc_digest.py $TAG > $OUT/digest.log
# the raw rocprofv3 databases are large (gpurun copies back at most 64 MiB): keep the digest and the logs only
for d in headline config3 config5 config3_packed headline_packed; do rm -rf $OUT/$d; done
tail -5 $OUT/digest.log
