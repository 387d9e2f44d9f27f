#!/bin/bash
# GPU test suite, one pytest process per file in sequence (never in parallel), progress and results appended to gpurun_out/gpu_tests.log
# (nothing is hidden behind a pipe: a silent run is taken to be hung).  usage: bash tools/gpu_tests.sh [file ...]
mkdir -p gpurun_out
LOG=gpurun_out/gpu_tests.log; : > $LOG
FILES=${@:-$(ls tests/test_*.py)}
rc=0
for f in $FILES; do
  echo "=== $f $(date +%T)" | tee -a $LOG
  timeout -k 10 600 python -m pytest $f -m gpu -q -x --durations=5 -p no:cacheprovider >> $LOG 2>&1
  r=$?; echo "--- rc $r $(date +%T)" | tee -a $LOG
  if [ $r -ne 0 ] && [ $r -ne 5 ]; then rc=$r; echo "FAILED: $f" | tee -a $LOG; break; fi
done
grep -E "passed|failed|error" $LOG | tail -20
exit $rc
