#!/bin/bash
# gpurun -- bash tools/calib/issue_run.sh   -> per-instruction cycle costs of one fp64 wave / two waves per SIMD (tools/calib/issue_probe.hip)
set -eo pipefail
OUT=gpurun_out/calib; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/issue_probe tools/calib/issue_probe.hip
$OUT/issue_probe | tee $OUT/issue_probe.txt
rm -f $OUT/issue_probe
