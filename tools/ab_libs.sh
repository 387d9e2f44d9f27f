#!/bin/bash
# A/B/C... of library builds with tools/bench_quick.py: usage  bash tools/ab_libs.sh "cfg;cfg;..." lib1.so lib2.so ...   (cfg = env assignments, e.g. "QN=20 QB=4096")
IFS=';' read -ra CFGS <<< "$1"; shift
for cfg in "${CFGS[@]}"; do
  echo "== $cfg"
  for lib in "$@"; do
    env $cfg KMPC_LIB=$lib python tools/bench_quick.py 2>/dev/null | sed "s|^|$(basename $lib .so | sed 's/libkmpc_hip//') |"
  done
done
