#!/bin/bash
# A/B of two builds of the library with tools/bench_quick.py: usage  bash tools/ab.sh base.so [QN=.. QB=..]   (new = the shipped libkmpc_hip.so)
BASE=$1; shift
for cfg in "QN=20 QB=4096" "QN=20 QB=4096 QSEED=20188541" "QN=20 QB=4096 QSEED=20204379" "QN=20 QB=262144" "QN=16 QB=4096" "QN=12 QB=262144" "QN=8 QB=4096"; do
  echo "== $cfg"
  env $cfg python tools/bench_quick.py 2>/dev/null | sed 's/^/new  /'
  env $cfg KMPC_LIB=$BASE python tools/bench_quick.py 2>/dev/null | sed 's/^/base /'
done
