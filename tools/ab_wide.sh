#!/bin/bash
BASE=$1; shift
for cfg in "QN=50 QB=4096" "QN=50 QB=16384" "QN=40 QB=4096" "QN=32 QB=16384" "QN=24 QB=4096" "QN=28 QB=4096"; do
  echo "== $cfg"
  env $cfg python tools/bench_quick.py 2>/dev/null | sed 's/^/new  /'
  env $cfg KMPC_LIB=$BASE python tools/bench_quick.py 2>/dev/null | sed 's/^/base /'
done
