#!/bin/bash
# A/B/C of library builds with tools/bench_quick.py and the fleet loop: usage  bash tools/ab3.sh base.so prev.so   (new = the shipped libkmpc_hip.so)
for cfg in "QN=20 QB=4096" "QN=20 QB=262144" "QN=8 QB=262144" "QN=8 QB=4096" "QN=12 QB=262144"; do
  echo "== $cfg"
  env $cfg python tools/bench_quick.py 2>/dev/null | sed 's/^/new  /'
  for L in "$@"; do env $cfg KMPC_LIB=$L python tools/bench_quick.py 2>/dev/null | sed "s|^|$L |"; done
done
for cfg in "QN=8 QB=4096 QWARM=1" "QN=20 QB=262144 QDT=f32"; do
  echo "== $cfg"
  env $cfg python tools/bench_quick.py 2>/dev/null | sed 's/^/new  /'
  for L in "$@"; do env $cfg KMPC_LIB=$L python tools/bench_quick.py 2>/dev/null | sed "s|^|$L |"; done
done
