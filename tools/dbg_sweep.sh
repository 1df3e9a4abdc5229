#!/bin/bash
# experiment sweep: tools/dbg_sweep.sh ENVVAR v1 v2 ... runs the scene-6 FHD perf probe once per value
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env "$var=$v" timeout -k 10 120 python tools/gpu_perf.py 6,1920,200 2>&1 | tail -1 || exit 1
done
