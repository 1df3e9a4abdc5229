#!/bin/bash
# Interleaved A/B of library builds on one GPU: SPEC=6,1920,1000 ROUNDS=2 tools/ab_perf.sh r01 cur ...
# (variants/libpt_amd_<name>.so, built by tools/build_variant.sh; "cur" = the in-tree libpt_amd.so)
SPEC=${SPEC:-6,1920,1000}; ROUNDS=${ROUNDS:-2}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    LIB="$ROOT/thu-acg-f2024-path-tracer_amd/variants/libpt_amd_$v.so"
    [ "$v" = "cur" ] && LIB="$ROOT/thu-acg-f2024-path-tracer_amd/libpt_amd.so"
    echo -n "[$v r$r] "
    PT_AMD_LIB="$LIB" timeout -k 10 200 python tools/gpu_perf.py $SPEC 2>&1 | tail -1 || exit 1
  done
done
