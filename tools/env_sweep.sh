#!/bin/bash
# experiment sweep over full env assignments: SPEC=6,1920,200 tools/env_sweep.sh "A=1 B=2" "A=2" ... (perf probe per set)
SPEC=${SPEC:-6,1920,200}
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 160 python tools/gpu_perf.py $SPEC 2>&1 | tail -1 || exit 1
done
