#!/bin/bash
# rocprofv3 kernel-trace stats of the bench workload. Usage: tools/run_stats.sh <tag> [spp]   (env passes through)
TAG=$1; SPP=${2:-200}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/stats_$TAG; mkdir -p "$OUT"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 bench.py --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/bench.log" 2>&1; echo "stats rc=$?"
cat "$OUT"/*/*kernel_stats.csv
