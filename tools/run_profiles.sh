#!/bin/bash
# Round profile set: kernel-trace stats + PMC byte counters (separate passes, each under its own
# timeout; --pmc only ever combined with --kernel-trace). Usage: tools/run_profiles.sh <tag>
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p "$OUT"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --spp 400 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 bench.py --spp 100 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$c.log" 2>&1; echo "$c rc=$?"
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py --spp 100 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/sq.log" 2>&1; echo "sq rc=$?"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/tcc" -- python3 bench.py --spp 100 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/tcc.log" 2>&1; echo "tcc rc=$?"
