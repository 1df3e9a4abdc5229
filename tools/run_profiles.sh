#!/bin/bash
# Round profile set for the headline workload (scene 6, 1920x1080 @ 4000 spp):
#   stats : rocprofv3 --kernel-trace --stats of the default bench command (1 warm-up + 2 timed frames)
#   pmc   : FETCH_SIZE / WRITE_SIZE / SQ counters, one --pmc pass each (never combined with other trace domains),
#           on one frame of the same workload
# Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p "$OUT"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$c.log" 2>&1; echo "$c rc=$?"
done
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/sq.log" 2>&1; echo "sq rc=$?"
# the raw per-dispatch CSVs are large: keep the kernel stats and per-kernel means only
python3 tools/pmc_summary.py "$OUT" "$TAG" 33554432 > "$OUT/pmc_summary.txt" 2>&1
cp profiles/${TAG}_pmc_summary.json "$OUT/" 2>/dev/null
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
ls -R "$OUT" | head -40
