#!/bin/bash
# Round profile set for the headline workload (scene 6, 1920x1080 @ 4000 spp):
#   stats : rocprofv3 --kernel-trace --stats of the default bench command (1 warm-up + 2 timed frames)
#   pmc   : FETCH_SIZE / WRITE_SIZE / SQ counters, one --pmc pass each (never combined with other trace domains),
#           on one frame of the same workload; the program stands directly after `--`
# Usage: tools/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/  (+ profiles/<tag>_pmc_summary.json, pmc_latest.json on the box)
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p "$OUT"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$c.log" 2>&1; echo "$c rc=$?"
done
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/sq.log" 2>&1; echo "sq rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/tcc.log" 2>&1; echo "tcc rc=$?"
# the raw per-dispatch CSVs are large: keep the kernel stats and per-kernel means only
SLOTS=$(python3 -c "
import json
for l in open('$OUT/stats.log'):
    if l.startswith('{'): print(json.loads(l)['config']['resident_paths']); break")
python3 tools/pmc_summary.py "$OUT" "$TAG" ${SLOTS:-33554432} > "$OUT/pmc_summary.txt" 2>&1
cp profiles/${TAG}_pmc_summary.json "$OUT/" 2>/dev/null
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/${TAG}_bench_4000spp_kernel_stats.csv" 2>/dev/null
grep -h '^{' "$OUT/stats.log" > "$OUT/${TAG}_bench_4000spp_under_rocprof.json"
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/pmc_summary.txt"; cat "$OUT/${TAG}_bench_4000spp_kernel_stats.csv"
