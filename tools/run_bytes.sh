#!/bin/bash
# HBM bytes per launch only (FETCH_SIZE and WRITE_SIZE, one --pmc pass each on one frame of the headline workload) — the quick
# check after a change to the path records. Usage: tools/run_bytes.sh <tag>  -> gpurun_out/bytes_<tag>/pmc_summary.txt
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/bytes_$TAG; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$c.log" 2>&1; echo "$c rc=$?"
done
python3 tools/pmc_summary.py "$OUT" "bytes_$TAG" 33554432 > "$OUT/pmc_summary.txt" 2>&1
mv profiles/bytes_${TAG}_pmc_summary.json "$OUT/" 2>/dev/null
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
grep -A3 "hbm_\|^k_" "$OUT/pmc_summary.txt" | head -40
python3 - "$OUT/bytes_${TAG}_pmc_summary.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for k in ("k_extend","k_shade"):
    print(k, {x: round(d[k][x]/1e9,3) for x in d[k] if x.startswith("hbm_")})
PY
