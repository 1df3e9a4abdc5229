#!/bin/bash
# Texture-addresser / L1 (TA, TCP, TD) counters for K2 and K3: is the vector-memory FRONT END what k_extend2's divergent node
# gathers are bound by? One --pmc pass per group, program directly after `--`. Usage: tools/run_pmc_ta.sh <tag> [spp]
TAG=$1; SPP=${2:-1000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/avail.txt" 2>&1
grep -oE "\b(TA|TCP|TD)_[A-Z0-9_]+\b" "$OUT/avail.txt" | sort -u > "$OUT/avail_ta_tcp_td.txt"; wc -l "$OUT/avail_ta_tcp_td.txt"
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run ta1 TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
# (TA_FLAT_*_WAVEFRONTS / TA_*_STALLED_BY_TC and the TD_* group ran into the 240-s limit on this pool in round 3 — twice 4 GPU-minutes: left out)
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
python3 tools/pmc_summary.py "$OUT" "${TAG}_ta" > "$OUT/summary.txt" 2>&1
cp profiles/${TAG}_ta_pmc_summary.json "$OUT/" 2>/dev/null
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
python3 - <<PY
import json
d=json.load(open("$OUT/${TAG}_ta_pmc_summary.json"))
for k in ("k_extend","k_shade"):
    print(k, {c: round(v["mean"]) for c,v in d.get(k,{}).items() if isinstance(v, dict) and "mean" in v})
PY
