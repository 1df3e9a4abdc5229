#!/bin/bash
# HBM byte counters + SQ occupancy/utilisation only (three separate --pmc passes). Usage: tools/run_pmc_hbm.sh <outdir> <spp>
set -e
OUT=gpurun_out/$1; SPP=${2:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name done"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
