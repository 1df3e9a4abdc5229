#!/bin/bash
# SQ occupancy / instruction-mix counters (two --pmc passes). Usage: tools/run_pmc_sq.sh <outdir> <spp>
set -e
OUT=gpurun_out/$1; SPP=${2:-100}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name done"; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVES
