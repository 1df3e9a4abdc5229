#!/bin/bash
# Attribution counters for K2/K3: one --pmc pass per counter group (never combined with trace domains
# other than --kernel-trace), program directly after `--`. Usage: tools/run_pmc_attr.sh <tag> [spp] [extra bench args]
TAG=$1; SPP=${2:-1000}; shift; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --spp "$SPP" --steps 1 --warmup 0 --no-cpu-baseline $EXTRA > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run insts SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU
run level SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
run ldsc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum
run mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_WAVES
run mix2 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_WAVES
SLOTS=$(python3 - <<PY
import json,glob
for f in glob.glob("$OUT/insts.log"):
    for l in open(f):
        if l.startswith("{"):
            print(json.loads(l)["config"]["resident_paths"]); break
PY
)
python3 tools/pmc_summary.py "$OUT" "${TAG}_attr" ${SLOTS:-16777216} > "$OUT/summary.txt" 2>&1
cp profiles/${TAG}_attr_pmc_summary.json "$OUT/" 2>/dev/null
grep -h '^{' "$OUT"/insts.log > "$OUT/bench_line.json"
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
cat "$OUT/summary.txt"
