cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03d; OUT=gpurun_out/r03d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -i -E "icache|ifetch|SQC_|INST_CACHE|SQ_INSTS_|SQ_WAIT" $OUT/avail.txt | head -80 > $OUT/avail_sel.txt
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --spp 1000 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run ic3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY SQ_WAIT_ANY
python3 - <<'PY'
import csv,glob,collections
for name in ("ic1","ic2","ic3"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(f"gpurun_out/r03d/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k="k_shade" if "k_shade" in r["Kernel_Name"] else "k_extend" if "k_extend" in r["Kernel_Name"] else None
            if not k: continue
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            n[(k,r["Counter_Name"])]+=1
    for k in agg:
        print(name,k,{c: round(v/n[(k,c)],1) for c,v in agg[k].items()})
PY
find $OUT -name "*.csv" -size +200k -delete
