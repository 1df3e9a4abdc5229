cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03n; O=gpurun_out/r03n
export PT_EXPERIMENT=1
for spp in 500 1000; do for slots in 4194304 8388608 16777216 33554432; do echo -n "spp $spp slots $slots: "; PT_POOL_SLOTS=$slots timeout -k 10 200 python tools/gpu_perf.py 6,1920,$spp 2>&1 | tail -1; done; done | tee $O/pool_sweep.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
unset PT_EXPERIMENT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace500 -- python3 bench.py --spp 500 --steps 1 --warmup 0 --no-cpu-baseline > $O/trace500.log 2>&1
python3 tools/trace_tail.py $O/trace500 $O/tail_500spp.json
python3 - <<'PY'
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r03n/trace500/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
t0=rows[0][0]
print("first kernels:", [(n, round((s-t0)/1e6,3), round((e-s)/1e6,3)) for s,e,n in rows[:5]])
print("last kernels:", [(n, round((s-t0)/1e6,3), round((e-s)/1e6,3)) for s,e,n in rows[-4:]])
gaps=sum(max(0,rows[i+1][0]-rows[i][1]) for i in range(len(rows)-1))/1e6
print("sum of gaps between kernels ms", round(gaps,2), "n kernels", len(rows), "span ms", round((rows[-1][1]-t0)/1e6,2))
PY
find $O/trace500 -name "*.csv" -size +1M -delete
