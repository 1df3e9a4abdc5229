cd $GRAFT_REPO_ROOT
bash tools/run_profiles.sh r03 > gpurun_out/prof_r03.log 2>&1
tail -30 gpurun_out/prof_r03.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace1000 -- python3 bench.py --spp 1000 --steps 1 --warmup 0 --no-cpu-baseline > $O/trace1000.log 2>&1
python3 tools/trace_tail.py $O/trace1000 $O/tail_1000spp.json
find $O/trace1000 -name "*.csv" -size +1M -delete
