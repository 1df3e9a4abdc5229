cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03v; O=gpurun_out/r03v
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 bench.py --scene 3 --width 1920 --steps 1 --warmup 0 --no-cpu-baseline > $O/c2.log 2>&1; echo "c2 rc=$?"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python3 bench.py --scene 5 --width 3840 --spp 1000 --steps 1 --warmup 0 --no-cpu-baseline > $O/c5.log 2>&1; echo "c5 rc=$?"
cp $O/c2/*/*kernel_stats.csv $O/r03_config2_kernel_stats.csv; cp $O/c5/*/*kernel_stats.csv $O/r03_config5_shard_kernel_stats.csv
grep -h '^{' $O/c2.log > $O/r03_config2_under_rocprof.json; grep -h '^{' $O/c5.log > $O/r03_config5_shard_under_rocprof.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -4 $O/r03_config2_kernel_stats.csv; head -4 $O/r03_config5_shard_kernel_stats.csv
