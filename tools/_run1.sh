set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r03a/pytest.log
export PT_EXPERIMENT=1
for v in "PT_ACCUM_LINEAR=1" "PT_X=0" "PT_ACCUM_LINEAR=1" "PT_X=0"; do echo "== $v"; env $v timeout -k 10 200 python tools/gpu_perf.py 6,1920,1000 3,1920,400 2>&1 | grep -v "^$" | awk 'NR%2==0'; done > gpurun_out/r03a/ab_accum.log 2>&1
cat gpurun_out/r03a/ab_accum.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03a/trace -- python3 bench.py --spp 1000 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03a/trace.log 2>&1
python3 tools/trace_tail.py gpurun_out/r03a/trace gpurun_out/r03a/tail_1000spp.json
find gpurun_out/r03a/trace -name "*.csv" -size +1M -delete
