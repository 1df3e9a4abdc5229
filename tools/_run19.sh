cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03s; O=gpurun_out/r03s
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
bash tools/run_pmc_ta.sh r03 1000 > $O/ta.log 2>&1; tail -12 $O/ta.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], r['kernel'], r['avg_launch_ms'], r['frac'], r['other_kernel'])"
