# usage (on the GPU box): bash tools/_ab.sh <tag> "<variants>" [specs...]   — quick parity subset, then interleaved A/B
TAG=$1; VARS=$2; shift; shift
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_exact or closest_hit or golden or axis_parallel or tie_rule or kernel_forms" > gpurun_out/$TAG/pytest.log 2>&1; rc=$?
tail -2 gpurun_out/$TAG/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/$TAG/pytest.log | head -20; exit 1; }
for spec in "${@:-6,1920,1000}"; do
  SPEC=$spec ROUNDS=2 bash tools/ab_perf.sh $VARS 2>&1 | tee -a gpurun_out/$TAG/ab.log
done
