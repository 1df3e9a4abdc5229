cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03t; O=gpurun_out/r03t
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_exact or closest_hit or golden or axis_parallel or tie_rule or kernel_forms or free_placement or million or device_bvh" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh f0 cur 2>&1 | tee -a $O/ab.log
SPEC=6,1920,250 ROUNDS=1 bash tools/ab_perf.sh f0 cur 2>&1 | tee -a $O/ab.log
