cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03at; O=gpurun_out/r03at
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_exact or golden or kernel_forms or compaction or eight_rank or slot_layouts or furnace or full_hd" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh base cur 2>&1 | tee -a $O/ab.log
SPEC=6,1920,250 ROUNDS=2 bash tools/ab_perf.sh base cur 2>&1 | tee -a $O/ab.log
SPEC=6,960,100 ROUNDS=2 bash tools/ab_perf.sh base cur 2>&1 | tee -a $O/ab.log
SPEC=3,1920,200 ROUNDS=1 bash tools/ab_perf.sh base cur 2>&1 | tee -a $O/ab.log
SPEC=6,1920,4000 ROUNDS=1 bash tools/ab_perf.sh base cur 2>&1 | tee -a $O/ab.log
