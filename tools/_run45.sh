cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03cand; O=gpurun_out/r03cand
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur c1024 c512 2>&1 | tee -a $O/ab.log
SPEC=6,1920,250 ROUNDS=1 bash tools/ab_perf.sh cur c1024 c512 2>&1 | tee -a $O/ab.log
