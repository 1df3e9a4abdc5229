cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03z3; O=gpurun_out/r03z3
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur warm2 2>&1 | tee -a $O/ab.log
