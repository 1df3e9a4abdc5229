cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03tt; O=gpurun_out/r03tt
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur t4 t2s2 t4s2 k3t 2>&1 | tee -a $O/ab.log
SPEC=6,1920,500 ROUNDS=1 bash tools/ab_perf.sh cur t4 t2s2 t4s2 k3t 2>&1 | tee -a $O/ab.log
