cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03cand; O=gpurun_out/r03cand
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur c1152 2>&1 | tee -a $O/ab2.log
SPEC=6,1920,250 ROUNDS=1 bash tools/ab_perf.sh cur c1152 2>&1 | tee -a $O/ab2.log
for r in 1 2; do SPEC=6,1920,1000 bash tools/env_sweep.sh "PT_EXPERIMENT=1 PT_EXT2=1164" 2>&1 | tee -a $O/ab2.log; done
