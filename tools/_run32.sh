cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03pool; O=gpurun_out/r03pool
E="PT_EXPERIMENT=1"
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E PT_POOL_SLOTS=67108864" "$E PT_POOL_SLOTS=100663296" "$E PT_POOL_SLOTS=134217728" 2>&1 | tee -a $O/ab2.log
for r in 1 2; do SPEC=5,3840,1000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=67108864" "$E PT_POOL_SLOTS=134217728" 2>&1 | tee -a $O/ab2.log; done
SPEC=3,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=67108864" "$E PT_POOL_SLOTS=134217728" 2>&1 | tee -a $O/ab2.log
SPEC=6,1920,2000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=50331648" "$E PT_POOL_SLOTS=67108864" 2>&1 | tee -a $O/ab2.log
