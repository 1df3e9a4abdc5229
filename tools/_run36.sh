cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03pool; O=gpurun_out/r03pool
E="PT_EXPERIMENT=1"
SPEC=3,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=268435456" 2>&1 | tee -a $O/ab4.log
SPEC=5,3840,1000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=268435456" 2>&1 | tee -a $O/ab4.log
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=268435456" 2>&1 | tee -a $O/ab4.log
