cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03pool; O=gpurun_out/r03pool
E="PT_EXPERIMENT=1"
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E PT_POOL_SLOTS=268435456" 2>&1 | tee -a $O/ab3.log
SPEC=6,1920,2000 bash tools/env_sweep.sh "$E PT_POOL_SLOTS=134217728" 2>&1 | tee -a $O/ab3.log
SPEC=6,1920,1000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=67108864" "$E PT_POOL_SLOTS=134217728" 2>&1 | tee -a $O/ab3.log
SPEC=6,1920,500 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=33554432" "$E PT_POOL_SLOTS=67108864" 2>&1 | tee -a $O/ab3.log
SPEC=6,1920,250 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=16777216" "$E PT_POOL_SLOTS=33554432" 2>&1 | tee -a $O/ab3.log
SPEC=5,1920,400 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=33554432" "$E PT_POOL_SLOTS=67108864" 2>&1 | tee -a $O/ab3.log
SPEC=3,1920,200 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=16777216" "$E PT_POOL_SLOTS=33554432" 2>&1 | tee -a $O/ab3.log
