cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03cmp; O=gpurun_out/r03cmp
E="PT_EXPERIMENT=1"
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=268435456" "$E PT_COMPACT_AT=70" "$E PT_COMPACT_AT=30" 2>&1 | tee -a $O/ab2.log
SPEC=5,3840,1000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=268435456" 2>&1 | tee -a $O/ab2.log
SPEC=6,1920,1000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=134217728" "$E PT_COMPACT_AT=70" 2>&1 | tee -a $O/ab2.log
SPEC=6,1920,500 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=67108864" "$E PT_COMPACT_AT=70" 2>&1 | tee -a $O/ab2.log
