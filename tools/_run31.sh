cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03pool; O=gpurun_out/r03pool
E="PT_EXPERIMENT=1"
for r in 1 2; do
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_POOL_SLOTS=50331648" "$E PT_POOL_SLOTS=67108864" "$E PT_POOL_SLOTS=25165824" 2>&1 | tee -a $O/ab.log
done
