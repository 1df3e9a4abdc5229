cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03w64; O=gpurun_out/r03w64
E="PT_EXPERIMENT=1"
for r in 1 2; do SPEC=6,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_EXT2=1164" "$E PT_EXT2=2164" 2>&1 | tee -a $O/ab.log; done
SPEC=6,1920,1000 bash tools/env_sweep.sh "$E" "$E PT_EXT2=1164" "$E PT_EXT2=2164" 2>&1 | tee -a $O/ab.log
SPEC=6,1920,250 bash tools/env_sweep.sh "$E" "$E PT_EXT2=1164" "$E PT_EXT2=2164" 2>&1 | tee -a $O/ab.log
