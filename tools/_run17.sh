cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03q; O=gpurun_out/r03q
V=$PWD/thu-acg-f2024-path-tracer_amd/variants
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur next pin ur 2>&1 | tee -a $O/ab.log
for r in 1 2; do SPEC=6,1920,1000 bash tools/env_sweep.sh "PT_EXPERIMENT=1 PT_EXT2=164" "PT_EXPERIMENT=1 PT_EXT2=1164" "PT_EXPERIMENT=1 PT_EXT2=1164 PT_AMD_LIB=$V/libpt_amd_rf8.so" "PT_EXPERIMENT=1 PT_EXT2=1164 PT_AMD_LIB=$V/libpt_amd_rf32.so" "PT_EXPERIMENT=1 PT_EXT2=1164 PT_AMD_LIB=$V/libpt_amd_ur.so" 2>&1 | tee -a $O/ab.log; done
SPEC=6,1920,250 bash tools/env_sweep.sh "PT_EXPERIMENT=1 PT_EXT2=164" "PT_EXPERIMENT=1 PT_EXT2=1164" 2>&1 | tee -a $O/ab.log
