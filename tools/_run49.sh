cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03ch4; O=gpurun_out/r03ch4
E="PT_EXPERIMENT=1"
PT_EXPERIMENT=1 PT_EXT2=4164 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden or closest_hit" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
for r in 1 2; do SPEC=6,1920,1000 bash tools/env_sweep.sh "$E" "$E PT_EXT2=4164" "$E PT_EXT2=1164" 2>&1 | tee -a $O/ab.log; done
SPEC=6,1920,250 bash tools/env_sweep.sh "$E" "$E PT_EXT2=4164" "$E PT_EXT2=1164" 2>&1 | tee -a $O/ab.log
SPEC=6,1920,4000 bash tools/env_sweep.sh "$E" "$E PT_EXT2=4164" "$E PT_EXT2=1164" 2>&1 | tee -a $O/ab.log
