cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03z; O=gpurun_out/r03z
PT_EXPERIMENT=1 PT_SHADE_VARIANT=32 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_exact or golden or compaction or eight_rank or slot_layouts" > $O/pytest32.log 2>&1; echo "rc=$?"; tail -2 $O/pytest32.log
for spec in 6,1920,1000 3,1920,200 5,1920,400; do
  for r in 1 2; do
    SPEC=$spec bash tools/env_sweep.sh "PT_EXPERIMENT=1 PT_SHADE_VARIANT=22" "PT_EXPERIMENT=1 PT_SHADE_VARIANT=32" 2>&1 | tee -a $O/ab.log
  done
done
