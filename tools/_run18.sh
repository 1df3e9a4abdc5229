cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03r; O=gpurun_out/r03r
E="PT_EXPERIMENT=1"
for spec in 6,1920,500 5,1920,800 5,3840,250 3,1920,800 6,1920,2000; do
  SPEC=$spec bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=32" "$E PT_SHADE_VARIANT=42" 2>&1 | tee -a $O/ab.log
done
SPEC=6,1920,1000 bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=42" "$E PT_SHADE_VARIANT=42 PT_WIDE_WINDOW_MIN=8" 2>&1 | tee -a $O/ab.log
SPEC=5,1920,400 bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=42" "$E PT_SHADE_VARIANT=42 PT_WIDE_WINDOW_MIN=8" 2>&1 | tee -a $O/ab.log
