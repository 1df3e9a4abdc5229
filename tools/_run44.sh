cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03k3b; O=gpurun_out/r03k3b
E="PT_EXPERIMENT=1"
PT_EXPERIMENT=1 PT_SHADE_VARIANT=52 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden" > $O/pytest52.log 2>&1; echo "rc=$?"; tail -1 $O/pytest52.log
for r in 1 2; do SPEC=6,1920,1000 bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=12" "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=52" "$E PT_SHADE_VARIANT=32" 2>&1 | tee -a $O/ab.log; done
SPEC=3,1920,200 bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=52" "$E PT_SHADE_VARIANT=32" 2>&1 | tee -a $O/ab.log
SPEC=5,1920,400 bash tools/env_sweep.sh "$E PT_SHADE_VARIANT=22" "$E PT_SHADE_VARIANT=52" "$E PT_SHADE_VARIANT=32" 2>&1 | tee -a $O/ab.log
