cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03x; O=gpurun_out/r03x
E="PT_EXPERIMENT=1"
for r in 1 2; do
SPEC=6,1920,1000 bash tools/env_sweep.sh "$E" "$E PT_SAH_BINS=32" "$E PT_SAH_BINS=64" "$E PT_SAH_SWEEP=256" "$E PT_SAH_SWEEP=4096 PT_SAH_BINS=32" "$E PT_SAH_SWEEP=1000000" 2>&1 | tee -a $O/ab.log
done
PT_EXPERIMENT=1 PT_SAH_SWEEP=1000000 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden or closest_hit" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
