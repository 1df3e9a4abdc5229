cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03cmp; O=gpurun_out/r03cmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "compaction or kernel_forms or eight_rank or slot_layouts or golden or furnace" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
timeout -k 10 300 python tools/gpu_dyn_soak.py 2>&1 | tail -1
PT_EXPERIMENT=1 timeout -k 10 300 python tools/gpu_pool_equiv.py 6 1920 1000 16777216 0 268435456 2>&1 | tee $O/equiv_scene6.log
PT_EXPERIMENT=1 timeout -k 10 300 python tools/gpu_pool_equiv.py 3 1920 1200 33554432 268435456 2>&1 | tee $O/equiv_scene3.log
for r in 1 2; do SPEC=6,1920,4000 bash tools/env_sweep.sh "PT_EXPERIMENT=1" 2>&1 | tee -a $O/ab.log; done
SPEC=3,1920,4000 bash tools/env_sweep.sh "PT_EXPERIMENT=1" 2>&1 | tee -a $O/ab.log
SPEC=5,3840,1000 bash tools/env_sweep.sh "PT_EXPERIMENT=1" 2>&1 | tee -a $O/ab.log
SPEC=6,1920,500 bash tools/env_sweep.sh "PT_EXPERIMENT=1" 2>&1 | tee -a $O/ab.log
