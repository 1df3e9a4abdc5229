cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03p; O=gpurun_out/r03p
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_exact or closest_hit or golden or axis_parallel or tie_rule or kernel_forms or free_placement or nested" > $O/pytest.log 2>&1; rc=$?
tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit 1; }
PT_AMD_LIB=$PWD/thu-acg-f2024-path-tracer_amd/variants/libpt_amd_p3.so timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden" > $O/pytest_p3.log 2>&1; echo "p3 rc=$?"; tail -1 $O/pytest_p3.log
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh p0 p1 p2 p3 p2u 2>&1 | tee -a $O/ab.log
SPEC=7,1920,200 ROUNDS=1 bash tools/ab_perf.sh p0 p1 p2 2>&1 | tee -a $O/ab.log
SPEC=3,1920,200 ROUNDS=1 bash tools/ab_perf.sh p0 p1 2>&1 | tee -a $O/ab.log
SPEC=5,1920,400 ROUNDS=1 bash tools/ab_perf.sh p0 p1 2>&1 | tee -a $O/ab.log
for r in 1 2; do SPEC=6,1920,1000 bash tools/env_sweep.sh "PT_EXPERIMENT=1 PT_EXT2=164" "PT_EXPERIMENT=1 PT_EXT2=1164" "PT_EXPERIMENT=1 PT_EXT2=2164" "PT_EXPERIMENT=1 PT_EXT2=8164" 2>&1 | tee -a $O/ab.log; done
PT_EXPERIMENT=1 PT_EXT2=2164 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden" > $O/pytest_2164.log 2>&1; echo "2164 rc=$?"; tail -1 $O/pytest_2164.log
PT_EXPERIMENT=1 PT_EXT2=1164 timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bit_exact or golden" > $O/pytest_1164.log 2>&1; echo "1164 rc=$?"; tail -1 $O/pytest_1164.log
