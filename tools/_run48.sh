cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03drain; O=gpurun_out/r03drain
V=$PWD/thu-acg-f2024-path-tracer_amd/variants
PT_AMD_LIB=$V/libpt_amd_d1.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "bit_exact or golden or closest_hit or kernel_forms or million" > $O/pytest.log 2>&1; echo "rc=$?"; tail -1 $O/pytest.log
SPEC=6,1920,1000 ROUNDS=2 bash tools/ab_perf.sh cur d1 d1s 2>&1 | tee -a $O/ab.log
SPEC=6,1920,250 ROUNDS=1 bash tools/ab_perf.sh cur d1 d1s 2>&1 | tee -a $O/ab.log
SPEC=6,1920,4000 ROUNDS=1 bash tools/ab_perf.sh cur d1 2>&1 | tee -a $O/ab.log
