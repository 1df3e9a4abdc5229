cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f
bash tools/run_bytes.sh r03f_stage > gpurun_out/r03f/bytes_stage.log 2>&1
PT_AMD_LIB=$GRAFT_REPO_ROOT/thu-acg-f2024-path-tracer_amd/variants/libpt_amd_nt.so bash tools/run_bytes.sh r03f_nt > gpurun_out/r03f/bytes_nt.log 2>&1
tail -n 3 gpurun_out/r03f/bytes_stage.log; tail -n 3 gpurun_out/r03f/bytes_nt.log
