cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03a
bash tools/run_bytes.sh r03a_tiled > gpurun_out/r03a/bytes_tiled.log 2>&1
PT_EXPERIMENT=1 PT_ACCUM_LINEAR=1 bash tools/run_bytes.sh r03a_linear > gpurun_out/r03a/bytes_linear.log 2>&1
tail -n 4 gpurun_out/r03a/bytes_tiled.log; tail -n 4 gpurun_out/r03a/bytes_linear.log
