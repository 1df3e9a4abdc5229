cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03w; O=gpurun_out/r03w
V=$PWD/thu-acg-f2024-path-tracer_amd/variants
PT_EXPERIMENT=1 PT_PROF=1 PT_AMD_LIB=$V/libpt_amd_stamps.so timeout -k 10 300 python tools/gpu_perf.py 3,1920,200 > $O/stamps_scene3.log 2>&1; echo "rc=$?"; tail -30 $O/stamps_scene3.log
PT_EXPERIMENT=1 PT_PROF=1 PT_AMD_LIB=$V/libpt_amd_stamps.so timeout -k 10 300 python tools/gpu_perf.py 6,1920,1000 > $O/stamps_scene6.log 2>&1; echo "rc=$?"; tail -30 $O/stamps_scene6.log
