cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03soak; O=gpurun_out/r03soak
timeout -k 10 500 python tools/gpu_fuzz.py 20000 20800 > $O/fuzz_mesh.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_mesh.log
timeout -k 10 500 python tools/gpu_fuzz.py 30000 30800 nomesh > $O/fuzz_nomesh.log 2>&1; echo "fuzz nomesh rc=$?"; tail -1 $O/fuzz_nomesh.log
timeout -k 10 300 python tools/gpu_fuzz_scenes.py > $O/fuzz_scenes.log 2>&1; echo "scenes rc=$?"; tail -1 $O/fuzz_scenes.log
