cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03u; O=gpurun_out/r03u
timeout -k 10 400 python tools/gpu_fuzz.py 5000 5400 > $O/fuzz_mesh.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_mesh.log
timeout -k 10 400 python tools/gpu_fuzz.py 6000 6400 nomesh > $O/fuzz_nomesh.log 2>&1; echo "fuzz nomesh rc=$?"; tail -2 $O/fuzz_nomesh.log
timeout -k 10 300 python tools/gpu_fuzz_scenes.py > $O/fuzz_scenes.log 2>&1; echo "scenes rc=$?"; tail -2 $O/fuzz_scenes.log
timeout -k 10 300 python tools/gpu_dyn_soak.py > $O/dyn.log 2>&1; echo "dyn rc=$?"; tail -1 $O/dyn.log
