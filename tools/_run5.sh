cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03h
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s -k "device_bvh" > gpurun_out/r03h/pytest.log 2>&1; echo rc=$?
tail -6 gpurun_out/r03h/pytest.log
timeout -k 10 500 python tools/gpu_lbvh_scale.py 8 > gpurun_out/r03h/lbvh_scale.jsonl 2> gpurun_out/r03h/lbvh_scale.err; echo rc=$?
cat gpurun_out/r03h/lbvh_scale.jsonl; tail -3 gpurun_out/r03h/lbvh_scale.err
