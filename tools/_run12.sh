cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03r; O=gpurun_out/r03r
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "compaction or kernel_forms or slot_layouts or eight_rank or sheen or mesh_crowd or full_hd or cornell_1920 or scene5_3840 or golden or demo_images" > $O/pytest.log 2>&1; echo rc=$?; tail -2 $O/pytest.log
PT_EXPERIMENT=1 timeout 300 python tools/gpu_dyn_soak.py 2>&1 | tail -1
for spec in 6,1920,1000 3,1920,200 5,1920,400; do SPEC=$spec ROUNDS=2 bash tools/ab_perf.sh base cur; done 2>&1 | sed 's/seg\/sample.*extend/ extend/' | cut -c1-150
