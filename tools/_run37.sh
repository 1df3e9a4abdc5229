cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03big; O=gpurun_out/r03big
PT_EXPERIMENT=1 timeout -k 10 300 python tools/gpu_pool_equiv.py 3 1920 1200 33554432 0 268435456 2>&1 | tee $O/equiv_scene3.log
PT_EXPERIMENT=1 timeout -k 10 300 python tools/gpu_pool_equiv.py 6 1920 1000 16777216 0 134217728 268435456 2>&1 | tee $O/equiv_scene6.log
python bench.py --scene 3 --width 1920 --steps 2 --warmup 1 > $O/bench_config2.json 2> $O/config2.err; echo "config2 rc=$?"
python3 -c "
import json
d=json.load(open('$O/bench_config2.json')); r=d['roofline']
print('config2', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['other_kernel'], d['config']['resident_paths'], d.get('cpu_baseline',{}).get('value'), d.get('frame_check'))"
