cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03final8; O=gpurun_out/r03final8
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python tools/gpu_fuzz.py 7000 7200 > $O/fuzz_mesh.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_mesh.log
timeout -k 10 300 python tools/gpu_dyn_soak.py > $O/dyn.log 2>&1; echo "dyn rc=$?"; tail -1 $O/dyn.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_profiles.sh r03 > $O/prof.log 2>&1; grep -E "hbm_bytes_per_launch" $O/prof.log | head -4
python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc=$?"
for spp in 2000 1000 500; do python bench.py --spp $spp --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_share_spp$spp.json 2>> $O/share.err; echo "share $spp rc=$?"; done
python bench.py --scene 3 --width 1920 --steps 2 --warmup 1 > $O/bench_config2.json 2> $O/config2.err; echo "config2 rc=$?"
python bench.py --scene 5 --width 3840 --spp 1000 --steps 2 --warmup 1 > $O/bench_config5_shard.json 2> $O/config5.err; echo "config5 rc=$?"
for f in headline share_spp2000 share_spp1000 share_spp500 config2 config5_shard; do python3 -c "
import json
d=json.load(open('$O/bench_$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['other_kernel'], r['algorithmic_bytes_per_launch'], r['traffic'], d['config']['resident_paths'], d.get('cpu_baseline',{}).get('value'))"; done
