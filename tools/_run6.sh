cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i; O=gpurun_out/r03i
python bench.py --steps 6 --warmup 2 > $O/headline.json 2> $O/headline.err; echo "headline rc=$?"
for spp in 2000 1000 500; do python bench.py --spp $spp --steps 6 --warmup 2 --no-cpu-baseline > $O/share_spp$spp.json 2>> $O/share.err; echo "share $spp rc=$?"; done
python bench.py --scene 3 --width 1920 --steps 2 --warmup 1 --no-cpu-baseline > $O/config2.json 2> $O/config2.err; echo "config2 rc=$?"
python bench.py --scene 5 --width 3840 --spp 1000 --steps 2 --warmup 1 --no-cpu-baseline > $O/config5_shard.json 2> $O/config5.err; echo "config5 rc=$?"
python bench.py --gpus 1 --steps 1 --warmup 0 --spp 100 --no-cpu-baseline > $O/gpus1.json 2>&1; echo "gpus1 rc=$?"
for f in headline share_spp2000 share_spp1000 share_spp500 config2 config5_shard; do python3 -c "
import json,sys
d=json.load(open('$O/$f.json')); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['other_kernel'], d['config']['resident_paths'])"; done
