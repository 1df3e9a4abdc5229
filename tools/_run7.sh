cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03l; O=gpurun_out/r03l
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
rm -f assets/*.rgb8 assets/bricks/*.rgb8
( cd thu-acg-f2024-path-tracer_amd && ./pt_render -s 5 --width 240 --spp 16 --out /tmp/s5.png --assets ../assets ) > $O/cli_s5.log 2>&1; echo "cli s5 rc=$?"; tail -2 $O/cli_s5.log
( cd thu-acg-f2024-path-tracer_amd && ./pt_render -s 2 --width 240 --spp 16 --out /tmp/s2.png --assets ../assets ) > $O/cli_s2.log 2>&1; echo "cli s2 rc=$?"; tail -2 $O/cli_s2.log
( cd thu-acg-f2024-path-tracer_amd && ./pt_render -s 6 --width 240 --spp 16 --out /tmp/s6f.png --assets ../assets --float-hdr ) > $O/cli_s6f.log 2>&1; echo "cli s6 float rc=$?"; tail -2 $O/cli_s6f.log
python - <<'PY'
import importlib, numpy as np
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
for n in ("s5", "s2", "s6f"):
    a = pt.load_png_rgb8(f"/tmp/{n}.png"); print(n, a.shape, a.mean(axis=(0, 1)).round(2))
PY
