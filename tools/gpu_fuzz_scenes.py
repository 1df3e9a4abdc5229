"""One-off soak over the seven built-in scenes: several widths / sample ranges / seeds, GPU vs oracle (det math), bit-exact.
    python tools/gpu_fuzz_scenes.py"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as orc
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
orc.set_math_mode(True)
ctx = pt.Context(0)
bad = 0
for sid in range(1, 8):
    images = {n: pt.decode_image_rgb8(os.path.join(pt.ASSET_DIR, n)) for n in pt.SCENE_IMAGE_FILES.get(sid, [])}
    for width, lo, hi, seed in ((72, 0, 3, 2), (101, 2, 5, 3), (56, 1, 4, 12345678901)):
        gs, os_ = pt.Scene(ctx), orc.Scene()
        gcam, ocam = gs.build_scene(sid, width, hi), os_.build_scene(sid, width, hi, images=images)
        ga, st = gs.render(gcam, seed, lo, hi, slots_per_pixel=1)
        oa, cnt = os_.render(ocam, seed, lo, hi)
        same = np.array_equal(ga, oa, equal_nan=True) and st.segments == cnt["segments"]
        bad += 0 if same else 1
        print(sid, width, (lo, hi), seed, "ok" if same else "MISMATCH", st.segments, flush=True)
        gs.close(); os_.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
