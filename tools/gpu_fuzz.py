"""One-off soak: random scenes (tests/common.random_scene) GPU vs oracle (det math), bit-exact, for a seed range.
    python tools/gpu_fuzz.py 4 40 [cam] [nomesh]     (nomesh: spheres / quads / cuboids only -> the batch K2 and its pair passes)
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from common import random_scene
import oracle_py as orc
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
orc.set_math_mode(True)
ctx = pt.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    spec = random_scene(seed, with_mesh="nomesh" not in sys.argv[3:], sphere_light=(seed % 2 == 1), n_objects=6 + seed % (16 if "nomesh" in sys.argv[3:] else 9))
    if "cam" in sys.argv[3:]:        # also randomise the camera
        r = np.random.default_rng(seed)
        spec.camera.update(vfov=float(r.uniform(20, 95)), defocus_angle=float(r.uniform(0, 3)), focal_length=float(r.uniform(3, 9)),
                           aspect_ratio=float(r.choice([1.0, 1.5, 0.7, 16 / 9])), image_width=int(r.integers(33, 80)),
                           look_from=tuple(r.uniform([-3, 0.5, -7], [3, 3, -4])), blur_strength=float(r.uniform(0, 1)),
                           max_depth=int(r.choice([3, 8, 50])))
    gs, os_ = pt.Scene(ctx), orc.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(orc.Camera, ores)
    ga, st = gs.render(gcam, 100 + seed, 0, 5, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 100 + seed, 0, 5)
    same = np.array_equal(ga, oa, equal_nan=True) and st.segments == cnt["segments"]
    bad += 0 if same else 1
    print(seed, "ok" if same else "MISMATCH", st.segments, "extend_variant", st.extend_variant, flush=True)
    gs.close(); os_.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
