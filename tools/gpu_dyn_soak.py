import importlib, os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import random_scene
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
ctx = pt.Context(0)
bad = 0
for seed in range(300, 340):
    spec = random_scene(seed, sphere_light=False, n_objects=6 + seed % 9)
    gs = pt.Scene(ctx); gres = spec.replay(gs); gcam = spec.make_camera(pt.Camera, gres)
    ref, st1 = gs.render(gcam, seed, 0, 12, slots_per_pixel=1)
    for pool in ("", "777", "5000"):
        if pool: os.environ["PT_POOL_SLOTS"] = pool
        else: os.environ.pop("PT_POOL_SLOTS", None)
        dyn, st = gs.render(gcam, seed, 0, 12)
        fin = np.isfinite(ref)
        ok = st.samples == st1.samples and st.segments == st1.segments and np.allclose(dyn[fin], ref[fin], rtol=1e-10, atol=1e-10) and np.array_equal(np.isfinite(dyn), fin)
        bad += 0 if ok else 1
        if not ok: print(seed, pool, "MISMATCH", st.samples, st1.samples, st.segments, st1.segments, np.nanmax(np.abs(dyn - ref)))
    gs.close()
os.environ.pop("PT_POOL_SLOTS", None)
print("dynamic soak mismatches:", bad)
