"""Diagnose one fuzz seed (tools/gpu_fuzz.py): the differing pixels, the oracle's per-sample path records next to the
GPU's per-sample radiance, and a field-by-field comparison of the intersect probes on 20 000 rays.
    python tools/gpu_fuzz_one.py 27"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from common import random_scene
import oracle_py as orc
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
orc.set_math_mode(True)
ctx = pt.Context(0)
seed = int(sys.argv[1])
spec = random_scene(seed, sphere_light=(seed % 2 == 1), n_objects=6 + seed % 9)
os_ = orc.Scene(); ores = spec.replay(os_); ocam = spec.make_camera(orc.Camera, ores)
oa, cnt = os_.render(ocam, 100 + seed, 0, 5)
for env in ({},):
    old = {k: os.environ.get(k) for k in env}; os.environ.update(env)
    gs = pt.Scene(ctx); gres = spec.replay(gs); gcam = spec.make_camera(pt.Camera, gres)
    ga, st = gs.render(gcam, 100 + seed, 0, 5, slots_per_pixel=1)
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    diff = ~((ga == oa) | (np.isnan(ga) & np.isnan(oa)))
    px = np.argwhere(diff.any(axis=2))
    print(env, "segments", st.segments, cnt["segments"], "differing pixels", len(px), px[:5].tolist(), flush=True)
    if len(px):
        y, x = px[0]; W = ga.shape[1]
        print("   gpu", ga[y, x], "orc", oa[y, x])
        for smp in range(5):
            rad, dump, n = os_.trace_sample(ocam, 100 + seed, int(y * W + x), smp)
            g1, s1 = gs.render(gcam, 100 + seed, smp, smp + 1, slots_per_pixel=1)
            print("   sample", smp, "orc rad", rad.tolist(), "gpu", g1[y, x].tolist(), "segs", n, "prims", dump[:, 1].astype(int).tolist()[:12], "t", np.round(dump[:, 0], 6).tolist()[:12])
    # many rays at the meshes: compare the intersect probes field by field
    rng = np.random.default_rng(5)
    o = np.tile(np.array([gcam.look_from[0], gcam.look_from[1], gcam.look_from[2]]), (20000, 1)) + rng.normal(0, 0.05, (20000, 3))
    tgt = rng.uniform([-3, 0, -3], [3, 2, 4], (20000, 3))
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, rng.uniform(0, 1, (20000, 1))], axis=1)
    go, oo = gs.intersect(rays), os_.intersect(rays)
    neq = ~((go == oo) | (np.isnan(go) & np.isnan(oo)))
    rows = np.argwhere(neq.any(axis=1)).ravel()
    print("intersect probe: rows differing", len(rows), "of", len(rays), "columns", np.argwhere(neq.any(axis=0)).ravel().tolist())
    for r_ in rows[:3]:
        print("   ray", r_, "gpu", [float.hex(v) for v in go[r_]], "\n        orc", [float.hex(v) for v in oo[r_]])
    gs.close()
os_.close()
