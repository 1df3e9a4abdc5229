"""Bring-up: find samples whose radiance differs between GPU and oracle and dump the oracle's path."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
import oracle_py as orc
from tools.gpu_trial import images_for
np.set_printoptions(precision=17, linewidth=250)
ctx = pt.Context(0)
orc.set_math_mode(os.environ.get('ORC_LIBM', '0') != '1')
sid, w, nsamp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-9
gs = pt.Scene(ctx); gcam = gs.build_scene(sid, w, nsamp)
os_ = orc.Scene(); ocam = os_.build_scene(sid, w, nsamp, images=images_for(sid))
shown = 0
for s in range(nsamp):
    ga, _ = gs.render(gcam, 1, s, s + 1, slots_per_pixel=1)
    oa, _ = os_.render(ocam, 1, s, s + 1)
    rel = np.abs(ga - oa) / np.maximum(np.abs(oa), 1e-3)
    bad = np.argwhere(rel.max(axis=2) > tol)
    print(f"sample {s}: {len(bad)} pixels differ by > {tol} (of {ga.shape[0]*ga.shape[1]}), exact {np.mean(ga == oa):.4f}")
    for (y, x) in bad:
        if shown >= 12: break
        shown += 1
        pix = y * w + x
        rad, dump, n = os_.trace_sample(ocam, 1, pix, s)
        print(f"  pixel ({y},{x}) sample {s}: gpu {ga[y,x]} orc {oa[y,x]} nseg {n}")
        print("     prim ids:", [int(d[1]) for d in dump], " t:", [float(f'{d[0]:.6g}') for d in dump])
        print("     throughput:", [list(np.round(d[5:8], 6)) for d in dump])
