"""A frame rendered on path pools of different sizes (PT_POOL_SLOTS, dynamic work assignment) is the same frame: equal sample and
segment counts, pixel sums equal up to f64 addition order. Usage: PT_EXPERIMENT=1 python tools/gpu_pool_equiv.py scene width spp pool [pool ...]
(pool 0 = the library's own rule). Used for the deep pools of the round-3 rule (134 M / 268 M slots), which no test reaches."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
sid, width, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = pt.Context(0)
gs = pt.Scene(ctx)
cam = gs.build_scene(sid, width, spp)
ref = None
for pool in sys.argv[4:]:
    if pool != "0": os.environ["PT_POOL_SLOTS"] = pool
    else: os.environ.pop("PT_POOL_SLOTS", None)
    acc, st = gs.render(cam, 1, 0, spp)
    line = f"pool {pool}: n_slots {st.n_slots} samples {st.samples} segments {st.segments} iterations {st.iterations} compactions {st.compactions} {st.ms_total:.1f} ms"
    if ref is None:
        ref = (acc, st.samples, st.segments)
    else:
        rel = float(np.max(np.abs(acc - ref[0]) / np.maximum(np.abs(ref[0]), 1e-300)))
        ok = st.samples == ref[1] and st.segments == ref[2] and rel < 1e-10 and np.isfinite(acc).all()
        line += f"  max rel diff vs first {rel:.2e}  {'OK' if ok else 'MISMATCH'}"
    print(line, flush=True)
gs.close()
