"""Bring-up: raw throughput of pt_render at a given size (no oracle)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
ctx = pt.Context(0)
for arg in sys.argv[1:]:
    sid, w, spp, k = (list(map(int, arg.split(","))) + [0])[:4]
    s = pt.Scene(ctx); cam = s.build_scene(sid, w, spp)
    for rep in range(2):
        acc, st = s.render(cam, 1, 0, spp, slots_per_pixel=k, profile=(rep == 1))
        H = acc.shape[0]
        print(f"scene {sid} {w}x{H}@{spp} k={st.slots_per_pixel} slots={st.n_slots}: {st.ms_total:.1f} ms {w*H*spp/st.ms_total/1e3:.1f} Msamples/s seg/sample {st.segments/st.samples:.3f} iters {st.iterations} "
              f"extend {st.ms_extend:.1f} ms shade {st.ms_shade:.1f} ms other {st.ms_other:.2f} blocks {st.blocks_extend}/{st.blocks_shade}", flush=True)
    s.close()
