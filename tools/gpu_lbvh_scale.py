"""The GPU BVH builder at scale: a lumpy icosphere(8) = 1,310,720 triangles. Build time of the depth-bounded
device LBVH against the host's binned SAH, the depth reached, and what the tree costs to traverse (K2 ms per launch on the same
frame). PT_LBVH_DEPTH (experiment switch) sets the depth budget: 20 / default (bal + 4 <= 23) / 26 (beyond k_extend2's stacks:
the batch kernel runs). Usage: python tools/gpu_lbvh_scale.py [subdiv=8] -> one JSON line per variant."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import icosphere
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
sub = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P, I = icosphere(sub)
P = (P * (1.0 + 0.12 * np.sin(9.0 * P[:, [0]]) * np.cos(7.0 * P[:, [1]]) + 0.05 * np.sin(31.0 * P[:, [2]]))).astype(np.float32)
ctx = pt.Context(0)
os.environ["PT_EXPERIMENT"] = "1"
ref = None
variants = [("host_sah", False, None, None), ("lbvh_default", True, None, None), ("lbvh_depth20", True, "20", None), ("lbvh_depth26", True, "26", None)]
if len(sys.argv) > 2:      # extra experiment variants: name:depth:median_below,...
    variants = [("host_sah", False, None, None)] + [(v.split(":")[0], True, v.split(":")[1] or None, v.split(":")[2] or None) for v in sys.argv[2].split(",")]
for name, device, depth, median in variants:
    if depth: os.environ["PT_LBVH_DEPTH"] = depth
    else: os.environ.pop("PT_LBVH_DEPTH", None)
    if median: os.environ["PT_LBVH_MEDIAN"] = median
    else: os.environ.pop("PT_LBVH_MEDIAN", None)
    s = pt.Scene(ctx)
    s.set_device_bvh_threshold(1000 if device else 0)
    m = s.mat_metal(s.tex_solid_rgb(0.9, 0.8, 0.6), s.tex_solid_f(0.2))
    s.world_add_object(s.mesh(1.0, P, I, None, None, m))
    s.world_add_object(s.quad((-6.0, -1.4, -6.0), (0.0, 0.0, 12.0), (12.0, 0.0, 0.0), s.mat_diffuse(s.tex_solid_rgb(0.7, 0.7, 0.7), -1)))
    t = time.time(); s.world_build(); t_build = time.time() - t
    n_dev, d = s.device_bvh_info()
    cam = pt.Camera(); cam.aspect_ratio = 1.0; cam.image_width = 1024; cam.vfov = 40; cam.max_depth = 50
    cam.look_from[:] = (0.0, 1.2, -3.6); cam.look_at[:] = (0.0, 0.0, 0.0); cam.vup[:] = (0, 1, 0); cam.focal_length = 3.0; cam.blur_strength = 0.5
    cam.env_color[:] = (0.5, 0.6, 0.8); cam.env_tex = -1
    s.render(cam, 1, 0, 8)
    acc, st = s.render(cam, 1, 0, 64, profile=True)
    if ref is None: ref = acc
    same = bool(np.allclose(acc, ref, rtol=1e-12, atol=1e-12))
    print(json.dumps({"variant": name, "triangles": len(I) // 3, "build_s": round(t_build, 3), "device_built": n_dev, "device_depth": d,
                      "k2": "k_extend2" if st.extend_variant == 0 else "k_extend (batch)", "k2_ms_per_launch": round(st.ms_extend / st.launches_extend, 4),
                      "k3_ms_per_launch": round(st.ms_shade / st.launches_shade, 4), "frame_ms": round(st.ms_total, 1), "segments": st.segments,
                      "same_frame_as_host_tree": same}), flush=True)
    s.close()
