"""north_star's tolerance measured directly: the HIP path against the FAITHFUL oracle (platform libm, what the
Rust reference calls through f64::sin etc.) at the full 4000 spp on a reduced frame.
    python tools/gpu_tolerance.py scene,width,spp ...
Prints per-channel RMSE of the linear mean image, the signed bias, and how many pixels differ at all."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
pt = importlib.import_module("thu-acg-f2024-path-tracer_amd")
import oracle_py as orc


def measure(ctx, sid, width, spp, seed=1, det=False):
    images = {n: pt.decode_image_rgb8(os.path.join(pt.ASSET_DIR, n)) for n in pt.SCENE_IMAGE_FILES.get(sid, [])}
    gs = pt.Scene(ctx); gcam = gs.build_scene(sid, width, spp)
    os_ = orc.Scene(); ocam = os_.build_scene(sid, width, spp, images=images)
    orc.set_math_mode(det)
    t = time.time(); ga, st = gs.render(gcam, seed, 0, spp); tg = time.time() - t            # default (dynamic) schedule
    t = time.time(); oa, cnt = os_.render(ocam, seed, 0, spp); to = time.time() - t
    orc.set_math_mode(False)
    gs.close(); os_.close()
    fin = np.isfinite(ga).all(axis=2) & np.isfinite(oa).all(axis=2)
    d = (ga - oa)[fin] / spp
    res = {"scene": sid, "width": width, "height": ga.shape[0], "spp": spp, "oracle_math": "det" if det else "libm",
           "rmse": np.sqrt(np.mean(d ** 2, axis=0)).tolist(), "bias": np.mean(d, axis=0).tolist(), "max_abs": float(np.abs(d).max()),
           "pixels_differing": float(np.mean(np.any(d != 0, axis=1))), "nonfinite_pixels_gpu_orc": [int((~np.isfinite(ga).all(axis=2)).sum()), int((~np.isfinite(oa).all(axis=2)).sum())],
           "mean_level": np.mean(oa[fin] / spp, axis=0).tolist(), "segments_gpu_orc": [int(st.segments), int(cnt["segments"])],
           "gpu_s": round(tg, 2), "oracle_s": round(to, 2)}
    return res


if __name__ == "__main__":
    ctx = pt.Context(0)
    for arg in sys.argv[1:]:
        f = arg.split(",")
        r = measure(ctx, int(f[0]), int(f[1]), int(f[2]), det=(len(f) > 3 and f[3] == "det"))
        print(json.dumps(r), flush=True)
