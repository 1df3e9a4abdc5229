"""Regenerates tests/golden/*.npz from the CPU oracle in deterministic-math mode.

The reference ships no golden vectors, so these fixtures pin the oracle against
itself across machines/compilers (pure IEEE arithmetic -> must reproduce exactly) and give the
GPU tests a committed target that does not need the oracle's code to be re-run."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as orc

orc.set_math_mode(True)
os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
for sid in (3, 6):
    s = orc.Scene()
    cam = s.build_scene(sid, 64, 16)
    acc, cnt = s.render(cam, 1, 0, 16)
    path = os.path.join(ROOT, "tests", "golden", f"scene{sid}_w64_spp16_seed1.npz")
    np.savez_compressed(path, accum=acc, segments=np.uint64(cnt["segments"]), scene=sid, width=64, spp=16, seed=1)
    print(path, acc.shape, cnt["segments"], acc.mean())
    s.close()
