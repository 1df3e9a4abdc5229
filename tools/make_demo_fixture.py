"""Coarse fixtures from the reference's own output artifacts (demo/*.png, 1920x1080 RGB8, unknown seed):
block means of the gamma-space image on a 48x27 grid (40x40-pixel blocks). Data only — the images
themselves stay in /root/reference. Writes tests/golden/reference_demo_blocks.npz.

    python tools/make_demo_fixture.py /root/reference/demo
"""
import os, sys
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = {1: "balls.png", 2: "earth.png", 4: "lights.png", 5: "bsdf.png", 6: "scene6.png"}   # main.rs:81,131,273,368,531
out = {}
for sid, name in SCENES.items():
    img = np.asarray(Image.open(os.path.join(sys.argv[1], name)).convert("RGB"), dtype=np.float64) / 255.0
    assert img.shape == (1080, 1920, 3)
    out[f"scene{sid}"] = img.reshape(27, 40, 48, 40, 3).mean(axis=(1, 3)).astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_demo_blocks.npz"), **out)
print({k: v.shape for k, v in out.items()})
