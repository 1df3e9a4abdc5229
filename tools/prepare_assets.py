"""Writes "<file>.rgb8" sidecars ("PTRGB8 <w> <h>\n" + raw RGB8) next to the JPEG/PNG assets so
that the C++ CLI (host/main.cpp), which has no JPEG/PNG decoder, can run scenes 2, 5 and 7.
Python callers do not need this: the binding hands decoded pixels over (pt_register_image)."""
import os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
assets = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "assets")
for name in ("earthmap.jpg", "envmap.jpg", "bricks/color.png", "bricks/normal.png"):
    p = os.path.join(assets, name)
    img = np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8))
    with open(p + ".rgb8", "wb") as f:
        f.write(f"PTRGB8 {img.shape[1]} {img.shape[0]}\n".encode())
        f.write(img.tobytes())
    print(p + ".rgb8", img.shape)
