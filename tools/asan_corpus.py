"""Corrupt-file corpus for the host-side parsers, run under AddressSanitizer + UBSan by `make -C oracle asan` (CPU only).
argv[1] = the instrumented build of csrc/pt_assets.cpp. Every call must either succeed or return -1 with a message — a
sanitizer report aborts the process. The same corpus goes through the oracle's own loaders (ORACLE_LIB = its sanitizer build)."""
import ctypes as C, os, struct, sys, tempfile, zlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as orc

lib = C.CDLL(sys.argv[1])
u8pp, f32pp, u32pp, u32p = C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)
lib.pt_load_png_rgb8.argtypes = lib.pt_load_hdr_rgb8.argtypes = lib.pt_load_jpeg_rgb8.argtypes = [C.c_char_p, u8pp, u32p, u32p]
lib.pt_load_hdr_rgbf32.argtypes = [C.c_char_p, f32pp, u32p, u32p]
lib.pt_load_obj.argtypes = [C.c_char_p, f32pp, u32p, u32pp, u32p, f32pp, u32p]
lib.pt_load_obj_single_index.argtypes = [C.c_char_p, f32pp, u32p, u32pp, u32p, f32pp, u32p, f32pp, u32p]
lib.pt_save_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
lib.pt_free.argtypes = [C.c_void_p]
lib.pt_last_error.restype = C.c_char_p
olib = orc.lib
stats = {"ok": 0, "rejected": 0}


def image(fn, path, ptr_t):
    p, w, h = ptr_t(), C.c_uint32(), C.c_uint32()
    rc = fn(path.encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc == 0:
        n = w.value * h.value * 3
        a = np.ctypeslib.as_array(p, (n,)).copy() if n else None       # touch every decoded byte
        lib.pt_free(p) if fn.__name__.startswith("pt_") else olib.orc_free(p)
        stats["ok"] += 1
        return a
    stats["rejected"] += 1
    assert rc == -1
    return None


def obj(path):
    for single in (False, True):
        pos, idx, uv, nrm = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
        npos, nidx, nuv, nn = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        if single: rc = lib.pt_load_obj_single_index(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(nrm), C.byref(nn), C.byref(uv), C.byref(nuv))
        else: rc = lib.pt_load_obj(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(uv), C.byref(nuv))
        if rc == 0:
            if nidx.value:
                I = np.ctypeslib.as_array(idx, (nidx.value,))
                assert I.max() < max(npos.value, 1)
            for q in (pos, idx, uv) + ((nrm,) if single else ()): lib.pt_free(q)
            stats["ok"] += 1
        else:
            stats["rejected"] += 1
    pos, idx, uv = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
    npos, nidx, nuv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if olib.orc_load_obj(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(uv), C.byref(nuv)) == 0:
        for q in (pos, idx, uv): olib.orc_free(q)


def mutations(data, rng, n_trunc=24, n_flip=40):
    yield data
    for k in sorted(set(int(x) for x in np.linspace(0, len(data), n_trunc))): yield data[:k]
    for _ in range(n_flip):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(b))); b[i] ^= 1 << int(rng.integers(0, 8))
        yield bytes(b)
    for _ in range(8):
        b = bytearray(data); i = int(rng.integers(0, max(1, len(b) - 64))); b[i:i + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        yield bytes(b)


def png_chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def main():
    rng = np.random.default_rng(7)
    A = os.path.join(ROOT, "assets")
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "f")

    def put(b):
        with open(path, "wb") as f: f.write(b)
    # ---- PNG: a small file written by the instrumented encoder itself, then mutated; hostile headers with VALID CRCs
    px = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    assert lib.pt_save_png(path.encode(), 13, 9, px.ctypes.data) == 0
    good = open(path, "rb").read()
    back = image(lib.pt_load_png_rgb8, path, C.POINTER(C.c_uint8))
    assert back is not None and np.array_equal(back.reshape(9, 13, 3), px)
    for m in mutations(good, rng): put(m); image(lib.pt_load_png_rgb8, path, C.POINTER(C.c_uint8))
    sig = good[:8]
    for w, h, depth, ctype in ((0x7FFFFFFF, 0x7FFFFFFF, 8, 2), (1 << 20, 1 << 20, 16, 6), (5, 5, 16, 3), (5, 5, 3, 2), (0, 7, 8, 2), (70000, 70000, 1, 0)):
        ihdr = struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)
        put(sig + png_chunk(b"IHDR", ihdr) + png_chunk(b"IDAT", zlib.compress(b"\0" * 64)) + png_chunk(b"IEND", b""))
        assert image(lib.pt_load_png_rgb8, path, C.POINTER(C.c_uint8)) is None, (w, h, depth, ctype)
    for name in ("bricks/color.png",):
        data = open(os.path.join(A, name), "rb").read()
        for m in mutations(data, rng, n_trunc=6, n_flip=6): put(m); image(lib.pt_load_png_rgb8, path, C.POINTER(C.c_uint8))
    # ---- JPEG: the reference's baseline asset and a small progressive 4:2:0 file with restart markers, mutated (header bytes hit hardest)
    from PIL import Image
    jp = os.path.join(tmp, "p.jpg")
    Image.fromarray(rng.integers(0, 256, (40, 56, 3), dtype=np.uint8), "RGB").save(jp, "JPEG", quality=70, subsampling=2, progressive=True, restart_marker_blocks=2)
    for src, n_t, n_f in ((os.path.join(A, "earthmap.jpg"), 12, 30), (jp, 40, 200)):
        data = open(src, "rb").read()
        for m in mutations(data, rng, n_trunc=n_t, n_flip=n_f): put(m); image(lib.pt_load_jpeg_rgb8, path, C.POINTER(C.c_uint8))
        for _ in range(n_f):                                   # flips confined to the headers (tables, frame, scan parameters)
            b = bytearray(data)
            for _ in range(3): b[int(rng.integers(2, min(len(b), 700)))] = int(rng.integers(0, 256))
            put(bytes(b)); image(lib.pt_load_jpeg_rgb8, path, C.POINTER(C.c_uint8))
    # ---- Radiance HDR: the reference's probe, mutated; hand-made bad runs and sizes
    data = open(os.path.join(A, "grace_probe_latlong.hdr"), "rb").read()
    for m in mutations(data, rng, n_trunc=10, n_flip=12):
        put(m)
        a = image(lib.pt_load_hdr_rgbf32, path, C.POINTER(C.c_float)); image(lib.pt_load_hdr_rgb8, path, C.POINTER(C.c_uint8))
        b = image(olib.orc_load_hdr_rgbf32, path, C.POINTER(C.c_float)); image(olib.orc_load_hdr_rgb8, path, C.POINTER(C.c_uint8))
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))          # the two decoders agree, on rejects too
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n"
    for body in (b"-Y 2 +X 8\n" + bytes([2, 2, 0, 8]) + bytes([200, 1]) * 4,                 # run of 72 into a scanline of 8
                 b"-Y 2 +X 8\n" + bytes([2, 2, 0, 8]) + bytes([0]) * 40,                       # zero-length literal runs
                 b"-Y 1000000 +X 1000000\n", b"-Y -3 +X 8\n", b"+X 8 -Y 2\n", b"-Y 2 +X 8\n" + bytes(range(32)),
                 b"-Y 1 +X 40000\n" + bytes([2, 2, 0x9C, 0x40])):                              # width with the RLE marker's high bit set
        put(head + body)
        for fn, t in ((lib.pt_load_hdr_rgbf32, C.c_float), (lib.pt_load_hdr_rgb8, C.c_uint8), (olib.orc_load_hdr_rgbf32, C.c_float), (olib.orc_load_hdr_rgb8, C.c_uint8)):
            image(fn, path, C.POINTER(t))
    # ---- OBJ: out-of-range, negative, zero, absurd indices; malformed lines; a real mesh cut short
    for text in ("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n", "v 0 0 0\nf 1 2 4\n", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -3\nf -4 -1 -2\n", "f 1 2 3\n", "v 0 0 0\nf 0 0 0\n",
                 "v 0 0 0\nv 1 1 1\nv 2 2 2\nf 99999999999999999999 1 2\n", "v 1 2\nf 1 1 1\n", "v a b c\nv 0 0 0\nv 0 0 0\nf 1/5/9 2/-7/1 3//\n",
                 "vt 0 0\nvn 0 0 1\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/1/1 2/2/2 3/-5/-9\n", "v 0 0 0\n" * 3 + "f " + " ".join(["1"] * 5000) + "\n", ""):
        put(text.encode()); obj(path)
    data = open(os.path.join(A, "spot.obj"), "rb").read()
    for m in mutations(data, rng, n_trunc=5, n_flip=10): put(m); obj(path)
    print(f"asan corpus: {stats['ok']} decoded, {stats['rejected']} rejected cleanly, no sanitizer report")


if __name__ == "__main__":
    main()
