"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding of oracle/liboracle.so (the f64 CPU
restatement of the reference's integrator). Importable only from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg — never from the product package.

The Scene class exposes the same method names as the product's Scene so that one scene
description can be replayed onto both.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")   # ORACLE_LIB: the sanitizer build (make -C oracle asan)
ASSET_DIR = os.path.join(os.path.dirname(_HERE), "assets")


class OracleError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [
        ("aspect_ratio", C.c_double),
        ("image_width", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("env_is_map", C.c_uint32),
        ("vfov", C.c_double),
        ("look_from", C.c_double * 3),
        ("look_at", C.c_double * 3),
        ("vup", C.c_double * 3),
        ("blur_strength", C.c_double),
        ("focal_length", C.c_double),
        ("defocus_angle", C.c_double),
        ("env_color", C.c_double * 3),
        ("env_tex", C.c_int32),
        ("_pad", C.c_int32),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise OracleError(f"{LIB_PATH} missing: run `make -C {_HERE}`")
    lib = C.CDLL(LIB_PATH)
    d3 = C.POINTER(C.c_double)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_destroy.argtypes = [C.c_void_p]
    lib.orc_scene_destroy.restype = None
    lib.orc_tex_solid_rgb.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    lib.orc_tex_solid_f.argtypes = [C.c_void_p, C.c_double]
    lib.orc_tex_checker.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    lib.orc_tex_image_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.orc_tex_image_rgbf32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.orc_scene_set_float_hdr.argtypes = [C.c_void_p, C.c_int]
    lib.orc_load_hdr_rgbf32.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.orc_mat_diffuse.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_mat_metal.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.orc_mat_glass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.orc_mat_principled.argtypes = [C.c_void_p, C.c_int, d3]
    lib.orc_mat_light.argtypes = [C.c_void_p, C.c_int]
    lib.orc_mat_mix.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    lib.orc_mat_sheen.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.orc_mat_clearcoat.argtypes = [C.c_void_p, C.c_double]
    lib.orc_sphere.argtypes = [C.c_void_p, C.c_double, d3, d3, C.c_int]
    lib.orc_quad.argtypes = [C.c_void_p, d3, d3, d3, C.c_int]
    lib.orc_cuboid.argtypes = [C.c_void_p, d3, d3, C.c_int]
    lib.orc_mesh.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                             C.c_uint32, C.c_void_p, C.c_int]
    lib.orc_instance.argtypes = [C.c_void_p, C.c_int, d3, C.c_double, d3]
    lib.orc_world_add_object.argtypes = [C.c_void_p, C.c_int]
    lib.orc_world_add_light.argtypes = [C.c_void_p, C.c_int]
    lib.orc_world_build.argtypes = [C.c_void_p]
    lib.orc_world_prim_count.argtypes = [C.c_void_p]
    lib.orc_world_prim_count.restype = C.c_uint32
    lib.orc_register_image.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.orc_build_scene.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint64, C.POINTER(Camera)]
    lib.orc_camera_init.argtypes = [C.POINTER(Camera), d3, C.POINTER(C.c_uint32)]
    lib.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    lib.orc_trace_sample.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint64, C.c_uint32, C.c_uint32, d3, C.c_void_p, C.c_uint32]
    lib.orc_resolve_u8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.orc_resolve_u8.restype = None
    lib.orc_philox4x32_10.restype = None
    lib.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.orc_rng_uniform.restype = C.c_double
    lib.orc_mat_probe.argtypes = [C.c_void_p, C.c_int, d3, d3, d3, d3]
    lib.orc_mat_sample_probe.argtypes = [C.c_void_p, C.c_int, d3, d3, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.orc_probe.argtypes = [C.c_int, d3]
    lib.orc_probe.restype = C.c_double
    lib.orc_intersect.argtypes = [C.c_void_p, d3, d3, C.c_double, d3]
    lib.orc_load_obj.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)),
                                 C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32)]
    lib.orc_load_hdr_rgb8.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_free.restype = None
    lib.orc_set_math_mode.argtypes = [C.c_int]
    lib.orc_set_math_mode.restype = None
    lib.orc_detmath.argtypes = [C.c_int, C.c_double, C.c_double]
    lib.orc_detmath.restype = C.c_double
    return lib


lib = _load()


def _check(rc, what="oracle call"):
    if rc < 0:
        raise OracleError(f"{what}: {lib.orc_last_error().decode()}")
    return rc


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def set_math_mode(det: bool):
    """False: platform libm (faithful to the Rust reference). True: the deterministic
    elementary functions the GPU kernels use (bit-exact parity mode)."""
    lib.orc_set_math_mode(1 if det else 0)


def detmath(which: int, a: float, b: float = 0.0) -> float:
    return lib.orc_detmath(which, a, b)


def probe(which: int, *args: float) -> float:
    return lib.orc_probe(which, (C.c_double * max(1, len(args)))(*args))


def philox4x32_10(ctr: Sequence[int], key: Sequence[int]):
    out = (C.c_uint32 * 4)()
    lib.orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def rng_uniform(seed: int, pixel: int, sample: int, draw: int) -> float:
    return lib.orc_rng_uniform(seed, pixel, sample, draw)


class Scene:
    def __init__(self):
        self.handle = C.c_void_p(lib.orc_scene_create())

    def close(self):
        if self.handle:
            lib.orc_scene_destroy(self.handle)
            self.handle = None

    def call(self, name: str, *args):
        return getattr(self, name)(*args)

    def tex_solid_rgb(self, r, g, b): return _check(lib.orc_tex_solid_rgb(self.handle, r, g, b))
    def tex_solid_f(self, v): return _check(lib.orc_tex_solid_f(self.handle, v))
    def tex_checker(self, scale, t1, t2): return _check(lib.orc_tex_checker(self.handle, scale, t1, t2))

    def tex_image_rgb8(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        return _check(lib.orc_tex_image_rgb8(self.handle, w, h, img.ctypes.data))

    def tex_image_rgbf32(self, img):
        img = np.ascontiguousarray(img, dtype=np.float32)
        h, w = img.shape[:2]
        return _check(lib.orc_tex_image_rgbf32(self.handle, w, h, img.ctypes.data))

    def set_float_hdr(self, on=True): return _check(lib.orc_scene_set_float_hdr(self.handle, 1 if on else 0))

    def mat_diffuse(self, color_tex, normal_map_tex=-1): return _check(lib.orc_mat_diffuse(self.handle, color_tex, normal_map_tex))
    def mat_metal(self, color_tex, rough_tex): return _check(lib.orc_mat_metal(self.handle, color_tex, rough_tex))
    def mat_glass(self, color_tex, rough_tex, aniso, ior): return _check(lib.orc_mat_glass(self.handle, color_tex, rough_tex, aniso, ior))
    def mat_principled(self, color_tex, params): return _check(lib.orc_mat_principled(self.handle, color_tex, (C.c_double * 11)(*params)))
    def mat_light(self, tex): return _check(lib.orc_mat_light(self.handle, tex))
    def mat_mix(self, t, m1, m2): return _check(lib.orc_mat_mix(self.handle, t, m1, m2))
    def mat_sheen(self, rgb, sheen_tint): return _check(lib.orc_mat_sheen(self.handle, rgb[0], rgb[1], rgb[2], sheen_tint))
    def mat_clearcoat(self, gloss): return _check(lib.orc_mat_clearcoat(self.handle, gloss))
    def mat_sample_probe(self, mat, n, wo, seed, n_samples, front=True):
        out = np.zeros((n_samples, 4), dtype=np.float64)
        lib.orc_mat_probe_front_face(1 if front else 0)
        _check(lib.orc_mat_sample_probe(self.handle, mat, _d3(n), _d3(wo), seed, n_samples, out.ctypes.data))
        lib.orc_mat_probe_front_face(1)
        return out[:, :3], out[:, 3] > 0

    def mat_probe(self, mat, n, wo, wi, front=True):
        """(pdf, eval) at a synthetic hit with normal n; front = HitInfo::front_face (hit_info.rs:24)."""
        out = (C.c_double * 4)()
        lib.orc_mat_probe_front_face(1 if front else 0)
        _check(lib.orc_mat_probe(self.handle, mat, _d3(n), _d3(wo), _d3(wi), out))
        lib.orc_mat_probe_front_face(1)
        return out[0], np.array(out[1:4])
    def sphere(self, r, p1, p2, mat): return _check(lib.orc_sphere(self.handle, r, _d3(p1), _d3(p2), mat))
    def quad(self, q, u, v, mat): return _check(lib.orc_quad(self.handle, _d3(q), _d3(u), _d3(v), mat))
    def cuboid(self, a, b, mat): return _check(lib.orc_cuboid(self.handle, _d3(a), _d3(b), mat))

    def mesh(self, scale, pos, idx, nrm, uv, mat):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1)
        nrm = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
        uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        return _check(lib.orc_mesh(self.handle, scale, len(pos), pos.ctypes.data, len(idx), idx.ctypes.data,
                                   0 if nrm is None else len(nrm), None if nrm is None else nrm.ctypes.data,
                                   0 if uv is None else len(uv), None if uv is None else uv.ctypes.data, mat))

    def instance(self, obj, axis, angle, translation): return _check(lib.orc_instance(self.handle, obj, _d3(axis), angle, _d3(translation)))
    def world_add_object(self, obj): return _check(lib.orc_world_add_object(self.handle, obj))
    def world_add_light(self, obj): return _check(lib.orc_world_add_light(self.handle, obj))
    def world_build(self): return _check(lib.orc_world_build(self.handle))
    def prim_count(self): return lib.orc_world_prim_count(self.handle)

    def register_image(self, name, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        _check(lib.orc_register_image(self.handle, name.encode(), w, h, img.ctypes.data))

    def build_scene(self, scene_id, width, spp, asset_dir=ASSET_DIR, scene_seed=1, images=None) -> Camera:
        """images: {name: rgb8 array} for the JPEG/PNG files the scene opens."""
        for name, img in (images or {}).items():
            self.register_image(name, img)
        cam = Camera()
        _check(lib.orc_build_scene(self.handle, scene_id, width, spp, asset_dir.encode(), None, 0, 0, scene_seed, C.byref(cam)), "orc_build_scene")
        return cam

    def render(self, cam, seed, spp_begin, spp_end, accum=None, nthreads=0):
        h = image_height(cam)
        if accum is None:
            accum = np.zeros((h, cam.image_width, 3), dtype=np.float64)
        counters = (C.c_uint64 * 4)()
        _check(lib.orc_render(self.handle, C.byref(cam), seed, spp_begin, spp_end, accum.ctypes.data, counters, nthreads), "orc_render")
        return accum, {"segments": counters[0], "box_tests": counters[1], "prim_tests": counters[2], "samples": counters[3]}

    def trace_sample(self, cam, seed, pixel, sample, max_rec=64):
        rad = (C.c_double * 3)()
        dump = np.zeros((max_rec, 8), dtype=np.float64)
        n = _check(lib.orc_trace_sample(self.handle, C.byref(cam), seed, pixel, sample, rad, dump.ctypes.data, max_rec), "orc_trace_sample")
        return np.array(rad), dump[: min(n, max_rec)], n

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        out = np.zeros((len(rays), 15), dtype=np.float64)
        for i, r in enumerate(rays):
            o = (C.c_double * 15)()
            _check(lib.orc_intersect(self.handle, _d3(r[0:3]), _d3(r[3:6]), float(r[6]), o), "orc_intersect")
            out[i] = np.array(o)
        return out


def camera_init(cam):
    out = (C.c_double * 18)()
    h = C.c_uint32()
    _check(lib.orc_camera_init(C.byref(cam), out, C.byref(h)), "orc_camera_init")
    v = np.array(out).reshape(6, 3)
    names = ["forward", "right", "up", "pixel00", "pixel_du", "pixel_dv"]
    return {n: v[i] for i, n in enumerate(names)}, h.value


def image_height(cam):
    return camera_init(cam)[1]


def resolve_u8(accum, total_spp):
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    out = np.empty(accum.shape, dtype=np.uint8)
    lib.orc_resolve_u8(accum.ctypes.data, accum.size // 3, total_spp, out.ctypes.data)
    return out


def load_obj(path):
    pos, idx, uv = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
    npos, nidx, nuv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(lib.orc_load_obj(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(uv), C.byref(nuv)), "orc_load_obj")
    P = np.ctypeslib.as_array(pos, (npos.value * 3,)).copy().reshape(-1, 3) if npos.value else np.zeros((0, 3), np.float32)
    I = np.ctypeslib.as_array(idx, (nidx.value,)).copy() if nidx.value else np.zeros(0, np.uint32)
    T = np.ctypeslib.as_array(uv, (nuv.value * 2,)).copy().reshape(-1, 2) if nuv.value else np.zeros((0, 2), np.float32)
    lib.orc_free(pos); lib.orc_free(idx); lib.orc_free(uv)
    return P, I, T


def load_hdr_rgbf32(path):
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.orc_load_hdr_rgbf32(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "orc_load_hdr_rgbf32")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.orc_free(p)
    return img


def load_hdr_rgb8(path):
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.orc_load_hdr_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "orc_load_hdr_rgb8")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.orc_free(p)
    return img
