"""MI355X-native wavefront path tracer — Python binding of the C ABI (include/pt_amd.h).

This is the drop-in for ONE path of chiefchewie/thu-acg-f2024-path-tracer: the per-pixel
integrator behind ``Camera::render`` (src/camera.rs:79). The binding is ctypes over
``libpt_amd.so`` (hand-written HIP kernels for gfx950 + C++ host runtime); there is no CPU
fallback — importing works without a GPU (so the symbol table can be checked), creating a
:class:`Context` does not.

Because the directory name contains hyphens, import it with
``importlib.import_module("thu-acg-f2024-path-tracer_amd")``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
# PT_AMD_LIB: A/B measurements of two builds of the library in one session (tools/ab_perf.sh); never a different backend
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "libpt_amd.so")
ASSET_DIR = os.path.join(REPO_ROOT, "assets")


class PtError(RuntimeError):
    pass


class Camera(C.Structure):
    """The 12 public fields of the reference's ``Camera`` (src/camera.rs:23-36)."""

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


class RenderOpts(C.Structure):
    _fields_ = [
        ("slots_per_pixel", C.c_uint32),
        ("accum_on_device", C.c_uint32),
        ("profile", C.c_uint32),
        ("overwrite", C.c_uint32),
        ("stream", C.c_void_p),
    ]


class RenderStats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("iterations", C.c_uint64),
        ("n_slots", C.c_uint32),
        ("slots_per_pixel", C.c_uint32),
        ("ms_total", C.c_double),
        ("ms_extend", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_other", C.c_double),
        ("launches_extend", C.c_uint64),
        ("launches_shade", C.c_uint64),
        ("extend_variant", C.c_uint32),
        ("shade_variant", C.c_uint32),
        ("blocks_extend", C.c_uint32),
        ("blocks_shade", C.c_uint32),
        ("compactions", C.c_uint32),
        ("n_alloc_end", C.c_uint32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/pt_amd.h declares (the not-gpu test checks the library exports them all)
ABI_SYMBOLS = [
    "pt_last_error", "pt_set_error_message", "pt_ctx_create", "pt_ctx_destroy", "pt_device_name",
    "pt_scene_create", "pt_scene_destroy", "pt_scene_ctx",
    "pt_tex_solid_rgb", "pt_tex_solid_f", "pt_tex_checker", "pt_tex_image_rgb8", "pt_tex_image_rgbf32", "pt_scene_set_float_hdr", "pt_scene_float_hdr",
    "pt_mat_diffuse", "pt_mat_metal", "pt_mat_glass", "pt_mat_principled", "pt_mat_light", "pt_mat_mix", "pt_mat_sheen", "pt_mat_clearcoat",
    "pt_sphere", "pt_quad", "pt_cuboid", "pt_mesh", "pt_instance",
    "pt_world_add_object", "pt_world_add_light", "pt_world_build", "pt_world_prim_count",
    "pt_world_set_device_bvh_threshold", "pt_world_device_bvh_info",
    "pt_load_obj", "pt_load_obj_single_index", "pt_load_hdr_rgb8", "pt_load_hdr_rgbf32", "pt_load_png_rgb8", "pt_load_jpeg_rgb8", "pt_free", "pt_register_image", "pt_find_registered_image", "pt_save_png",
    "pt_build_scene", "pt_camera_init", "pt_render", "pt_resolve_u8", "pt_intersect", "pt_math_probe",
    "pt_shard_range", "pt_comm_create", "pt_comm_destroy", "pt_comm_rank", "pt_comm_world", "pt_comm_barrier", "pt_comm_allreduce_f64",
    "pt_bootstrap_exchange", "pt_render_multi",
]


def _load():
    if not os.path.exists(LIB_PATH):
        raise PtError(
            f"{LIB_PATH} is missing: build it with `make -C {_HERE}` (or __graft_entry__.build()). "
            "There is no Python/CPU fallback for the render path."
        )
    lib = C.CDLL(LIB_PATH)
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_scene_create.restype = C.c_void_p
    lib.pt_scene_create.argtypes = [C.c_void_p]
    lib.pt_scene_destroy.argtypes = [C.c_void_p]
    lib.pt_scene_destroy.restype = None
    lib.pt_scene_ctx.restype = C.c_void_p
    lib.pt_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.pt_ctx_destroy.argtypes = [C.c_void_p]
    lib.pt_ctx_destroy.restype = None
    lib.pt_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    d3 = C.POINTER(C.c_double)
    lib.pt_tex_solid_rgb.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
    lib.pt_tex_solid_f.argtypes = [C.c_void_p, C.c_double]
    lib.pt_tex_checker.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    lib.pt_tex_image_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pt_tex_image_rgbf32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pt_scene_set_float_hdr.argtypes = [C.c_void_p, C.c_int]
    lib.pt_scene_float_hdr.argtypes = [C.c_void_p]
    lib.pt_load_hdr_rgbf32.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pt_mat_diffuse.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pt_mat_metal.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pt_mat_glass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.pt_mat_principled.argtypes = [C.c_void_p, C.c_int, d3]
    lib.pt_mat_light.argtypes = [C.c_void_p, C.c_int]
    lib.pt_mat_mix.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]
    lib.pt_mat_sheen.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
    lib.pt_mat_clearcoat.argtypes = [C.c_void_p, C.c_double]
    lib.pt_sphere.argtypes = [C.c_void_p, C.c_double, d3, d3, C.c_int]
    lib.pt_quad.argtypes = [C.c_void_p, d3, d3, d3, C.c_int]
    lib.pt_cuboid.argtypes = [C.c_void_p, d3, d3, C.c_int]
    lib.pt_mesh.argtypes = [C.c_void_p, C.c_double, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                            C.c_uint32, C.c_void_p, C.c_int]
    lib.pt_instance.argtypes = [C.c_void_p, C.c_int, d3, C.c_double, d3]
    lib.pt_world_add_object.argtypes = [C.c_void_p, C.c_int]
    lib.pt_world_add_light.argtypes = [C.c_void_p, C.c_int]
    lib.pt_world_build.argtypes = [C.c_void_p]
    lib.pt_world_prim_count.argtypes = [C.c_void_p]
    lib.pt_world_prim_count.restype = C.c_uint32
    if hasattr(lib, "pt_world_set_device_bvh_threshold"):
        lib.pt_world_set_device_bvh_threshold.argtypes = [C.c_void_p, C.c_uint32]
        lib.pt_world_device_bvh_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pt_register_image.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pt_find_registered_image.argtypes = [C.c_void_p, C.c_char_p]
    lib.pt_save_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pt_build_scene.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.POINTER(Camera)]
    lib.pt_camera_init.argtypes = [C.POINTER(Camera), d3, C.POINTER(C.c_uint32)]
    lib.pt_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                              C.POINTER(RenderOpts), C.POINTER(RenderStats)]
    lib.pt_resolve_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.pt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.pt_math_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.pt_load_obj.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)),
                                C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32)]
    lib.pt_load_hdr_rgb8.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    if hasattr(lib, "pt_load_jpeg_rgb8"):
        lib.pt_load_jpeg_rgb8.argtypes = lib.pt_load_hdr_rgb8.argtypes
    if hasattr(lib, "pt_load_png_rgb8"):
        lib.pt_load_png_rgb8.argtypes = lib.pt_load_hdr_rgb8.argtypes
        fp, up = C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32)
        lib.pt_load_obj_single_index.argtypes = [C.c_char_p, fp, up, C.POINTER(C.POINTER(C.c_uint32)), up, fp, up, fp, up]
    lib.pt_free.argtypes = [C.c_void_p]
    lib.pt_free.restype = None
    if os.environ.get("PT_AMD_LIB") and not hasattr(lib, "pt_shard_range"):
        return lib                                   # A/B run against a build that predates the multi-GPU entry points
    lib.pt_shard_range.argtypes = [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pt_shard_range.restype = None
    lib.pt_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_double, C.POINTER(C.c_void_p)]
    lib.pt_comm_destroy.argtypes = [C.c_void_p]
    lib.pt_comm_destroy.restype = None
    lib.pt_comm_rank.argtypes = [C.c_void_p]
    lib.pt_comm_world.argtypes = [C.c_void_p]
    lib.pt_comm_barrier.argtypes = [C.c_void_p]
    lib.pt_comm_allreduce_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    lib.pt_bootstrap_exchange.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_uint32, C.c_double]
    lib.pt_render_multi.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                    C.POINTER(RenderOpts), C.POINTER(RenderStats)]
    return lib


lib = _load()


def _check(rc, what="pt call"):
    if rc < 0:
        raise PtError(f"{what}: {lib.pt_last_error().decode()}")
    return rc


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def decode_image_rgb8(path: str) -> np.ndarray:
    """JPEG/PNG -> RGB8 (H, W, 3) via PILLOW — an independent decoder, kept for the tests (the oracle is fed these pixels, the
    product decodes the same files itself: pt_load_jpeg_rgb8 / pt_load_png_rgb8) and for formats the library has no decoder
    for (hand the result to Scene.register_image). Alpha is dropped like ``to_rgb8`` does, texture.rs:67."""
    from PIL import Image

    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8))


class Context:
    """One GPU. Raises PtError when no HIP device is present (no CPU fallback)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib.pt_ctx_create(device, C.byref(h)), "pt_ctx_create")
        self.handle = h

    def name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(lib.pt_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def close(self):
        if self.handle:
            lib.pt_ctx_destroy(self.handle)
            self.handle = None

    def math_probe(self, which: int, ab: np.ndarray) -> np.ndarray:
        ab = np.ascontiguousarray(ab, dtype=np.float64).reshape(-1, 2)
        out = np.empty(len(ab), dtype=np.float64)
        _check(lib.pt_math_probe(self.handle, which, ab.ctypes.data, len(ab), out.ctypes.data), "pt_math_probe")
        return out

    def resolve_u8(self, accum: np.ndarray, total_spp: int) -> np.ndarray:
        accum = np.ascontiguousarray(accum, dtype=np.float64)
        out = np.empty(accum.shape, dtype=np.uint8)
        _check(lib.pt_resolve_u8(self.handle, accum.ctypes.data, accum.size // 3, total_spp, out.ctypes.data), "pt_resolve_u8")
        return out


def shard_range(spp: int, rank: int, world: int):
    """Sample range [lo, hi) that rank `rank` of `world` renders (pt_shard_range)."""
    lo, hi = C.c_uint32(), C.c_uint32()
    lib.pt_shard_range(spp, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def bootstrap_exchange(path: str, rank: int, payload: bytes, timeout_s: float = 60.0) -> bytes:
    """The file rendezvous pt_comm_create uses for the RCCL unique id: rank 0 publishes `payload`, the others get it."""
    buf = C.create_string_buffer(payload, len(payload))
    _check(lib.pt_bootstrap_exchange(path.encode(), rank, buf, len(payload), timeout_s), "pt_bootstrap_exchange")
    return buf.raw


class Comm:
    """One rank of an RCCL communicator over the GPUs of a node (one process per GPU)."""

    def __init__(self, ctx: Context, rank: int, world: int, id_path: Optional[str] = None, timeout_s: float = 120.0):
        h = C.c_void_p()
        _check(lib.pt_comm_create(ctx.handle, rank, world, None if id_path is None else id_path.encode(), timeout_s, C.byref(h)), "pt_comm_create")
        self.handle, self.ctx = h, ctx
        self.rank, self.world = int(lib.pt_comm_rank(h)), int(lib.pt_comm_world(h))     # what the communicator says, not what was asked for

    def close(self):
        if self.handle:
            lib.pt_comm_destroy(self.handle)
            self.handle = None

    def barrier(self):
        _check(lib.pt_comm_barrier(self.handle), "pt_comm_barrier")

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        _check(lib.pt_comm_allreduce_f64(self.handle, v.ctypes.data, v.size, 1 if op == "max" else 0), "pt_comm_allreduce_f64")
        return v


class Scene:
    """World + builder (hittable/world.rs). Method names follow the C ABI minus the prefix."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.handle = C.c_void_p(lib.pt_scene_create(ctx.handle))
        if not self.handle:
            raise PtError("pt_scene_create failed")

    def close(self):
        if self.handle:
            lib.pt_scene_destroy(self.handle)
            self.handle = None

    # generic dispatcher used by tests that replay one scene description onto this API and
    # onto the oracle's isomorphic one
    def call(self, name: str, *args):
        return getattr(self, name)(*args)

    def tex_solid_rgb(self, r, g, b): return _check(lib.pt_tex_solid_rgb(self.handle, r, g, b), "tex_solid_rgb")
    def tex_solid_f(self, v): return _check(lib.pt_tex_solid_f(self.handle, v), "tex_solid_f")
    def tex_checker(self, scale, t1, t2): return _check(lib.pt_tex_checker(self.handle, scale, t1, t2), "tex_checker")

    def tex_image_rgb8(self, img: np.ndarray):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        return _check(lib.pt_tex_image_rgb8(self.handle, w, h, img.ctypes.data), "tex_image_rgb8")

    def tex_image_rgbf32(self, img: np.ndarray):
        """ImageTexture that keeps f32 samples (no to_rgb8 squash, texture.rs:67): the float-HDR option."""
        img = np.ascontiguousarray(img, dtype=np.float32)
        h, w = img.shape[:2]
        return _check(lib.pt_tex_image_rgbf32(self.handle, w, h, img.ctypes.data), "tex_image_rgbf32")

    def set_float_hdr(self, on: bool = True):
        """Scene scripts (build_scene) load Radiance .hdr files as f32 textures from now on."""
        return _check(lib.pt_scene_set_float_hdr(self.handle, 1 if on else 0), "set_float_hdr")

    def mat_diffuse(self, color_tex, normal_map_tex=-1): return _check(lib.pt_mat_diffuse(self.handle, color_tex, normal_map_tex), "mat_diffuse")
    def mat_metal(self, color_tex, rough_tex): return _check(lib.pt_mat_metal(self.handle, color_tex, rough_tex), "mat_metal")
    def mat_glass(self, color_tex, rough_tex, aniso, ior): return _check(lib.pt_mat_glass(self.handle, color_tex, rough_tex, aniso, ior), "mat_glass")

    def mat_principled(self, color_tex, params: Sequence[float]):
        assert len(params) == 11
        return _check(lib.pt_mat_principled(self.handle, color_tex, (C.c_double * 11)(*params)), "mat_principled")

    def mat_light(self, tex): return _check(lib.pt_mat_light(self.handle, tex), "mat_light")
    def mat_mix(self, t, m1, m2): return _check(lib.pt_mat_mix(self.handle, t, m1, m2), "mat_mix")
    def mat_sheen(self, rgb, sheen_tint): return _check(lib.pt_mat_sheen(self.handle, rgb[0], rgb[1], rgb[2], sheen_tint), "mat_sheen")
    def mat_clearcoat(self, gloss): return _check(lib.pt_mat_clearcoat(self.handle, gloss), "mat_clearcoat")
    def sphere(self, r, p1, p2, mat): return _check(lib.pt_sphere(self.handle, r, _d3(p1), _d3(p2), mat), "sphere")
    def quad(self, q, u, v, mat): return _check(lib.pt_quad(self.handle, _d3(q), _d3(u), _d3(v), mat), "quad")
    def cuboid(self, a, b, mat): return _check(lib.pt_cuboid(self.handle, _d3(a), _d3(b), mat), "cuboid")

    def mesh(self, scale, pos, idx, nrm, uv, mat):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1)
        nrm = None if nrm is None else np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
        uv = None if uv is None else np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        return _check(lib.pt_mesh(self.handle, scale, len(pos), pos.ctypes.data, len(idx), idx.ctypes.data,
                                  0 if nrm is None else len(nrm), None if nrm is None else nrm.ctypes.data,
                                  0 if uv is None else len(uv), None if uv is None else uv.ctypes.data, mat), "mesh")

    def instance(self, obj, axis, angle, translation): return _check(lib.pt_instance(self.handle, obj, _d3(axis), angle, _d3(translation)), "instance")
    def world_add_object(self, obj): return _check(lib.pt_world_add_object(self.handle, obj), "world_add_object")
    def world_add_light(self, obj): return _check(lib.pt_world_add_light(self.handle, obj), "world_add_light")
    def world_build(self): return _check(lib.pt_world_build(self.handle), "world_build")
    def prim_count(self) -> int: return lib.pt_world_prim_count(self.handle)

    def set_device_bvh_threshold(self, min_triangles: int):
        """Meshes with at least this many triangles get their BVH built on the GPU at the next world_build (0 = never)."""
        _check(lib.pt_world_set_device_bvh_threshold(self.handle, min_triangles), "pt_world_set_device_bvh_threshold")

    def device_bvh_info(self):
        n, d = C.c_uint32(), C.c_uint32()
        _check(lib.pt_world_device_bvh_info(self.handle, C.byref(n), C.byref(d)), "pt_world_device_bvh_info")
        return n.value, d.value

    def register_image(self, name: str, img: np.ndarray):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        _check(lib.pt_register_image(self.handle, name.encode(), w, h, img.ctypes.data), "register_image")

    def build_scene(self, scene_id: int, width: int, spp: int, asset_dir: str = ASSET_DIR, scene_seed: int = 1) -> Camera:
        """Run the reference's scene script N (main.rs `-s N`). Every image the scripts open (.hdr, .png, .jpg) is decoded by the
        library itself (pt_load_hdr_rgb8 / pt_load_png_rgb8 / pt_load_jpeg_rgb8); pixels handed over with register_image first
        take precedence."""
        cam = Camera()
        _check(lib.pt_build_scene(self.handle, scene_id, width, spp, asset_dir.encode(), scene_seed, C.byref(cam)), "pt_build_scene")
        return cam

    def render(self, cam: Camera, seed: int, spp_begin: int, spp_end: int, accum=None, slots_per_pixel: int = 0,
               profile: bool = False, device_ptr: Optional[int] = None, stream: Optional[int] = None, overwrite: bool = False):
        """Camera::render (camera.rs:79) without gamma/quantise: returns (accum, stats) where
        accum[(y, x, c)] += sum over samples [spp_begin, spp_end) of trace(y, x) (``overwrite``: = instead of +=).
        ``device_ptr``: write into device memory instead (e.g. ``tensor.data_ptr()``)."""
        h = image_height(cam)
        opts = RenderOpts(slots_per_pixel, 1 if device_ptr is not None else 0, 1 if profile else 0, 1 if overwrite else 0, stream)
        stats = RenderStats()
        if device_ptr is not None:
            ptr = C.c_void_p(device_ptr)
        else:
            if accum is None:
                accum = np.zeros((h, cam.image_width, 3), dtype=np.float64)
            assert accum.dtype == np.float64 and accum.flags["C_CONTIGUOUS"] and accum.size == h * cam.image_width * 3
            ptr = C.c_void_p(accum.ctypes.data)
        _check(lib.pt_render(self.handle, C.byref(cam), seed, spp_begin, spp_end, ptr, C.byref(opts), C.byref(stats)), "pt_render")
        return accum, stats

    def render_multi(self, cam: Camera, seed: int, spp_total: int, comm: Comm, accum=None, slots_per_pixel: int = 0, profile: bool = False,
                     overwrite: bool = False):
        """Camera::render over all ranks of `comm` (pt_render_multi): spp sharding + one RCCL reduce onto rank 0.
        Returns (accum on rank 0 / None elsewhere, this rank's stats). ``overwrite``: the frame replaces ``accum``'s content
        instead of being added to it (a host that renders frame after frame into one buffer)."""
        h = image_height(cam)
        opts = RenderOpts(slots_per_pixel, 0, 1 if profile else 0, 1 if overwrite else 0, None)
        stats = RenderStats()
        ptr = None
        if comm.rank == 0:
            if accum is None:
                accum = np.zeros((h, cam.image_width, 3), dtype=np.float64)
            assert accum.dtype == np.float64 and accum.flags["C_CONTIGUOUS"] and accum.size == h * cam.image_width * 3
            ptr = C.c_void_p(accum.ctypes.data)
        _check(lib.pt_render_multi(self.handle, C.byref(cam), seed, spp_total, comm.handle, ptr, C.byref(opts), C.byref(stats)), "pt_render_multi")
        return (accum if comm.rank == 0 else None), stats

    def intersect(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 7)
        out = np.empty((len(rays), 15), dtype=np.float64)
        _check(lib.pt_intersect(self.handle, rays.ctypes.data, len(rays), out.ctypes.data), "pt_intersect")
        return out


# every non-HDR image a scene script opens (for hosts that decode everything themselves, e.g. the test oracle via Pillow)
SCENE_IMAGE_FILES = {2: ["earthmap.jpg"], 5: ["envmap.jpg"], 7: ["bricks/color.png", "bricks/normal.png"]}


def camera_init(cam: Camera):
    """Camera::init (camera.rs:51-77) -> (dict of derived vectors, image_height)."""
    out = (C.c_double * 18)()
    h = C.c_uint32()
    _check(lib.pt_camera_init(C.byref(cam), out, C.byref(h)), "pt_camera_init")
    v = np.array(out).reshape(6, 3)
    names = ["forward", "right", "up", "pixel00", "pixel_du", "pixel_dv"]
    return {n: v[i] for i, n in enumerate(names)}, h.value


def image_height(cam: Camera) -> int:
    return camera_init(cam)[1]


def load_obj(path: str):
    pos, idx, uv = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)()
    npos, nidx, nuv = C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_obj(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(uv), C.byref(nuv)), "pt_load_obj")
    P = np.ctypeslib.as_array(pos, (npos.value * 3,)).copy().reshape(-1, 3) if npos.value else np.zeros((0, 3), np.float32)
    I = np.ctypeslib.as_array(idx, (nidx.value,)).copy() if nidx.value else np.zeros(0, np.uint32)
    T = np.ctypeslib.as_array(uv, (nuv.value * 2,)).copy().reshape(-1, 2) if nuv.value else np.zeros((0, 2), np.float32)
    lib.pt_free(pos); lib.pt_free(idx); lib.pt_free(uv)
    return P, I, T


def load_obj_single_index(path: str):
    """OBJ with vn / separate index streams -> (positions, indices, normals | None, texcoords | None), one index per corner."""
    pos, idx, nrm, uv = C.POINTER(C.c_float)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    npos, nidx, nnrm, nuv = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_obj_single_index(path.encode(), C.byref(pos), C.byref(npos), C.byref(idx), C.byref(nidx), C.byref(nrm), C.byref(nnrm),
                                        C.byref(uv), C.byref(nuv)), "pt_load_obj_single_index")
    arr = lambda p, n, k, dt: np.ctypeslib.as_array(p, (n * k,)).copy().reshape(-1, k) if n else None
    out = (arr(pos, npos.value, 3, np.float32), np.ctypeslib.as_array(idx, (nidx.value,)).copy() if nidx.value else np.zeros(0, np.uint32),
           arr(nrm, nnrm.value, 3, np.float32), arr(uv, nuv.value, 2, np.float32))
    for p in (pos, idx, nrm, uv):
        lib.pt_free(p)
    return out


def load_jpeg_rgb8(path: str) -> np.ndarray:
    """Baseline / progressive JPEG -> RGB8 by the library's own decoder (csrc/pt_jpeg.cpp)."""
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_jpeg_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "pt_load_jpeg_rgb8")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.pt_free(p)
    return img


def load_png_rgb8(path: str) -> np.ndarray:
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_png_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "pt_load_png_rgb8")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.pt_free(p)
    return img


def load_hdr_rgbf32(path: str) -> np.ndarray:
    """Radiance .hdr -> f32 RGB (the decode of texture.rs:62-66 WITHOUT .to_rgb8())."""
    p = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_hdr_rgbf32(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "pt_load_hdr_rgbf32")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.pt_free(p)
    return img


def load_hdr_rgb8(path: str) -> np.ndarray:
    p = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(), C.c_uint32()
    _check(lib.pt_load_hdr_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)), "pt_load_hdr_rgb8")
    img = np.ctypeslib.as_array(p, (h.value, w.value, 3)).copy()
    lib.pt_free(p)
    return img


def save_png(path: str, rgb8: np.ndarray):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w = rgb8.shape[:2]
    _check(lib.pt_save_png(path.encode(), w, h, rgb8.ctypes.data), "pt_save_png")
