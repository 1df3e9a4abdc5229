"""Shared helpers for the tests: a backend-neutral scene description that is replayed onto
the product API (GPU) and onto the oracle API (CPU), plus a small ulp metric."""
from __future__ import annotations

import numpy as np

GOLDEN_DIR = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.int64(-(2**63)) - ia[ia < 0]
    ib[ib < 0] = np.int64(-(2**63)) - ib[ib < 0]
    with np.errstate(over="ignore"):
        d = np.abs(ia - ib).astype(np.float64)      # exact in int64 (values of like sign / near zero)
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


class SceneSpec:
    """Records builder calls; handles are indices into the call list's results."""

    def __init__(self):
        self.calls = []   # (method, args) where an arg wrapped in Ref(i) is the result of call i
        self.camera = None

    class Ref(int):
        pass

    def add(self, method, *args):
        self.calls.append((method, args))
        return SceneSpec.Ref(len(self.calls) - 1)

    def replay(self, scene):
        results = []
        for method, args in self.calls:
            real = [results[a] if isinstance(a, SceneSpec.Ref) else a for a in args]
            results.append(scene.call(method, *real))
        return results

    def make_camera(self, cam_cls, results):
        c = cam_cls()
        for k, v in self.camera.items():
            if k == "env_tex":
                v = results[v] if isinstance(v, SceneSpec.Ref) else v
            if isinstance(v, (tuple, list)):
                for i, x in enumerate(v):
                    getattr(c, k)[i] = x
            else:
                setattr(c, k, v)
        return c


def default_camera(width=48, aspect=1.0, spp=4, **kw):
    cam = dict(aspect_ratio=aspect, image_width=width, samples_per_pixel=spp, max_depth=50, env_is_map=0, vfov=50.0,
               look_from=(0.0, 1.0, -6.0), look_at=(0.0, 0.5, 0.0), vup=(0.0, 1.0, 0.0), blur_strength=0.5, focal_length=6.0,
               defocus_angle=0.5, env_color=(0.6, 0.7, 0.9), env_tex=-1)
    cam.update(kw)
    return cam


def icosphere(subdiv=1):
    """Small closed triangle mesh (float32 positions, uint32 indices) for mesh tests."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float32), np.array(f, dtype=np.uint32).reshape(-1)


def random_scene(seed, with_mesh=True, with_lights=True, n_objects=10, sphere_light=False):
    """A random but valid scene touching every primitive and material kind."""
    rng = np.random.default_rng(seed)
    s = SceneSpec()
    U = lambda lo, hi: float(rng.uniform(lo, hi))
    col = lambda: (U(0.05, 0.95), U(0.05, 0.95), U(0.05, 0.95))

    def material(allow_mix=True):
        k = int(rng.integers(0, 7 if allow_mix else 6))
        if k == 4:
            return s.add("mat_sheen", col(), U(0, 1))
        if k == 5:
            return s.add("mat_clearcoat", U(0, 1))
        if k == 6:
            return s.add("mat_mix", U(-0.2, 1.2), material(False), material(False))
        tex = s.add("tex_solid_rgb", *col())
        if rng.random() < 0.25:
            tex = s.add("tex_checker", U(0.2, 1.5), tex, s.add("tex_solid_rgb", *col()))
        if k == 0:
            return s.add("mat_diffuse", tex, -1)
        if k == 1:
            return s.add("mat_metal", tex, s.add("tex_solid_f", U(0.0, 0.6)))
        if k == 2:
            return s.add("mat_glass", tex, s.add("tex_solid_f", U(0.001, 0.4)), 0.0, U(1.2, 1.8))
        p = [U(0, 1), U(0.01, 0.8), U(0, 1), U(0, 1), U(0, 1), U(1.2, 1.8), U(0, 1), U(0, 1), U(0, 1), U(0, 1), U(0, 1)]
        return s.add("mat_principled", tex, p)

    ground = s.add("mat_diffuse", s.add("tex_checker", 0.7, s.add("tex_solid_rgb", 0.2, 0.3, 0.1), s.add("tex_solid_rgb", 0.9, 0.9, 0.9)), -1)
    s.add("world_add_object", s.add("quad", (-20.0, 0.0, -20.0), (0.0, 0.0, 40.0), (40.0, 0.0, 0.0), ground))
    for _ in range(n_objects):
        kind = int(rng.integers(0, 5 if with_mesh else 4))
        pos = (U(-3, 3), U(0.3, 2.0), U(-2, 4))
        m = material()
        if kind == 0:
            obj = s.add("sphere", U(0.2, 0.8), pos, pos, m)
        elif kind == 1:
            p2 = (pos[0], pos[1] + U(0, 0.5), pos[2])
            obj = s.add("sphere", U(0.2, 0.5), pos, p2, m)
        elif kind == 2:
            obj = s.add("quad", pos, (U(0.3, 1.5), U(-0.3, 0.3), 0.0), (0.0, U(0.3, 1.5), U(-0.3, 0.3)), m)
        elif kind == 3:
            box = s.add("cuboid", (0.0, 0.0, 0.0), (U(0.3, 1.0), U(0.3, 1.5), U(0.3, 1.0)), m)
            obj = s.add("instance", box, (0.0, 1.0, 0.0), U(-1.0, 1.0), (pos[0], 0.0, pos[2]))
        else:
            P, I = icosphere(int(rng.integers(0, 3)))
            mesh = s.add("mesh", U(0.3, 0.8), P, I, None, None, m)
            axis = np.array([U(-1, 1), U(0.2, 1), U(-1, 1)]); axis /= np.linalg.norm(axis)
            obj = s.add("instance", mesh, tuple(axis), U(-2, 2), pos)
        s.add("world_add_object", obj)
    if with_lights:
        lm = s.add("mat_light", s.add("tex_solid_rgb", 8.0, 8.0, 7.0))
        s.add("world_add_light", s.add("quad", (-1.0, 4.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), lm))
        if sphere_light:   # NB: the reference's Sphere::pdf (sphere.rs:124-135) yields NaN/inf for origins ON the sphere
            lm2 = s.add("mat_light", s.add("tex_solid_rgb", 5.0, 4.0, 3.0))
            c = (U(-2, 2), 3.0, U(-1, 2))
            s.add("world_add_light", s.add("sphere", 0.3, c, c, lm2))
    s.add("world_build")
    s.camera = default_camera(env_color=(0.6, 0.7, 0.9) if not with_lights else (0.05, 0.05, 0.08))
    return s


# ---- the reference's own rendered outputs (demo/*.png) as coarse fixtures ---------------------------
# tests/golden/reference_demo_blocks.npz (tools/make_demo_fixture.py): 48x27 block means of the gamma-space
# 1920x1080 images the reference repository ships for its scene scripts 1, 2, 4, 5, 6 (unknown seed; the
# reference's RNG cannot be seeded). Scene 6: the blocks covering the "spot" mesh are left out — in that
# image spot is smooth-shaded, bright and translucent-looking, which the code of the mounted commit cannot
# produce from assets/spot.obj (no normals: the cow next to it, same loader, is faceted in the same image),
# so that object was rendered with an older material/loader; everything else in the frame agrees.
DEMO_TOL = {2: (0.02, 0.995), 4: (0.01, 0.999), 5: (0.025, 0.99), 6: (0.025, 0.985)}   # scene -> (max mean-abs-diff, min correlation)


def demo_block_stats(scene_id, rgb8):
    """(mean absolute difference per channel, correlation per channel) between the 48x27 block means of an
    RGB8 render of `scene_id` (any size that is a multiple of 48x27) and the reference's demo image."""
    import os
    ref = np.load(os.path.join(GOLDEN_DIR, "reference_demo_blocks.npz"))[f"scene{scene_id}"].astype(np.float64)
    img = np.asarray(rgb8, dtype=np.float64) / 255.0
    h, w, _ = img.shape
    assert h % 27 == 0 and w % 48 == 0, (h, w)
    blocks = img.reshape(27, h // 27, 48, w // 48, 3).mean(axis=(1, 3))
    keep = np.ones((27, 48), dtype=bool)
    if scene_id == 6:
        keep[1:13, 27:36] = False
    d = np.abs(blocks - ref)[keep]
    corr = np.array([np.corrcoef(blocks[..., c][keep], ref[..., c][keep])[0, 1] for c in range(3)])
    return d.mean(axis=0), corr, blocks.mean(axis=(0, 1)), ref.mean(axis=(0, 1))


# ---- the lights / MIS branch against deterministic quadrature (tests/refs_numpy.py) ---------------------------------
# A Lambert floor under ONE emitter, camera straight above the floor and below the emitter, black environment,
# max_depth = 2: the pixel value is E[brdf / (0.5 bsdf_pdf + 0.5 light_pdf) * Le] of trace()'s first bounce
# (camera.rs:199-216, list.rs:78-96, quad.rs:80-98 / sphere.rs:110-135) and nothing else — quirk Q5 needs a third segment.
MIS_ALBEDO, MIS_EMISSION = (0.8, 0.6, 0.4), (6.0, 5.0, 4.0)
MIS_QUAD = ((-0.5, 2.0, -0.5), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
MIS_SPHERE = ((0.0, 2.0, 0.0), 0.4)
MIS_TRI = ((0.9, 2.0, -0.4), (1.9, 2.0, 0.1), (1.0, 2.0, 0.8))          # the one-triangle mesh light of the two-light scene
MIS_TRI_EMISSION = (3.0, 6.0, 9.0)
MIS_QUAD2 = ((-1.6, 2.0, -0.5), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0))           # the quad, moved aside so that the two never overlap in direction
MIS_BOX = ((-0.5, 2.0, -0.4), (0.5, 2.5, 0.4))                                 # a cuboid light (cuboid.rs:78-84)
MIS_INST = ((1.0, 0.0, 0.0), 0.6, (0.3, 0.4, -1.0))                             # Instance::new(quad MIS_QUAD, axis, angle, translation): a TILTED quad light
MIS_CAM = dict(width=24, aspect=1.0, vfov=50.0, look_from=(0.0, 1.0, 0.0), look_at=(0.0, 0.0, 0.0), vup=(0.0, 0.0, 1.0), focal_length=1.0)


def mis_scene(light):
    s = SceneSpec()
    floor = s.add("mat_diffuse", s.add("tex_solid_rgb", *MIS_ALBEDO), -1)
    s.add("world_add_object", s.add("quad", (-4.0, 0.0, -4.0), (0.0, 0.0, 8.0), (8.0, 0.0, 0.0), floor))
    lm = s.add("mat_light", s.add("tex_solid_rgb", *MIS_EMISSION))
    if light == "quad":
        s.add("world_add_light", s.add("quad", *MIS_QUAD, lm))
    elif light == "cuboid":
        s.add("world_add_light", s.add("cuboid", *MIS_BOX, lm))
    elif light == "instquad":
        s.add("world_add_light", s.add("instance", s.add("quad", *MIS_QUAD, lm), *MIS_INST))
    elif light == "two":
        s.add("world_add_light", s.add("quad", *MIS_QUAD2, lm))
        lt = s.add("mat_light", s.add("tex_solid_rgb", *MIS_TRI_EMISSION))
        s.add("world_add_light", s.add("mesh", 1.0, np.array(MIS_TRI, dtype=np.float32), np.array([0, 1, 2], dtype=np.uint32), None, None, lt))
    else:
        s.add("world_add_light", s.add("sphere", MIS_SPHERE[1], MIS_SPHERE[0], MIS_SPHERE[0], lm))
    s.add("world_build")
    c = MIS_CAM
    s.camera = default_camera(width=c["width"], aspect=c["aspect"], spp=1, max_depth=2, vfov=c["vfov"], look_from=c["look_from"], look_at=c["look_at"],
                              vup=c["vup"], focal_length=c["focal_length"], defocus_angle=0.0, blur_strength=0.5, env_color=(0.0, 0.0, 0.0))
    return s


def mis_expected(light):
    """Per-pixel expectation (H, W, 3) of the reference's estimator by quadrature, and the true integral (same thing for
    the quad light; for the sphere light the reference's estimator is biased and both are returned)."""
    import refs_numpy as R
    c = MIS_CAM
    fr = R.camera_frame(c["width"], c["aspect"], c["vfov"], c["look_from"], c["look_at"], c["vup"], c["focal_length"])
    H, W = fr["height"], c["width"]
    rows, cols = np.divmod(np.arange(H * W), W)
    off = R.pixel_footprint()
    pts = R.floor_points(fr, rows, cols, off).reshape(-1, 3)
    pts = pts + np.array([0.0, 1e-3, 0.0])          # the second segment starts EPS above the surface (camera.rs:217-222)
    if light == "quad":
        est = true = R.quad_light_floor_radiance(pts, MIS_ALBEDO, MIS_EMISSION, *MIS_QUAD)
    elif light == "cuboid":
        est = true = R.box_light_floor_radiance(pts, MIS_ALBEDO, MIS_EMISSION, *MIS_BOX)
    elif light == "instquad":
        est = true = R.quad_light_floor_radiance(pts, MIS_ALBEDO, MIS_EMISSION, *R.rigid_quad(*MIS_QUAD, *MIS_INST))
    elif light == "two":
        tri = [np.array(p, dtype=np.float32).astype(np.float64) for p in MIS_TRI]      # the mesh stores f32 positions
        q, u, v = (np.asarray(a, float) for a in MIS_QUAD2)
        lights = [dict(kind="quad", verts=[q, q + u, q + u + v, q + v], emission=MIS_EMISSION), dict(kind="tri", verts=tri, emission=MIS_TRI_EMISSION)]
        est, true = R.coplanar_lights_floor_radiance(pts - np.array([0.0, 1e-3, 0.0]), MIS_ALBEDO, lights)   # it offsets the segment itself
    else:
        est, true = R.sphere_light_floor_radiance(pts, MIS_ALBEDO, MIS_EMISSION, *MIS_SPHERE)
    shape = (H, W, len(off), 3)
    return est.reshape(shape).mean(axis=2), true.reshape(shape).mean(axis=2)


def mis_zscores(render, expected, n_batches=16, spp_per_batch=256, seed=3):
    """render(seed, spp_begin, spp_end) -> sum accumulator (H, W, 3). Returns (z per pixel and channel, z of the image mean, mean image)."""
    batches = np.stack([render(seed, k * spp_per_batch, (k + 1) * spp_per_batch) / spp_per_batch for k in range(n_batches)])
    mean = batches.mean(axis=0)
    sem = batches.std(axis=0, ddof=1) / np.sqrt(n_batches)
    z = (mean - expected) / sem
    g = batches.mean(axis=(1, 2))                                      # (n_batches, 3) image means
    zg = (g.mean(axis=0) - expected.mean(axis=(0, 1))) / (g.std(axis=0, ddof=1) / np.sqrt(n_batches))
    return z, zg, mean


def shared_and_nested_instances_scene(shared=True):
    """One mesh placed three times through instances (shared geometry when `shared`, three equal mesh objects otherwise),
    an instance of an instance of that mesh, a twice-wrapped cuboid and a twice-wrapped quad LIGHT (instance.rs:20-75:
    Instance::new takes any Arc<dyn Hittable>)."""
    spec = SceneSpec()
    rgb = lambda r, g, b: spec.add("tex_solid_rgb", r, g, b)
    floor = spec.add("mat_diffuse", spec.add("tex_checker", 0.8, rgb(0.2, 0.3, 0.1), rgb(0.9, 0.9, 0.9)), -1)
    metal = spec.add("mat_metal", rgb(0.9, 0.8, 0.6), spec.add("tex_solid_f", 0.15))
    glass = spec.add("mat_glass", rgb(1.0, 1.0, 1.0), spec.add("tex_solid_f", 0.05), 0.0, 1.5)
    spec.add("world_add_object", spec.add("quad", (-8.0, 0.0, -8.0), (0.0, 0.0, 16.0), (16.0, 0.0, 0.0), floor))
    P, I = icosphere(1)
    mesh = spec.add("mesh", 0.6, P, I, None, None, metal)
    placements = [((0.0, 1.0, 0.0), 0.3, (-2.0, 0.7, 0.0)), ((1.0, 0.0, 0.0), 1.1, (0.0, 0.7, 0.5)), ((0.0, 0.0, 1.0), -0.7, (2.0, 0.7, -0.5))]
    for axis, angle, tr in placements:
        m = mesh if shared else spec.add("mesh", 0.6, P, I, None, None, metal)
        spec.add("world_add_object", spec.add("instance", m, axis, angle, tr))
    m = mesh if shared else spec.add("mesh", 0.6, P, I, None, None, metal)
    inner = spec.add("instance", m, (0.0, 1.0, 0.0), 0.5, (0.3, 0.0, 0.0))
    spec.add("world_add_object", spec.add("instance", inner, (1.0, 0.0, 0.0), -0.4, (0.0, 2.2, 1.0)))
    box = spec.add("instance", spec.add("cuboid", (0.0, 0.0, 0.0), (0.7, 0.9, 0.7), glass), (0.0, 1.0, 0.0), 0.6, (0.0, 0.0, 0.0))
    spec.add("world_add_object", spec.add("instance", box, (0.0, 1.0, 0.0), 0.3, (-1.0, 0.0, 2.2)))
    lm = spec.add("mat_light", rgb(9.0, 8.0, 7.0))
    lq = spec.add("instance", spec.add("quad", (-0.6, 0.0, -0.6), (1.2, 0.0, 0.0), (0.0, 0.0, 1.2), lm), (1.0, 0.0, 0.0), 0.2, (0.0, 4.0, 0.0))
    spec.add("world_add_light", spec.add("instance", lq, (0.0, 0.0, 1.0), -0.15, (0.5, 0.0, 0.5)))
    spec.add("world_build")
    spec.camera = default_camera(width=64, look_from=(0.0, 2.0, -6.5), look_at=(0.0, 0.9, 0.0), vfov=45.0, env_color=(0.05, 0.06, 0.09))
    return spec


def free_placement_scene():
    """World::add_object / add_light and Instance::new take any Arc<dyn Hittable> (world.rs:18-24, instance.rs:20-30): ONE
    sphere, ONE cuboid and ONE mesh object are each placed in the world directly AND under instances (the mesh under two, one
    of them nested), the sphere is added to the world twice (coincident placements: every hit of it is an exact t tie, decided
    by the per-placement ids), and one quad is both a light of the lights list and, instanced, an ordinary object."""
    spec = SceneSpec()
    rgb = lambda r, g, b: spec.add("tex_solid_rgb", r, g, b)
    floor = spec.add("mat_diffuse", spec.add("tex_checker", 0.7, rgb(0.25, 0.2, 0.3), rgb(0.9, 0.9, 0.85)), -1)
    metal = spec.add("mat_metal", rgb(0.85, 0.8, 0.5), spec.add("tex_solid_f", 0.2))
    glass = spec.add("mat_glass", rgb(1.0, 1.0, 1.0), spec.add("tex_solid_f", 0.1), 0.0, 1.5)
    red = spec.add("mat_diffuse", rgb(0.8, 0.2, 0.15), -1)
    spec.add("world_add_object", spec.add("quad", (-8.0, 0.0, -8.0), (0.0, 0.0, 16.0), (16.0, 0.0, 0.0), floor))
    ball = spec.add("sphere", 0.5, (-2.2, 0.5, 0.0), (-2.2, 0.5, 0.0), metal)
    spec.add("world_add_object", ball)
    spec.add("world_add_object", ball)                                                       # the same Arc twice
    spec.add("world_add_object", spec.add("instance", ball, (0.0, 1.0, 0.0), 0.4, (0.3, 0.6, 1.5)))
    box = spec.add("cuboid", (0.0, 0.0, 0.0), (0.6, 0.8, 0.6), glass)
    spec.add("world_add_object", box)
    spec.add("world_add_object", spec.add("instance", box, (0.0, 1.0, 0.0), 0.7, (1.2, 0.0, -1.0)))
    P, I = icosphere(1)
    mesh = spec.add("mesh", 0.55, P, I, None, None, red)
    spec.add("world_add_object", mesh)                                                       # mesh vertices sit around the origin
    inner = spec.add("instance", mesh, (1.0, 0.0, 0.0), 0.9, (0.0, 1.4, 0.0))
    spec.add("world_add_object", inner)
    spec.add("world_add_object", spec.add("instance", inner, (0.0, 0.0, 1.0), -0.5, (2.0, 0.2, 0.6)))
    lm = spec.add("mat_light", rgb(10.0, 9.0, 8.0))
    lq = spec.add("quad", (-0.7, 3.5, -0.7), (1.4, 0.0, 0.0), (0.0, 0.0, 1.4), lm)
    spec.add("world_add_light", lq)
    spec.add("world_add_object", spec.add("instance", lq, (0.0, 0.0, 1.0), 0.3, (-3.0, 0.5, 1.0)))
    spec.add("world_build")
    spec.camera = default_camera(width=64, look_from=(0.0, 2.2, -6.0), look_at=(0.0, 0.8, 0.0), vfov=48.0, env_color=(0.06, 0.07, 0.1))
    return spec


def white_furnace_scene(width=256, aspect=1.0):
    """A closed-form property of the whole pipeline (no oracle needed, any image size): white Lambertian objects — a floor, a sphere,
    a cuboid and a mesh — under a CONSTANT environment and no lights list. A cosine-sampled diffuse bounce has eval / pdf = albedo = 1
    (diffuse.rs:53-63 over camera.rs:207-213), Russian roulette keeps a path of luminance 1 with probability 1 (camera.rs:190-196),
    so every sample carries exactly the environment's colour out of the scene, whatever it bounced off. No Instance on purpose: under
    a rotation the reference keeps the shading normal in object space (instance.rs:49-53), samples some directions INTO
    the surface and loses those paths inside the closed object — 1.6 % of the samples of this scene when the cuboid and the mesh are
    instanced (measured on the oracle); that is the reference's behaviour, reproduced, and not what this test is about."""
    spec = SceneSpec()
    white = spec.add("mat_diffuse", spec.add("tex_solid_rgb", 1.0, 1.0, 1.0), -1)
    spec.add("world_add_object", spec.add("quad", (-6.0, 0.0, -6.0), (0.0, 0.0, 12.0), (12.0, 0.0, 0.0), white))
    # (nothing touches anything: in the cusp between a resting sphere and the floor a path bounces until max_depth cuts it — seen)
    spec.add("world_add_object", spec.add("sphere", 0.8, (-1.3, 1.5, 0.3), (-1.3, 1.5, 0.3), white))
    spec.add("world_add_object", spec.add("cuboid", (0.4, 0.5, -0.8), (1.3, 1.8, 0.1), white))
    P, I = icosphere(2)
    P = (np.asarray(P, dtype=np.float64) + np.array([0.0, 5.5, 0.6])).astype(np.float32)     # (scaled by 0.6 below: centre (0, 3.3, 0.36))
    spec.add("world_add_object", spec.add("mesh", 0.6, P, I, None, None, white))
    spec.add("world_build")
    spec.camera = default_camera(width=width, aspect=aspect, look_from=(0.0, 1.8, -5.5), look_at=(0.0, 0.9, 0.0), vfov=50.0,
                                 env_color=(0.7, 0.8, 0.9))
    return spec
