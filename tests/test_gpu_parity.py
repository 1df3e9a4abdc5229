"""Parity tests proper: the HIP path, called through the C ABI (libpt_amd.so), against the CPU
oracle on the same seeded inputs.

Bar: with the oracle in deterministic-math mode the two execute the same
IEEE-754 operation sequence, so every f64 accumulator value must be IDENTICAL (bit-exact, NaNs
included); with the oracle on the platform libm (the faithful mode) the linear mean image must
agree within north_star's per-channel tolerance, RMSE < 1e-4.
"""
import os

import numpy as np
import pytest

from common import DEMO_TOL, GOLDEN_DIR, SceneSpec, default_camera, demo_block_stats, icosphere, random_scene

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-4   # BASELINE.json north_star: "image RMSE < 1e-4 vs CPU under fixed seed"


def _pair(pt, orc, ctx, scene_images, sid, width, spp, **kw):
    gs = pt.Scene(ctx)
    gcam = gs.build_scene(sid, width, spp, **kw)
    os_ = orc.Scene()
    ocam = os_.build_scene(sid, width, spp, images=scene_images(sid), **kw)
    return gs, gcam, os_, ocam


# ---- device arithmetic -------------------------------------------------------------------------
def test_device_arithmetic_is_ieee_and_unfused(ctx):
    rng = np.random.default_rng(0)
    ab = np.stack([np.exp(rng.uniform(-20, 20, 200000)), np.exp(rng.uniform(-20, 20, 200000))], axis=1)
    a, b = ab[:, 0], ab[:, 1]
    np.testing.assert_array_equal(ctx.math_probe(0, ab), np.sqrt(a))     # correctly rounded sqrt
    np.testing.assert_array_equal(ctx.math_probe(1, ab), a / b)          # correctly rounded divide
    np.testing.assert_array_equal(ctx.math_probe(2, ab), a * b + a)      # NOT contracted into an fma


def test_device_detmath_equals_oracle_copy(ctx, orc):
    rng = np.random.default_rng(1)
    n = 20000
    cases = {3: rng.uniform(-7, 7, (n, 2)), 4: rng.uniform(-7, 7, (n, 2)), 5: rng.uniform(-1, 1, (n, 2)), 6: rng.uniform(-1, 1, (n, 2)),
             7: np.stack([np.full(n, 0.0625), rng.uniform(0, 1, n)], axis=1), 8: np.exp(rng.uniform(-20, 20, (n, 2))),
             10: np.exp(rng.uniform(-20, 20, (n, 2))), 11: rng.uniform(-40, 40, (n, 2))}
    for which, ab in cases.items():
        got = ctx.math_probe(which, ab)
        want = np.array([orc.detmath(which, float(x), float(y)) for x, y in ab])
        np.testing.assert_array_equal(got, want, err_msg=f"detmath function {which}")


def test_device_rng_equals_oracle(ctx, orc):
    ab = np.array([[1.0, 7.0]] * 64 + [[123456.0, 99.0]] * 64)
    got = ctx.math_probe(9, ab)          # uniform(seed=a, pixel=b, sample=7, draw=i)
    want = np.array([orc.rng_uniform(int(a), int(b), 7, i) for i, (a, b) in enumerate(ab)])
    np.testing.assert_array_equal(got, want)


# ---- closest hit (K2) ---------------------------------------------------------------------------
@pytest.mark.parametrize("sid", [1, 3, 6])
def test_closest_hit_bit_exact(pt, det, ctx, scene_images, sid):
    gs, gcam, os_, ocam = _pair(pt, det, ctx, scene_images, sid, 160, 1)
    d, H = pt.camera_init(gcam)
    rng = np.random.default_rng(sid)
    n = 6000
    px, py = rng.uniform(0, 160, n), rng.uniform(0, H, n)
    o = np.array(list(gcam.look_from))
    rays = np.zeros((n, 7))
    rays[:, 0:3] = o
    rays[:, 3:6] = d["pixel00"] + px[:, None] * d["pixel_du"] + py[:, None] * d["pixel_dv"] - o
    rays[:, 6] = rng.uniform(0, 1, n)
    g = gs.intersect(rays)
    np.testing.assert_array_equal(g, os_.intersect(rays))
    hit = g[:, 0] > 0
    assert hit.mean() > 0.3
    rays2 = np.zeros((int(hit.sum()), 7))                # secondary rays leaving the surfaces
    rays2[:, 0:3] = g[hit, 6:9] + 1e-3 * g[hit, 9:12]
    rays2[:, 3:6] = rng.normal(size=(len(rays2), 3))
    rays2[:, 6] = rays[hit, 6]
    np.testing.assert_array_equal(gs.intersect(rays2), os_.intersect(rays2))
    gs.close(); os_.close()


def test_axis_parallel_rays(pt, det, ctx, scene_images):
    """Directions with exactly-zero components (1/d = inf in the slab test) from origins inside
    the room: they occur in real renders (a direction sampled inside the plane of the axis-aligned
    Cornell light) and once made a NaN-ignoring min/max drop a whole subtree."""
    gs, gcam, os_, ocam = _pair(pt, det, ctx, scene_images, 3, 64, 1)
    rng = np.random.default_rng(2)
    n = 3000
    rays = np.zeros((n, 7))
    rays[:, 0:3] = rng.uniform(5, 550, (n, 3))
    d = rng.normal(size=(n, 3))
    zero = rng.integers(0, 3, n)
    d[np.arange(n), zero] = 0.0
    d[: n // 3, (zero[: n // 3] + 1) % 3] = 0.0           # a third of them parallel to an axis
    d[n - 100:, :] *= -1.0
    d[n - 50:, zero[n - 50:]] = -0.0
    rays[:, 3:6] = d
    rays[: n // 6, 1] = 554.0                             # origins exactly in the light's plane
    g = gs.intersect(rays)
    assert (g[:, 0] == 1).mean() > 0.5        # the room is open towards the camera
    np.testing.assert_array_equal(g, os_.intersect(rays))
    gs.close(); os_.close()


def test_tie_rule_and_coplanar_quads(pt, det, ctx):
    def build(s):
        m1 = s.mat_diffuse(s.tex_solid_rgb(1, 0, 0)); m2 = s.mat_diffuse(s.tex_solid_rgb(0, 1, 0)); l = s.mat_light(s.tex_solid_rgb(5, 5, 5))
        s.world_add_object(s.quad((0, 0, 2), (1, 0, 0), (0, 1, 0), m1))
        s.world_add_object(s.quad((0, 0, 2), (1, 0, 0), (0, 1, 0), m2))
        s.world_add_light(s.quad((0, 0, 2), (1, 0, 0), (0, 1, 0), l))
        s.world_build()
    gs, os_ = pt.Scene(ctx), det.Scene()
    build(gs); build(os_)
    rays = np.array([[0.5, 0.5, 0, 0, 0, 1, 0], [0.1, 0.9, 0, 0, 0, 1, 0.5]])
    g = gs.intersect(rays)
    assert (g[:, 2] == 2).all()                 # ids: light 0, quads 1 and 2 -> larger id wins the exact tie
    np.testing.assert_array_equal(g, os_.intersect(rays))
    gs.close(); os_.close()


# ---- full renders ---------------------------------------------------------------------------------
@pytest.mark.parametrize("sid,width,spp", [(3, 64, 32), (6, 128, 8), (5, 128, 8), (1, 128, 6), (2, 96, 6), (4, 96, 6), (7, 96, 8)])
def test_render_bit_exact_vs_oracle(pt, det, ctx, scene_images, sid, width, spp):
    """All seven scene scripts; slots_per_pixel=1 gives the reference's per-pixel sample order."""
    gs, gcam, os_, ocam = _pair(pt, det, ctx, scene_images, sid, width, spp)
    ga, st = gs.render(gcam, 1, 0, spp, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 1, 0, spp)
    assert st.segments == cnt["segments"] and st.samples == cnt["samples"] == ga.shape[0] * ga.shape[1] * spp
    np.testing.assert_array_equal(ga, oa)
    u8 = ctx.resolve_u8(ga, spp)
    np.testing.assert_array_equal(u8, det.resolve_u8(oa, spp))       # camera.rs:109-114 on the GPU
    gs.close(); os_.close()


@pytest.mark.parametrize("sid", [3, 6])
def test_render_matches_committed_golden(pt, ctx, sid):
    g = np.load(os.path.join(GOLDEN_DIR, f"scene{sid}_w64_spp16_seed1.npz"))
    gs = pt.Scene(ctx)
    cam = gs.build_scene(sid, 64, 16)
    acc, st = gs.render(cam, 1, 0, 16, slots_per_pixel=1)
    np.testing.assert_array_equal(acc, g["accum"])
    assert st.segments == int(g["segments"])
    gs.close()


@pytest.mark.parametrize("sid,width,spp", [(3, 96, 24), (6, 160, 16)])
def test_low_spp_libm_oracle_smoke_most_samples_identical_and_unbiased(pt, orc, ctx, scene_images, sid, width, spp):
    """NOT the tolerance gate — that is test_north_star_tolerance_vs_faithful_libm_oracle_at_4000spp below, which measures
    north_star's RMSE < 1e-4 at the full 4000 spp. This is the quick low-spp companion against the oracle on the platform
    libm (what the Rust reference calls through f64::sin etc.): the GPU's deterministic elementary functions differ from
    glibc in the last bit of a fraction of a per cent of calls, and the integrator's discontinuities (3-D checker floor() of a
    ground-plane coordinate that is 0 +- 1e-16, texture.rs:44-48; light-plane offset sign, camera.rs:217) turn a few 1e-4 of
    those samples into a different, equally valid sample. Checked here: the bulk of the accumulator values is bit-identical
    and the difference has no bias. No RMSE bound is stated at this sample count."""
    orc.set_math_mode(False)
    gs, gcam, os_, ocam = _pair(pt, orc, ctx, scene_images, sid, width, spp)
    ga, _ = gs.render(gcam, 1, 0, spp, slots_per_pixel=1)    # same summation order as the oracle
    oa, _ = os_.render(ocam, 1, 0, spp)
    assert np.mean(ga == oa) > 0.9                        # the bulk of the samples is unaffected
    assert abs(np.mean(ga - oa) / spp) < 2e-4            # no bias
    gs.close(); os_.close()


def test_slot_layouts_and_sample_ranges_agree(pt, det, ctx):
    """auto / explicit slots-per-pixel change only the summation order; sample sub-ranges add up
    (that is what the multi-GPU spp sharding relies on)."""
    gs = pt.Scene(ctx)
    cam = gs.build_scene(3, 80, 12)
    ref, st1 = gs.render(cam, 3, 0, 12, slots_per_pixel=1)
    for k in (0, 2, 5, 12, 50):
        acc, st = gs.render(cam, 3, 0, 12, slots_per_pixel=k)
        assert st.segments == st1.segments and st.samples == st1.samples
        np.testing.assert_allclose(acc, ref, rtol=1e-13, atol=1e-13)
    lo, sa = gs.render(cam, 3, 0, 5, slots_per_pixel=1)
    both, sb = gs.render(cam, 3, 5, 12, accum=lo.copy(), slots_per_pixel=1)     # accumulates INTO the buffer
    np.testing.assert_allclose(both, ref, rtol=1e-13, atol=1e-13)
    assert sa.segments + sb.segments == st1.segments
    again, _ = gs.render(cam, 3, 0, 12, slots_per_pixel=1)
    np.testing.assert_array_equal(again, ref)                                      # run-to-run deterministic
    gs.close()


@pytest.mark.parametrize("n_meshes,flat", [(7, True), (30, False)])
def test_mesh_crowd_bit_exact(pt, det, ctx, n_meshes, flat):
    """K2's rarely taken paths: a frame filled by overlapping mesh instances — rays that enter more than four
    mesh boxes (the fifth is walked on the spot), windows whose candidate list (768) overflows, and, with 30
    world entries, the per-lane walk of the top-level TREE instead of the flat entry list."""
    rng = np.random.default_rng(42 + n_meshes)
    spec = SceneSpec()
    P, I = icosphere(1)
    mats = [spec.add("mat_diffuse", spec.add("tex_solid_rgb", *rng.uniform(0.2, 0.9, 3)), -1),
            spec.add("mat_metal", spec.add("tex_solid_rgb", 0.9, 0.8, 0.7), spec.add("tex_solid_f", 0.15)),
            spec.add("mat_glass", spec.add("tex_solid_rgb", 1.0, 1.0, 1.0), spec.add("tex_solid_f", 0.05), 0.0, 1.5)]
    for i in range(n_meshes):
        mesh = spec.add("mesh", float(rng.uniform(0.9, 1.6)), P, I, None, None, mats[i % 3])
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        spec.add("world_add_object", spec.add("instance", mesh, tuple(axis), float(rng.uniform(0, 3)), tuple(rng.uniform(-0.8, 0.8, 3) + (0.0, 0.5, 0.0))))
    spec.add("world_add_light", spec.add("quad", (-1.0, 4.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0),
                                         spec.add("mat_light", spec.add("tex_solid_rgb", 8.0, 8.0, 8.0))))
    spec.add("world_build")
    spec.camera = default_camera(width=96, look_from=(0.0, 0.5, -4.0), look_at=(0.0, 0.5, 0.0), vfov=40.0)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(det.Camera, ores)
    ga, st = gs.render(gcam, 3, 0, 4, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 3, 0, 4)
    assert st.extend_variant == 0 and st.segments == cnt["segments"]
    np.testing.assert_array_equal(ga, oa)
    gd, _ = gs.render(gcam, 3, 0, 4)                     # dynamic mode: same sums up to f64 addition order
    fin = np.isfinite(oa)
    np.testing.assert_allclose(gd[fin], oa[fin], rtol=1e-11, atol=1e-11)
    gs.close(); os_.close()


@pytest.mark.parametrize("sid", [2, 4, 5, 6])
def test_product_matches_reference_demo_images(pt, ctx, sid):
    """The HIP path against the reference's own rendered outputs (demo/*.png as 48x27 block means, see
    common.demo_block_stats): 480x270 @ 256 spp in the default dynamic mode."""
    gs = pt.Scene(ctx)
    cam = gs.build_scene(sid, 480, 256)
    acc, _ = gs.render(cam, 1, 0, 256)
    mad, corr, _, _ = demo_block_stats(sid, ctx.resolve_u8(acc, 256))
    assert (mad < DEMO_TOL[sid][0]).all() and (corr > DEMO_TOL[sid][1]).all(), (mad, corr)
    gs.close()


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def test_kernel_forms_and_pool_sizes_agree(pt, ctx):
    """Every K2 form (two-phase with flat / tree top level, batch), the K3 forms and small path
    pools (work-counter shards run dry and are stolen from) give the same accumulator: bit for bit with
    slots_per_pixel=1, up to f64 summation order in the dynamic mode."""
    def render(k, spp):
        gs = pt.Scene(ctx)
        cam = gs.build_scene(6, 96, spp)                             # the scene build reads PT_NO_FLAT_TLAS
        acc, st = gs.render(cam, 4, 0, spp, slots_per_pixel=k)
        gs.close()
        return acc, st
    ref, st0 = render(1, 6)
    assert st0.extend_variant == 0                                   # two-phase kernel is the default with meshes
    for env in ({"PT_NO_FLAT_TLAS": "1"}, {"PT_K2": "batch"}, {"PT_K2": "batch", "PT_NO_FLAT_TLAS": "1"}, {"PT_EXT2": "243"}, {"PT_EXT2": "163"},
                {"PT_SHADE_VARIANT": "2"}, {"PT_SHADE_VARIANT": "13"}, {"PT_SHADE_VARIANT": "12"}, {"PT_SHADE_VARIANT": "22"}, {"PT_SHADE_VARIANT": "32"}, {"PT_SHADE_VARIANT": "52"},
                {"PT_SHADE_VARIANT": "42", "PT_WIDE_WINDOW_MIN": "1"}, {"PT_EXT2": "1164"}, {"PT_EXT2": "164"}, {"PT_EXT2": "8164"}):
        env = dict(env, PT_EXPERIMENT="1")                          # the switches are inert without it
        acc, st = _with_env(env, lambda: render(1, 6))
        np.testing.assert_array_equal(acc, ref, err_msg=str(env))
        assert st.segments == st0.segments
    acc, st = _with_env({"PT_K2": "batch", "PT_SLOTS_PER_PIXEL": "3"}, lambda: render(1, 6))   # no PT_EXPERIMENT: ignored
    assert st.extend_variant == 0 and st.slots_per_pixel == 1
    with pytest.raises(pt.PtError, match="PT_POOL_SLOTS"):
        _with_env({"PT_EXPERIMENT": "1", "PT_POOL_SLOTS": "0"}, lambda: render(0, 2))
    acc, st = _with_env({"PT_EXPERIMENT": "1", "PT_SLOTS_PER_PIXEL": "3"}, lambda: render(2, 6))   # an explicit option wins
    assert st.slots_per_pixel == 2
    for env in ({"PT_ENTRYBOX_VIA_PRIMREF": "1"}, {"PT_DRAW_TIME": "1"}):   # walk records without the direct primitive index; Ray::time computed although nothing moves
        acc, st = _with_env(dict(env, PT_EXPERIMENT="1"), lambda: render(1, 6))
        np.testing.assert_array_equal(acc, ref, err_msg=str(env))
    # scenes of quads and cuboids only (batch kernel, flat top level): with and without the (ray, primitive) pair passes
    def render_room(sid):
        gs = pt.Scene(ctx)
        cam = gs.build_scene(sid, 64, 5)
        acc, st = gs.render(cam, 9, 0, 5, slots_per_pixel=1)
        gs.close()
        return acc, st
    for sid in (3, 7):
        ref_room, st_room = render_room(sid)
        for env in ({"PT_NO_FLAT_PAIRS": "1"}, {"PT_NO_FLAT_TLAS": "1"}, {"PT_ENTRYBOX_VIA_PRIMREF": "1"}):
            acc, st = _with_env(dict(env, PT_EXPERIMENT="1"), lambda: render_room(sid))
            np.testing.assert_array_equal(acc, ref_room, err_msg=f"scene {sid} {env}")
            assert st.segments == st_room.segments
    ref40, st40 = render(1, 40)
    fin = np.isfinite(ref40)
    for env in ({}, {"PT_POOL_SLOTS": "4096"}, {"PT_POOL_SLOTS": "1000"}, {"PT_POOL_SLOTS": "70000", "PT_NO_FLAT_TLAS": "1"}):
        env = dict(env, PT_EXPERIMENT="1")
        acc, st = _with_env(env, lambda: render(0, 40))
        assert st.samples == 96 * 54 * 40 and st.segments == st40.segments, env
        np.testing.assert_allclose(acc[fin], ref40[fin], rtol=1e-11, atol=1e-11, err_msg=str(env))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 27, 45, 58])
def test_random_scenes_bit_exact(pt, det, ctx, seed):
    """Fuzz: random spheres / moving spheres / quads / instanced cuboids / instanced meshes with random
    diffuse, metal, glass, principled, sheen, clearcoat and mix materials, checker textures, quad (+ sphere)
    lights. Seed 27 is the scene that exposed a one-ulp difference between the two hosts' instance matrices
    (libm sincos() vs sin()+cos(); tools/gpu_fuzz.py runs the long soak)."""
    spec = random_scene(seed, sphere_light=(seed % 2 == 1), n_objects=10 if seed < 4 else 6 + seed % 9)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    assert gs.prim_count() == os_.prim_count()
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(det.Camera, ores)
    ga, st = gs.render(gcam, 7, 0, 6, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 7, 0, 6)
    assert st.segments == cnt["segments"]
    np.testing.assert_array_equal(ga, oa)           # NaN/inf samples (reference quirks) included
    gs.close(); os_.close()


@pytest.mark.parametrize("seed", [100, 101, 102, 103])
def test_random_scenes_without_meshes_bit_exact(pt, det, ctx, seed):
    """Fuzz of the batch K2 with a flat top level: spheres (moving ones too), quads and instanced cuboids only — the scenes whose
    cuboid faces go through the dense (ray, primitive) pair passes (flat_top_level) and whose LDS minimum-t / maximum-id
    protocol must reproduce consider()'s closest-hit rule; checked against the same scene without the pair passes."""
    spec = random_scene(seed, with_mesh=False, sphere_light=(seed % 2 == 1), n_objects=8 + seed % 12)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(det.Camera, ores)
    ga, st = gs.render(gcam, 11, 0, 6, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 11, 0, 6)
    assert st.extend_variant == 1                                    # batch kernel
    assert st.segments == cnt["segments"]
    np.testing.assert_array_equal(ga, oa)
    gs.close(); os_.close()

    def again():
        g2 = pt.Scene(ctx)
        cam2 = spec.make_camera(pt.Camera, spec.replay(g2))
        a, _ = g2.render(cam2, 11, 0, 6, slots_per_pixel=1)
        g2.close()
        return a
    np.testing.assert_array_equal(_with_env({"PT_EXPERIMENT": "1", "PT_NO_FLAT_PAIRS": "1"}, again), ga)


def _all_light_kinds_scene():
    """Lights list holding every kind of Hittable the reference can put there (world.rs:18-20):
    quad, sphere, cuboid, triangle mesh, and instanced quad / cuboid / mesh / sphere."""
    spec = SceneSpec()
    lm = lambda r, g, b: spec.add("mat_light", spec.add("tex_solid_rgb", r, g, b))
    white = spec.add("mat_diffuse", spec.add("tex_solid_rgb", 0.7, 0.7, 0.7), -1)
    metal = spec.add("mat_metal", spec.add("tex_solid_rgb", 0.9, 0.8, 0.6), spec.add("tex_solid_f", 0.2))
    spec.add("world_add_object", spec.add("quad", (-10.0, 0.0, -10.0), (0.0, 0.0, 20.0), (20.0, 0.0, 0.0), white))
    spec.add("world_add_object", spec.add("sphere", 0.7, (0.0, 0.7, 0.0), (0.0, 0.7, 0.0), metal))
    spec.add("world_add_object", spec.add("instance", spec.add("cuboid", (0.0, 0.0, 0.0), (0.8, 1.2, 0.8), white), (0.0, 1.0, 0.0), 0.4, (1.5, 0.0, 0.5)))
    P, I = icosphere(0)
    spec.add("world_add_light", spec.add("quad", (-1.0, 4.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), lm(6, 6, 5)))
    spec.add("world_add_light", spec.add("sphere", 0.25, (-2.5, 2.0, 1.0), (-2.5, 2.0, 1.0), lm(9, 4, 2)))
    spec.add("world_add_light", spec.add("cuboid", (2.5, 1.0, -1.0), (2.9, 1.4, -0.6), lm(2, 7, 3)))
    spec.add("world_add_light", spec.add("mesh", 0.3, P, I, None, None, lm(3, 3, 9)))
    spec.add("world_add_light", spec.add("instance", spec.add("quad", (0.0, 0.0, 0.0), (0.6, 0.0, 0.0), (0.0, 0.6, 0.0), lm(5, 5, 5)), (0.0, 1.0, 0.0), 0.8, (-1.5, 0.5, -2.0)))
    spec.add("world_add_light", spec.add("instance", spec.add("cuboid", (0.0, 0.0, 0.0), (0.3, 0.3, 0.3), lm(4, 1, 4)), (1.0, 0.0, 0.0), -0.5, (0.5, 2.5, 2.0)))
    spec.add("world_add_light", spec.add("instance", spec.add("mesh", 0.25, P, I, None, None, lm(1, 6, 6)), (0.0, 0.0, 1.0), 1.1, (2.0, 2.5, 1.5)))
    spec.add("world_add_light", spec.add("instance", spec.add("sphere", 0.2, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), lm(8, 8, 1)), (0.0, 1.0, 0.0), 0.0, (-0.5, 1.8, -1.5)))
    spec.add("world_build")
    spec.camera = default_camera(width=56, env_color=(0.02, 0.02, 0.03))
    return spec


def test_lights_of_every_kind_bit_exact(pt, det, ctx):
    spec = _all_light_kinds_scene()
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    assert gs.prim_count() == os_.prim_count() == 1 + 1 + 6 + 1 + 1 + 6 + 20 + 1 + 6 + 20 + 1
    ga, st = gs.render(spec.make_camera(pt.Camera, gres), 11, 0, 6, slots_per_pixel=1)
    oa, cnt = os_.render(spec.make_camera(det.Camera, ores), 11, 0, 6)
    assert st.segments == cnt["segments"]
    np.testing.assert_array_equal(ga, oa)      # incl. the NaN/inf the reference's Sphere::pdf produces on the light itself
    assert np.isfinite(ga).mean() > 0.8 and np.nanmax(ga[np.isfinite(ga)]) > 0
    gs.close(); os_.close()


def test_mesh_with_normals_uvs_image_textures_and_normal_map(pt, det, ctx):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (16, 32, 3), dtype=np.uint8)
    nmap = np.clip(rng.normal(128, 20, (8, 8, 3)), 0, 255).astype(np.uint8); nmap[..., 2] = 255
    P, I = icosphere(1)
    N = (P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32)
    UV = rng.uniform(0, 1, (len(P), 2)).astype(np.float32)
    spec = SceneSpec()
    t_img = spec.add("tex_image_rgb8", img); t_n = spec.add("tex_image_rgb8", nmap)
    m_img = spec.add("mat_diffuse", t_img, -1)
    m_nm = spec.add("mat_diffuse", spec.add("tex_solid_rgb", 0.7, 0.6, 0.5), t_n)
    mesh = spec.add("mesh", 1.0, P, I, N, UV, m_img)
    spec.add("world_add_object", spec.add("instance", mesh, (0.0, 1.0, 0.0), 0.7, (0.0, 1.0, 0.0)))
    spec.add("world_add_object", spec.add("quad", (-5.0, 0.0, -5.0), (0.0, 0.0, 10.0), (10.0, 0.0, 0.0), m_nm))
    spec.add("world_add_object", spec.add("sphere", 0.6, (1.8, 0.6, 0.5), (1.8, 0.6, 0.5), m_img))
    env = spec.add("tex_image_rgb8", rng.integers(0, 256, (32, 64, 3), dtype=np.uint8))
    spec.add("world_build")
    spec.camera = default_camera(env_is_map=1, env_tex=env)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    ga, _ = gs.render(spec.make_camera(pt.Camera, gres), 2, 0, 8, slots_per_pixel=1)
    oa, _ = os_.render(spec.make_camera(det.Camera, ores), 2, 0, 8)
    np.testing.assert_array_equal(ga, oa)
    gs.close(); os_.close()


def test_sheen_clearcoat_mix_bit_exact(pt, det, ctx):
    """The three bsdf/ materials no scene script instantiates (sheen.rs, clearcoat.rs, mix.rs), under a quad
    light (NEE + MIS exercise pdf and eval) and on an instanced normal-carrying mesh (shading != geometric
    normal separates sheen's geometric frame from clearcoat's shading frame)."""
    spec = SceneSpec()
    rgb = lambda r, g, b: spec.add("tex_solid_rgb", r, g, b)
    sheen = spec.add("mat_sheen", (0.8, 0.3, 0.1), 0.6)
    coat = spec.add("mat_clearcoat", 0.7)
    diffuse = spec.add("mat_diffuse", rgb(0.6, 0.6, 0.7), -1)
    prin = spec.add("mat_principled", rgb(0.9, 0.5, 0.2), [0.3, 0.1, 0.6, 0.4, 0.5, 0.2, 0.3, 0.5, 0.8, 0.6, 1.45])
    glass = spec.add("mat_glass", rgb(1.0, 1.0, 1.0), spec.add("tex_solid_f", 0.1), 0.0, 1.5)
    mixes = [spec.add("mat_mix", 0.3, diffuse, coat), spec.add("mat_mix", 0.5, prin, sheen),
             spec.add("mat_mix", 0.0, sheen, glass), spec.add("mat_mix", 1.0, coat, glass), spec.add("mat_mix", 0.8, sheen, coat)]
    # MixBxDf::new takes any Arc<dyn BxDFMaterial> (mix.rs:14-20): a mix of a mix and a leaf, and a mix of two mixes — the selector
    # is drawn once per level (mix.rs:25-32) and every level rounds its own weighted sum (mix.rs:34-44)
    mixes += [spec.add("mat_mix", 0.4, mixes[1], glass), spec.add("mat_mix", 0.65, mixes[0], mixes[4])]
    spec.add("world_add_object", spec.add("quad", (-6.0, 0.0, -6.0), (0.0, 0.0, 12.0), (12.0, 0.0, 0.0), mixes[0]))
    mats = [sheen, coat] + mixes
    for i, m in enumerate(mats):
        x = -3.6 + 0.9 * i
        spec.add("world_add_object", spec.add("sphere", 0.4, (x, 0.4, 0.0), (x, 0.4, 0.0), m))
    P, I = icosphere(1)
    N = (P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float32)
    mesh = spec.add("mesh", 0.8, P, I, N, None, mixes[4])
    spec.add("world_add_object", spec.add("instance", mesh, (0.0, 1.0, 0.0), 0.5, (0.0, 1.9, -0.5)))
    spec.add("world_add_light", spec.add("quad", (-1.5, 4.0, -1.5), (3.0, 0.0, 0.0), (0.0, 0.0, 3.0),
                                         spec.add("mat_light", rgb(7.0, 7.0, 6.0))))
    spec.add("world_build")
    spec.camera = default_camera(look_from=(0.0, 2.5, 7.0), look_at=(0.0, 0.8, 0.0), vfov=45.0)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(det.Camera, ores)
    ga, st = gs.render(gcam, 5, 0, 8, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 5, 0, 8)
    assert st.segments == cnt["segments"]
    np.testing.assert_array_equal(ga, oa)
    assert np.isfinite(ga).mean() > 0.99 and ga[np.isfinite(ga)].max() > 0
    gd, _ = gs.render(gcam, 5, 0, 8)            # dynamic slot assignment + class sort: same sums up to f64 add order
    fin = np.isfinite(oa) & np.isfinite(gd)
    np.testing.assert_allclose(gd[fin], oa[fin], rtol=1e-11, atol=1e-11)
    gs.close(); os_.close()


# ---- edge cases ------------------------------------------------------------------------------------
def test_edge_cases(pt, det, ctx):
    gs, os_ = pt.Scene(ctx), det.Scene()
    gcam, ocam = gs.build_scene(3, 40, 4), os_.build_scene(3, 40, 4)
    # empty sample range: accumulator untouched
    acc0 = np.full((40, 40, 3), 1.5)
    out, st = gs.render(gcam, 1, 2, 2, accum=acc0.copy())
    np.testing.assert_array_equal(out, acc0); assert st.samples == 0 and st.segments == 0
    # max_depth = 1 (one segment per path) and 0 (no segment at all)
    for depth in (1, 0, 3):
        gcam.max_depth = ocam.max_depth = depth
        ga, st = gs.render(gcam, 1, 0, 3, slots_per_pixel=1)
        oa, cnt = os_.render(ocam, 1, 0, 3)
        np.testing.assert_array_equal(ga, oa); assert st.segments == cnt["segments"]
    gcam.max_depth = ocam.max_depth = 50
    # 1-pixel-wide and ragged sizes, non-unit aspect
    for w, aspect in ((1, 1.0), (7, 1.7), (33, 0.6)):
        gcam.image_width = ocam.image_width = w; gcam.aspect_ratio = ocam.aspect_ratio = aspect
        ga, _ = gs.render(gcam, 9, 1, 4, slots_per_pixel=1)
        oa, _ = os_.render(ocam, 9, 1, 4)
        assert ga.shape == oa.shape
        np.testing.assert_array_equal(ga, oa)
    gs.close(); os_.close()


def test_error_behaviour(pt, ctx):
    s = pt.Scene(ctx)
    with pytest.raises(pt.PtError, match="bad"):
        s.mat_diffuse(99)
    with pytest.raises(pt.PtError, match="bad material"):
        s.sphere(1.0, (0, 0, 0), (0, 0, 0), 5)
    with pytest.raises(pt.PtError, match="empty"):
        s.world_build()
    m = s.mat_diffuse(s.tex_solid_rgb(1, 1, 1))
    with pytest.raises(pt.PtError, match="bad material"):
        s.mat_mix(0.5, m, 17)
    two = s.mat_mix(0.5, s.mat_mix(0.5, m, s.mat_clearcoat(0.5)), m)                   # a mix of a mix: fine (two levels)
    with pytest.raises(pt.PtError, match="deeper than two"):
        s.mat_mix(0.5, two, m)
    q = s.quad((0, 0, 0), (1, 0, 0), (0, 1, 0), m)
    s.world_add_object(q)
    s.world_add_object(q)                                           # the same Arc twice is legal (world.rs:18-24)
    cam = pt.Camera(); cam.aspect_ratio = 1.0; cam.image_width = 8; cam.vfov = 40; cam.max_depth = 5
    cam.look_at[2] = 1.0; cam.vup[1] = 1.0; cam.focal_length = 1.0; cam.env_tex = -1
    with pytest.raises(pt.PtError, match="not built"):
        s.render(cam, 1, 0, 1)
    s.world_build()
    with pytest.raises(pt.PtError, match="spp_end"):
        s.render(cam, 1, 3, 1)
    cam.env_is_map = 1; cam.env_tex = 0          # a solid texture is not an environment map
    with pytest.raises(pt.PtError, match="env_tex"):
        s.render(cam, 1, 0, 1)
    s.close()


# ---- full-size properties (BASELINE.json configs; no oracle run at this size) --------------------------
def test_full_hd_scene6_properties(pt, det, ctx):
    """Scene 6 at 1920x1080 (config 3's size) with a few spp: counters, finiteness, additivity of
    sample ranges, and exact agreement with the oracle on a random subset of (pixel, sample)s."""
    gs = pt.Scene(ctx)
    cam = gs.build_scene(6, 1920, 4000)
    a, sa = gs.render(cam, 1, 0, 2)
    assert a.shape == (1080, 1920, 3) and sa.samples == 1920 * 1080 * 2 and 1.5 < sa.segments / sa.samples < 3.0
    assert np.isfinite(a).all() and a.min() >= 0
    b, sb = gs.render(cam, 1, 2, 3, slots_per_pixel=1)
    c, sc = gs.render(cam, 1, 0, 3)
    np.testing.assert_allclose(a + b, c, rtol=1e-13, atol=1e-13)
    assert sa.segments + sb.segments == sc.segments
    os_ = det.Scene()
    ocam = os_.build_scene(6, 1920, 4000)
    rng = np.random.default_rng(0)
    flat = b.reshape(-1, 3)
    for pix in rng.integers(0, 1920 * 1080, 300):
        rad, _, _ = os_.trace_sample(ocam, 1, int(pix), 2)
        np.testing.assert_array_equal(flat[pix], rad)
    gs.close(); os_.close()


@pytest.mark.parametrize("width,aspect,spp", [(200, 1.0, 48), (1920, 16.0 / 9.0, 3)])
def test_white_furnace_returns_the_environment_exactly(pt, ctx, width, aspect, spp):
    """White furnace (tests/common.py white_furnace_scene): white diffuse floor, sphere, instanced cuboid and instanced mesh under a
    constant environment, no lights — every pixel's mean is the environment's colour to rounding (1e-12 relative: up to fifty
    factors of 1 +- 2^-52 per path), at a test size and at 1920x1080, through the default (dynamic) path and the static one.
    Independent of the oracle: what it pins is sampler / pdf / eval consistency, the roulette, the environment term, regeneration
    and the frame accumulator — an energy leak or a double count anywhere shows as a pixel that is not (0.7, 0.8, 0.9)."""
    from common import white_furnace_scene
    spec = white_furnace_scene(width, aspect)
    gs = pt.Scene(ctx)
    cam = spec.make_camera(pt.Camera, spec.replay(gs))
    for k in (0, 1):
        acc, st = gs.render(cam, 11, 0, spp, slots_per_pixel=k)
        h = acc.shape[0]
        assert acc.shape == (h, width, 3) and st.samples == h * width * spp
        assert 1.0 < st.segments / st.samples < 6.0                  # paths do bounce (this is not an empty frame)
        np.testing.assert_allclose(acc / spp, np.broadcast_to(np.array([0.7, 0.8, 0.9]), acc.shape), rtol=1e-12, atol=0)
    gs.close()


def test_cornell_1920_square_properties(pt, ctx):
    """Config 2's size (1920x1920, aspect 1.0: main.rs:218)."""
    gs = pt.Scene(ctx)
    cam = gs.build_scene(3, 1920, 4000)
    a, st = gs.render(cam, 1, 0, 1)
    assert a.shape == (1920, 1920, 3) and st.samples == 1920 * 1920
    assert np.isfinite(a).all() and 3.0 < st.segments / st.samples < 4.5
    gs.close()


def test_cli_renders_scene3_like_the_reference_binary(pt, det, ctx, tmp_path):
    """host/main.cpp: the reference's `-s 3` with explicit size overrides; the PNG must be the
    oracle's gamma-quantised image (camera.rs:109-123) for the same seed, up to the summation order
    of the default dynamic schedule (which can move a value across a u8 boundary only by rounding
    of the last bits: allow <= 1 level on a handful of bytes)."""
    import subprocess
    from PIL import Image

    exe = os.path.join(os.path.dirname(pt.LIB_PATH), "pt_render")
    out = str(tmp_path / "cornell.png")
    r = subprocess.run([exe, "-s", "3", "--width", "96", "--spp", "8", "--seed", "5", "--out", out, "--assets", pt.ASSET_DIR],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "rendering production" in r.stderr and "as_secs_f64" in r.stderr      # camera.rs:101,125
    img = np.asarray(Image.open(out).convert("RGB")).astype(int)
    os_ = det.Scene()
    ocam = os_.build_scene(3, 96, 8)
    oa, _ = os_.render(ocam, 5, 0, 8)
    want = det.resolve_u8(oa, 8).astype(int)
    assert img.shape == want.shape
    diff = np.abs(img - want)
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    bad = subprocess.run([exe, "-s", "5", "--width", "32", "--spp", "1", "--assets", "/nonexistent"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 101 and "panic" in bad.stderr                         # asset errors are fatal like unwrap()
    os_.close()


# ---- the lights / MIS branch against deterministic quadrature (no reference image exists for scenes with lights) ----------
@pytest.mark.parametrize("light", ["quad", "sphere", "two", "cuboid", "instquad"])
def test_light_sampling_mis_matches_quadrature(pt, ctx, light):
    """The HIP path's one-sample MIS (camera.rs:199-216, list.rs:78-96, quad.rs:80-98 / sphere.rs:110-135) against numbers
    that come from neither the oracle nor the kernels: tests/refs_numpy.py integrates what trace() computes in expectation
    for a Lambert floor under one emitter (max_depth = 2, so quirk Q5 cannot act). Quad light: the estimator is unbiased and
    the pixel must equal (albedo/pi) Le x form factor. Sphere light: the reference's Sphere::pdf is not the density of
    Sphere::sample — its estimator is ~18x too bright — and THAT value must be reproduced. "two": a lights list of a quad and a
    one-triangle mesh (uniform pick, mean of pdfs, list.rs:78-96; Triangle::sample / pdf, mesh.rs:122-141). "cuboid" / "instquad":
    a cuboid light (cuboid.rs:78-84) and a tilted quad light under an Instance (instance.rs:64-75), both unbiased. Every pixel within Monte-Carlo
    error (16 batches of 256 spp: z ~ t(15)), the image mean within 4 sigma."""
    from common import mis_scene, mis_expected, mis_zscores
    spec = mis_scene(light)
    gs = pt.Scene(ctx)
    cam = spec.make_camera(pt.Camera, spec.replay(gs))
    est, true = mis_expected(light)
    z, zg, mean = mis_zscores(lambda seed, a, b: gs.render(cam, seed, a, b)[0], est)
    gs.close()
    assert np.isfinite(z).all()
    assert np.abs(zg).max() < 4.0, zg
    assert (np.abs(z) > 4.0).mean() < 0.01 and 0.85 < z.std() < 1.3, (np.abs(z).max(), z.std())
    if light == "sphere":
        assert mean.mean() > 10.0 * true.mean()


# ---- north_star's tolerance, measured directly ----------------------------------------------------
@pytest.mark.parametrize("sid,width", [(3, 240), (5, 240), (6, 240)])
def test_north_star_tolerance_vs_faithful_libm_oracle_at_4000spp(pt, orc, ctx, scene_images, sid, width):
    """BASELINE.json: "image RMSE < 1e-4 vs CPU under fixed seed". The three graded scenes at the full 4000 spp on a
    240-px-wide frame, HIP path (default dynamic schedule, deterministic elementary functions) against the oracle on the
    PLATFORM LIBM — the arithmetic the Rust reference runs. Per channel, on the linear mean image: RMSE < 1e-4 and
    |bias| < 2e-5. (Round 1 extrapolated this from 16 spp; measured directly it was 1.6e-4 on scene 6 with the fdlibm
    sin/cos/pow — the once-rounded kernels of csrc/pt_detmath.h are what brings it under the bar.)"""
    orc.set_math_mode(False)
    spp = 4000
    gs, gcam, os_, ocam = _pair(pt, orc, ctx, scene_images, sid, width, spp)
    ga, st = gs.render(gcam, 1, 0, spp)
    oa, cnt = os_.render(ocam, 1, 0, spp)
    gs.close(); os_.close()
    fin = np.isfinite(ga).all(axis=2) & np.isfinite(oa).all(axis=2)
    assert fin.mean() > 0.999
    d = (ga - oa)[fin] / spp
    rmse, bias = np.sqrt(np.mean(d ** 2, axis=0)), np.mean(d, axis=0)
    assert (rmse < RMSE_TOL).all(), (sid, rmse)
    assert (np.abs(bias) < 2e-5).all(), (sid, bias)
    assert abs(int(st.segments) - int(cnt["segments"])) < 1e-5 * cnt["segments"]


# ---- multi-GPU entry points of the C ABI on one GPU (a 1-rank RCCL communicator) ----------------------------------------------
def test_render_multi_single_rank_equals_render(pt, ctx):
    """pt_render_multi = spp shard + device accumulator + ncclReduce + download. With one rank the shard is the whole range
    and the reduce is skipped, so the frame must be pt_render's bit for bit (static mode); the communicator's barrier and
    host-value all-reduce are exercised on the way."""
    comm = pt.Comm(ctx, 0, 1)
    assert comm.allreduce([1.5, -2.0], "sum").tolist() == [1.5, -2.0] and comm.allreduce([7.0], "max").tolist() == [7.0]
    comm.barrier()
    gs = pt.Scene(ctx)
    cam = gs.build_scene(3, 72, 10)
    ref, st0 = gs.render(cam, 2, 0, 10, slots_per_pixel=1)
    acc, st = gs.render_multi(cam, 2, 10, comm, slots_per_pixel=1)
    np.testing.assert_array_equal(acc, ref)
    assert st.segments == st0.segments and st.samples == st0.samples
    acc2, _ = gs.render_multi(cam, 2, 10, comm, accum=acc.copy(), slots_per_pixel=1)     # ADDS to the root accumulator
    np.testing.assert_array_equal(acc2, 2.0 * ref)
    dyn, _ = gs.render_multi(cam, 2, 10, comm)
    np.testing.assert_allclose(dyn, ref, rtol=1e-12, atol=1e-12)
    gs.close(); comm.close()
    assert pt.shard_range(4000, 3, 8) == (1500, 2000)


def test_accum_on_device_and_caller_stream(pt, ctx):
    """pt_render_opts.accum_on_device + .stream (pt_amd.h): the kernels add into a caller-owned DEVICE buffer on a
    caller-owned stream. Checked against the host-accumulator path, bit for bit in static mode; the device buffer is
    pre-filled so that 'adds to' is visible."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    gs = pt.Scene(ctx)
    cam = gs.build_scene(6, 64, 6)
    ref, st0 = gs.render(cam, 9, 0, 6, slots_per_pixel=1)
    base = np.full(ref.shape, 0.25)
    dptr, stream = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), ref.nbytes) == 0 and hip.hipStreamCreate(C.byref(stream)) == 0
    assert hip.hipMemcpy(dptr, base.ctypes.data, ref.nbytes, 1) == 0          # hipMemcpyHostToDevice
    _, st = gs.render(cam, 9, 0, 6, slots_per_pixel=1, device_ptr=dptr.value, stream=stream.value)
    assert hip.hipStreamSynchronize(stream) == 0
    out = np.empty_like(ref)
    assert hip.hipMemcpy(out.ctypes.data, dptr, ref.nbytes, 2) == 0           # hipMemcpyDeviceToHost
    np.testing.assert_array_equal(out, base + ref)
    assert st.segments == st0.segments
    hip.hipFree(dptr); hip.hipStreamDestroy(stream)
    gs.close()


@pytest.mark.parametrize("k", [1, 0])
def test_eight_rank_shards_into_one_device_accumulator_equal_the_full_render(pt, ctx, k):
    """What pt_render_multi computes on 8 GPUs, rehearsed on ONE (RCCL refuses two ranks on one device): the eight sample
    ranges pt_shard_range(spp, r, 8) rendered one after the other through pt_render(accum_on_device) into ONE device buffer
    — the in-place ncclReduce(sum) of the per-rank buffers does exactly these additions — against the single full render.
    k = 1 (static, the reference's per-pixel sample order; each shard's sums are exact partial sums in sample order, so the
    total differs from the full render only by the association of the additions): <= 1e-13 relative; k = 0 (dynamic,
    the default: atomic adds in any order): the same bound. Sample and segment counts add up exactly."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    gs = pt.Scene(ctx)
    spp = 37                                   # not a multiple of 8: ranks 0..4 take 5 samples, 5..7 take 4
    cam = gs.build_scene(6, 96, spp)
    full, st_full = gs.render(cam, 3, 0, spp, slots_per_pixel=k)
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), full.nbytes) == 0 and hip.hipMemset(dptr, 0, full.nbytes) == 0
    seg = smp = 0
    covered = []
    for r in range(8):
        lo, hi = pt.shard_range(spp, r, 8)
        covered += list(range(lo, hi))
        _, st = gs.render(cam, 3, lo, hi, slots_per_pixel=k, device_ptr=dptr.value)
        seg += st.segments
        smp += st.samples
    assert covered == list(range(spp))
    out = np.empty_like(full)
    assert hip.hipMemcpy(out.ctypes.data, dptr, full.nbytes, 2) == 0           # hipMemcpyDeviceToHost
    hip.hipFree(dptr)
    assert seg == st_full.segments and smp == st_full.samples == 96 * 54 * spp
    np.testing.assert_allclose(out, full, rtol=1e-13, atol=1e-13)
    gs.close()


def test_end_of_frame_pool_compaction_changes_no_result(pt, ctx):
    """The thinning pool is compacted at the frame's end (k_compact_scan / k_compact_move): live slots move to the front and the
    launches shrink. A frame whose whole budget fits the pool at once (one sample per slot: the render is nothing BUT its end) must
    compact (at least once: the host acts on the live count of its last poll) and give the same sums as the static one-slot-per-pixel render, up to the order of the f64 additions;
    counts are exact. Cornell box (lights list, deep paths: the longest tail) and scene 6 (meshes, two-phase K2)."""
    for sid, width, spp in ((3, 160, 24), (6, 192, 20)):
        gs = pt.Scene(ctx)
        cam = gs.build_scene(sid, width, spp)
        ref, st_ref = gs.render(cam, 5, 0, spp, slots_per_pixel=1)
        dyn, st = gs.render(cam, 5, 0, spp)
        assert st.compactions >= 1 and st.n_alloc_end < st.n_slots // 8, (st.compactions, st.n_alloc_end, st.n_slots)
        assert st.samples == st_ref.samples and st.segments == st_ref.segments
        fin = np.isfinite(ref)
        np.testing.assert_allclose(dyn[fin], ref[fin], rtol=1e-11, atol=1e-11)
        off, st_off = _with_env({"PT_EXPERIMENT": "1", "PT_NO_COMPACT_POOL": "1"}, lambda: gs.render(cam, 5, 0, spp))
        assert st_off.compactions == 0 and st_off.segments == st.segments
        np.testing.assert_allclose(off[fin], ref[fin], rtol=1e-11, atol=1e-11)
        gs.close()


def test_two_gpu_render_multi_equals_single_gpu(pt, ctx, tmp_path):
    """pt_render_multi over TWO ranks (RCCL ncclReduce over xGMI) against the single-GPU frame — runs only where two GPUs are
    visible (the 1-GPU boxes of this pool skip it; RCCL refuses two ranks on one device). `bench.py --gpus 2` starts its own two
    ranks; rank 0's reduced frame must equal the one-rank frame up to the order of the f64 additions, and the JSON must say
    n_gpus = 2 (the communicator's size)."""
    import ctypes as C, json, subprocess, sys
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    assert hip.hipGetDeviceCount(C.byref(n)) == 0
    if n.value < 2:
        pytest.skip(f"needs two GPUs ({n.value} visible): the world > 1 path of pt_render_multi stays unverified on hardware")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    frames = {}
    for gpus in (1, 2):
        out = str(tmp_path / f"f{gpus}.npy")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--width", "256", "--spp", "16", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline", "--dump-frame", out], capture_output=True, text=True, timeout=600,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
        assert r.returncode == 0, r.stderr[-3000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == gpus and line["frame_check"]["finite"]
        frames[gpus] = np.load(out)
    np.testing.assert_allclose(frames[2], frames[1], rtol=1e-12, atol=1e-12)


def test_scene5_3840x2160_properties(pt, det, ctx, scene_images):
    """Config 5's frame size (scene 5, 3840x2160, the 87 MB environment atlas, a 199 MB accumulator) with 2 spp: shape,
    finiteness, counters, additivity of sample ranges, and ~100 random (pixel, sample)s bit-equal to the oracle's trace."""
    gs = pt.Scene(ctx)
    cam = gs.build_scene(5, 3840, 8000)
    a, sa = gs.render(cam, 1, 0, 1)
    assert a.shape == (2160, 3840, 3) and sa.samples == 3840 * 2160 and 1.2 < sa.segments / sa.samples < 2.2
    assert np.isfinite(a).all() and a.min() >= 0
    b, sb = gs.render(cam, 1, 1, 2, slots_per_pixel=1)
    c, sc = gs.render(cam, 1, 0, 2)
    np.testing.assert_allclose(a + b, c, rtol=1e-13, atol=1e-13)
    assert sa.segments + sb.segments == sc.segments
    os_ = det.Scene()
    ocam = os_.build_scene(5, 3840, 8000, images=scene_images(5))
    rng = np.random.default_rng(5)
    flat = b.reshape(-1, 3)
    for pix in rng.integers(0, 3840 * 2160, 100):
        rad, _, _ = os_.trace_sample(ocam, 1, int(pix), 1)
        np.testing.assert_array_equal(flat[pix], rad)
    gs.close(); os_.close()


def test_shared_and_nested_instances_bit_exact(pt, det, ctx):
    """One mesh under three instances (shared tree and triangles, ids per placement), an instance of an instance of it, a
    twice-wrapped cuboid and a twice-wrapped quad light (lights.sample / lights.pdf through an instance chain): every
    accumulator value identical to the oracle's Hittable composition (instance.rs:20-75)."""
    from common import shared_and_nested_instances_scene
    spec = shared_and_nested_instances_scene(True)
    gs, os_ = pt.Scene(ctx), det.Scene()
    gres, ores = spec.replay(gs), spec.replay(os_)
    assert gs.prim_count() == os_.prim_count() == 1 + 4 * 80 + 6 + 1
    gcam, ocam = spec.make_camera(pt.Camera, gres), spec.make_camera(det.Camera, ores)
    ga, st = gs.render(gcam, 5, 0, 8, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 5, 0, 8)
    assert st.segments == cnt["segments"] and st.extend_variant == 0
    np.testing.assert_array_equal(ga, oa)
    rng = np.random.default_rng(8)
    rays = np.zeros((4000, 7))
    rays[:, 0:3] = rng.uniform(-3, 3, (4000, 3)) + (0.0, 3.0, 0.0)
    rays[:, 3:6] = rng.normal(size=(4000, 3))
    g = gs.intersect(rays)
    np.testing.assert_array_equal(g, os_.intersect(rays))
    assert len(np.unique(g[g[:, 0] > 0, 2])) > 100                  # hits spread over the placements' id ranges
    gd, _ = _with_env({"PT_EXPERIMENT": "1", "PT_K2": "batch"}, lambda: gs.render(gcam, 5, 0, 8, slots_per_pixel=1))
    np.testing.assert_array_equal(gd, oa)                          # the batch form of K2 walks the same chains
    gs.close(); os_.close()


def test_objects_placed_directly_and_under_instances_bit_exact(pt, det, ctx):
    """world.rs:18-24 / instance.rs:20-30 take any Arc<dyn Hittable>: one sphere added to the world twice and under an instance,
    one cuboid and one mesh placed directly and under (nested) instances, a quad that is a light AND, instanced, an object —
    every accumulator value and the segment count equal the oracle's, through both forms of K2; coincident placements are exact
    t ties decided by the per-placement ids."""
    from common import free_placement_scene
    spec = free_placement_scene()
    gs, os_ = pt.Scene(ctx), det.Scene()
    gcam = spec.make_camera(pt.Camera, spec.replay(gs))
    ocam = spec.make_camera(det.Camera, spec.replay(os_))
    assert gs.prim_count() == os_.prim_count() == 258
    ga, gst = gs.render(gcam, 7, 0, 6, slots_per_pixel=1)
    oa, cnt = os_.render(ocam, 7, 0, 6)
    np.testing.assert_array_equal(ga, oa)
    assert gst.segments == cnt["segments"]
    gd, _ = _with_env({"PT_EXPERIMENT": "1", "PT_K2": "batch"}, lambda: gs.render(gcam, 7, 0, 6, slots_per_pixel=1))
    np.testing.assert_array_equal(gd, oa)                          # the batch form of K2
    gs.close(); os_.close()


def test_float_hdr_environment_and_textures_bit_exact(pt, det, ctx, scene_images):
    """The float-HDR option (pt_scene_set_float_hdr / pt_tex_image_rgbf32): scenes 4 and 6 with grace_probe_latlong.hdr kept as
    f32 samples — no .to_rgb8() squash (texture.rs:67), same lookup (texture.rs:73-91) — and a float image used as a quad's colour
    texture: every accumulator value equals the oracle's twin; the frames are brighter than the RGB8 ones (the probe's lights
    reach 1088 where RGB8 stops at 1)."""
    for sid, width, spp in ((4, 96, 6), (6, 112, 6)):
        gs, os_ = pt.Scene(ctx), det.Scene()
        gs.set_float_hdr(True); os_.set_float_hdr(True)
        gcam = gs.build_scene(sid, width, spp)
        ocam = os_.build_scene(sid, width, spp, images=scene_images(sid))
        ga, st = gs.render(gcam, 1, 0, spp, slots_per_pixel=1)
        oa, cnt = os_.render(ocam, 1, 0, spp)
        np.testing.assert_array_equal(ga, oa)
        assert st.segments == cnt["segments"]
        g8 = pt.Scene(ctx)
        a8, _ = g8.render(g8.build_scene(sid, width, spp), 1, 0, spp, slots_per_pixel=1)
        assert ga.sum() > 1.5 * a8.sum() and ga.max() > 100 * spp * 0.01
        gs.close(); os_.close(); g8.close()
    rng = np.random.default_rng(5)
    img = (rng.random((8, 16, 3)) * 6.0).astype(np.float32)
    spec = SceneSpec()
    t = spec.add("tex_image_rgbf32", img)
    spec.add("world_add_object", spec.add("quad", (-2.0, 0.0, -2.0), (0.0, 0.0, 4.0), (4.0, 0.0, 0.0), spec.add("mat_diffuse", t, -1)))
    spec.add("world_add_object", spec.add("sphere", 0.6, (0.0, 0.6, 0.0), (0.0, 0.6, 0.0), spec.add("mat_metal", t, spec.add("tex_solid_f", 0.3))))
    spec.add("world_build")
    spec.camera = default_camera(width=48, look_from=(0.0, 2.5, -4.0), look_at=(0.0, 0.3, 0.0), vfov=50.0, env_color=(0.6, 0.7, 0.9))
    gs, os_ = pt.Scene(ctx), det.Scene()
    gcam, ocam = spec.make_camera(pt.Camera, spec.replay(gs)), spec.make_camera(det.Camera, spec.replay(os_))
    ga, _ = gs.render(gcam, 2, 0, 8, slots_per_pixel=1)
    oa, _ = os_.render(ocam, 2, 0, 8)
    np.testing.assert_array_equal(ga, oa)
    gs.close(); os_.close()


def test_device_bvh_builder_bit_exact(pt, det, ctx):
    """pt_world_set_device_bvh_threshold: mesh BVHs built by the GPU LBVH builder (csrc/pt_bvh_device.hip) instead of the host's
    binned SAH. The closest hit is tree-independent (minimum t, ties -> larger id), so hits, segment counts and every
    accumulator value must equal the oracle's (its own tree is the reference's sweep SAH) and the host-built scene's."""
    def build(scene, device):
        spec = SceneSpec()
        rgb = lambda r, g, b: spec.add("tex_solid_rgb", r, g, b)
        spec.add("world_add_object", spec.add("quad", (-8.0, 0.0, -8.0), (0.0, 0.0, 16.0), (16.0, 0.0, 0.0),
                                              spec.add("mat_diffuse", spec.add("tex_checker", 0.9, rgb(0.2, 0.2, 0.3), rgb(0.9, 0.9, 0.8)), -1)))
        mats = [spec.add("mat_metal", rgb(0.9, 0.7, 0.5), spec.add("tex_solid_f", 0.1)), spec.add("mat_glass", rgb(1, 1, 1), spec.add("tex_solid_f", 0.02), 0.0, 1.5)]
        for k, (sub, sc, tr) in enumerate(((4, 0.9, (-1.2, 0.95, 0.0)), (5, 0.8, (1.1, 0.85, 0.6)))):
            P, I = icosphere(sub)
            P = (P * (1.0 + 0.15 * np.sin(7.0 * P[:, [0]]) * np.cos(5.0 * P[:, [1]]))).astype(np.float32)      # a lumpy ball
            m = spec.add("mesh", sc, P, I, None, None, mats[k])
            spec.add("world_add_object", spec.add("instance", m, (0.3, 1.0, 0.2), 0.4 + k, tr))
        spec.add("world_add_light", spec.add("quad", (-1.0, 4.0, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0), spec.add("mat_light", rgb(8, 8, 7))))
        spec.camera = default_camera(width=72, look_from=(0.0, 1.6, -5.0), look_at=(0.0, 0.8, 0.0), vfov=42.0, env_color=(0.1, 0.12, 0.2))
        if device is not None:
            scene.set_device_bvh_threshold(1000 if device else 0)
        res = spec.replay(scene)
        scene.world_build()
        return spec, res
    gd, gh, os_ = pt.Scene(ctx), pt.Scene(ctx), det.Scene()
    spec, rd = build(gd, True)
    _, rh = build(gh, False)
    _, ro = build(os_, None)
    n_dev, depth = gd.device_bvh_info()
    assert n_dev == 2 and 8 <= depth <= 20, (n_dev, depth)
    assert gh.device_bvh_info()[0] == 0
    cam_d, cam_h, cam_o = spec.make_camera(pt.Camera, rd), spec.make_camera(pt.Camera, rh), spec.make_camera(det.Camera, ro)
    ad, sd = gd.render(cam_d, 4, 0, 6, slots_per_pixel=1)
    ah, sh = gh.render(cam_h, 4, 0, 6, slots_per_pixel=1)
    ao, cnt = os_.render(cam_o, 4, 0, 6)
    assert sd.segments == sh.segments == cnt["segments"]
    np.testing.assert_array_equal(ad, ao)
    np.testing.assert_array_equal(ah, ao)
    rng = np.random.default_rng(12)
    rays = np.zeros((5000, 7))
    rays[:, 0:3] = rng.uniform(-2.5, 2.5, (5000, 3)) + (0.0, 2.0, 0.0)
    rays[:, 3:6] = rng.normal(size=(5000, 3))
    g = gd.intersect(rays)
    np.testing.assert_array_equal(g, os_.intersect(rays))
    assert (g[:, 0] > 0).mean() > 0.4
    gd.close(); gh.close(); os_.close()


def _brute_force_closest(P, I, rays, t_min=1e-3):
    """Triangle::intersects (mesh.rs:50-82) over EVERY triangle, in numpy with the kernels' operation order (IEEE f64, nothing
    fused): closest t, ties -> larger face index. Returns (hit, t, face) per ray."""
    V = P.astype(np.float64)[I.reshape(-1, 3)]
    v0 = V[:, 0]
    e1, e2 = V[:, 1] - v0, V[:, 2] - v0
    dot = lambda a, b: (a[..., 0] * b[..., 0]) + (a[..., 1] * b[..., 1]) + (a[..., 2] * b[..., 2])
    cross = lambda a, b: np.stack([a[..., 1] * b[..., 2] - b[..., 1] * a[..., 2], a[..., 2] * b[..., 0] - b[..., 2] * a[..., 0],
                                   a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]], axis=-1)
    out = []
    for r in rays:
        o, d = r[0:3], r[3:6]
        d = d * (1.0 / np.sqrt(dot(d, d)))                               # Ray::new normalises (ray.rs:23-29): v * length_recip
        h = cross(d[None, :], e2)
        a = dot(e1, h)
        with np.errstate(divide="ignore", invalid="ignore"):
            f = 1.0 / a
            sv = o[None, :] - v0
            u = f * dot(sv, h)
            q = cross(sv, e1)
            v = f * dot(d[None, :], q)
            t = f * dot(e2, q)
        ok = (np.abs(a) >= 1e-8) & (u >= 0.0) & (u <= 1.0) & (v >= 0.0) & (u + v <= 1.0) & (t >= t_min) & (t <= np.inf)
        if not ok.any():
            out.append((False, 0.0, -1))
            continue
        tt = np.where(ok, t, np.inf)
        tmin = tt.min()
        out.append((True, float(tmin), int(np.flatnonzero(tt == tmin).max())))
    return out


def test_device_bvh_builder_at_a_million_triangles(pt, ctx):
    """The GPU BVH builder at the size it exists for: a lumpy icosphere(8) — 1,310,720 triangles — built by the
    depth-bounded LBVH of csrc/pt_bvh_device.hip. The tree must fit the traversal stacks (leaf depth <= 29: no fall-back to
    the host builder, which round 2's unbounded radix tree needed at this size), hits must equal a numpy brute force over all
    triangles bit for bit (t, face, closest with ties -> larger id; the oracle's O(n^2) SAH build cannot run at this size), and a
    small render must equal, value for value, the render of the same mesh under the host's binned-SAH tree."""
    import time
    P, I = icosphere(8)
    P = (P * (1.0 + 0.12 * np.sin(9.0 * P[:, [0]]) * np.cos(7.0 * P[:, [1]]) + 0.05 * np.sin(31.0 * P[:, [2]]))).astype(np.float32)
    assert len(I) // 3 == 1310720

    def build(device):
        s = pt.Scene(ctx)
        s.set_device_bvh_threshold(1 << 19 if device else 0)
        m = s.mat_metal(s.tex_solid_rgb(0.9, 0.8, 0.6), s.tex_solid_f(0.2))
        s.world_add_object(s.mesh(1.0, P, I, None, None, m))
        s.world_add_object(s.quad((-6.0, -1.4, -6.0), (0.0, 0.0, 12.0), (12.0, 0.0, 0.0), s.mat_diffuse(s.tex_solid_rgb(0.7, 0.7, 0.7), -1)))
        t = time.time()
        s.world_build()
        return s, time.time() - t
    gd, t_dev = build(True)
    n_dev, depth = gd.device_bvh_info()
    assert n_dev == 1 and 18 <= depth <= 29, (n_dev, depth)
    rng = np.random.default_rng(21)
    rays = np.zeros((240, 7))
    rays[:, 0:3] = rng.normal(size=(240, 3)) * 0.3 + rng.choice([-3.0, 3.0], size=(240, 1)) * rng.normal(size=(240, 3)) * 0.5
    rays[:120, 3:6] = -rays[:120, 0:3] + rng.normal(size=(120, 3)) * 0.4       # aimed at the ball
    rays[120:, 3:6] = rng.normal(size=(120, 3))                                 # anywhere (some start inside)
    g = gd.intersect(rays)
    want = _brute_force_closest(P, I, rays)
    n_mesh_hits = 0
    for k, (hit, t, face) in enumerate(want):
        if g[k, 0] > 0 and g[k, 2] >= len(I) // 3:                              # the floor quad (id after the mesh's faces) was closer
            assert (not hit) or g[k, 1] <= t
            continue
        assert bool(g[k, 0] > 0) == hit, k
        if hit:
            assert g[k, 1] == t and int(g[k, 2]) == face, (k, g[k, 1], t, g[k, 2], face)
            n_mesh_hits += 1
    assert n_mesh_hits > 80
    cam = default_camera(width=64, look_from=(0.0, 1.2, -3.6), look_at=(0.0, 0.0, 0.0), vfov=40.0, env_color=(0.5, 0.6, 0.8))
    spec = SceneSpec(); spec.camera = cam
    a_dev, st_dev = gd.render(spec.make_camera(pt.Camera, {}), 2, 0, 3, slots_per_pixel=1)
    gd.close()
    gh, t_host = build(False)
    assert gh.device_bvh_info()[0] == 0
    a_host, st_host = gh.render(spec.make_camera(pt.Camera, {}), 2, 0, 3, slots_per_pixel=1)
    gh.close()
    np.testing.assert_array_equal(a_dev, a_host)
    assert st_dev.segments == st_host.segments and st_dev.extend_variant == 0      # the two-phase K2 (its 28/32-entry stacks cover the tree)
    print(f"[lbvh] 1,310,720 triangles: device build {t_dev:.2f} s (depth {depth}), host binned SAH {t_host:.2f} s")
