// The reference's seven scene scripts (src/main.rs:14-618) written against the C++ mirror of
// its API (pt.hpp). Each function returns the World + Camera the script sets up; the caller
// (main.cpp's `-s N`, or pt_build_scene for Python/tests) then builds and renders.
// Literals are the reference's; `scene_seed` replaces the unseeded thread_rng() that
// balls_scene uses at build time (main.rs:38-47), drawn in the same order.
#include "scenes.hpp"

#include <cmath>

using namespace path_tracer;

namespace {
// Seedable stand-in for rand's thread_rng at scene-BUILD time only (balls_scene): the same
// Philox4x32-10 stream the renderer uses, keyed (scene_seed, 0xBA115), sample 0.
struct BuildRng {
    uint64_t seed;
    uint32_t draw = 0;
    double gen() {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
        uint32_t c0 = draw >> 1, c1 = 0u, c2 = (uint32_t)(seed >> 32), c3 = 0u, k0 = (uint32_t)seed, k1 = 0xBA115u;
        for (int r = 0; r < 10; ++r) {
            uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += W0; k1 += W1;
        }
        uint64_t v = (draw & 1u) ? (((uint64_t)c3 << 32) | c2) : (((uint64_t)c1 << 32) | c0);
        ++draw;
        return (double)(v >> 11) * (1.0 / 9007199254740992.0);
    }
};
Camera base_camera(size_t width, size_t spp) {
    Camera camera = Camera::new_();
    camera.image_width = width;
    camera.samples_per_pixel = spp;
    camera.max_depth = 50;
    camera.vup = Vec3::new_(0.0, 1.0, 0.0);
    camera.blur_strength = 0.5;
    return camera;
}
}  // namespace

SceneSetup balls_scene(size_t width, size_t spp, uint64_t scene_seed) {   // main.rs:14-82
    World world = World::new_();
    auto checker = CheckerTexture<Vec3>::new_(0.32, SolidTexture<Vec3>::new_(Vec3::new_(0.2, 0.3, 0.1)),
                                              SolidTexture<Vec3>::new_(Vec3::new_(0.9, 0.9, 0.9)));
    world.add_object(Sphere::new_still(1000.0, Vec3::new_(0.0, -1000.0, 0.0), DiffuseBRDF::new_(checker)));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(0.0, 1.0, 0.0), GlassBSDF::basic(1.5)));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(-4.0, 1.0, 0.0), DiffuseBRDF::from_rgb(Vec3::new_(0.4, 0.2, 0.1))));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(4.0, 1.0, 0.0), MetalBRDF::from_rgb(Vec3::new_(0.7, 0.6, 0.5), 0.0)));
    BuildRng rng{scene_seed};
    for (int ai = -11; ai < 11; ++ai)
        for (int bi = -11; bi < 11; ++bi) {
            double a = ai, b = bi;
            double choose_mat = rng.gen();
            double cx = a + 0.9 * rng.gen();
            double cz = b + 0.9 * rng.gen();
            Vec3 center = Vec3::new_(cx, 0.2, cz);
            Vec3 d = center - Vec3::new_(4.0, 0.2, 0.0);
            if (std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z) > 0.9) {
                if (choose_mat < 0.8) {
                    double r1 = rng.gen(), g1 = rng.gen(), b1 = rng.gen(), r2 = rng.gen(), g2 = rng.gen(), b2 = rng.gen();
                    auto m = DiffuseBRDF::from_rgb(Vec3::new_(r1 * r2, g1 * g2, b1 * b2));
                    Vec3 pos2 = center + Vec3::new_(0.0, 0.5 * rng.gen(), 0.0);
                    world.add_object(Sphere::new_moving(0.2, center, pos2, m));
                } else if (choose_mat < 0.95) {
                    double r = 0.5 + 0.5 * rng.gen(), g = 0.5 + 0.5 * rng.gen(), bb = 0.5 + 0.5 * rng.gen();
                    world.add_object(Sphere::new_still(0.2, center, MetalBRDF::from_rgb(Vec3::new_(r, g, bb), 0.0)));
                } else {
                    world.add_object(Sphere::new_still(0.2, center, GlassBSDF::basic(1.5)));
                }
            }
        }
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 16.0 / 9.0;
    camera.vfov = 20.0;
    camera.look_from = Vec3::new_(13.0, 2.0, 3.0);
    camera.look_at = Vec3::ZERO;
    camera.focal_length = 10.0;
    camera.defocus_angle = 0.6;
    camera.environment = EnvironmentType::Color(Vec3::new_(0.7, 0.8, 1.0));
    return {std::move(world), camera, "demo/balls.png"};
}

SceneSetup earth_scene(size_t width, size_t spp) {   // main.rs:84-132
    World world = World::new_();
    world.add_object(Sphere::new_still(1.0, Vec3::new_(4.9, 1.0, 3.0), DiffuseBRDF::new_(ImageTexture::new_("assets/earthmap.jpg"))));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(0.0, 1.0, 0.0), DiffuseBRDF::from_rgb(Vec3::new_(0.4, 0.2, 0.1))));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(4.0, 1.0, 0.0), MetalBRDF::from_rgb(Vec3::new_(0.7, 0.6, 0.5), 0.1)));
    auto checker = CheckerTexture<Vec3>::new_(0.62, SolidTexture<Vec3>::new_(Vec3::new_(0.9, 0.0, 0.1)),
                                              SolidTexture<Vec3>::new_(Vec3::new_(0.9, 0.9, 0.9)));
    world.add_object(Sphere::new_still(1000.0, Vec3::new_(0.0, -1000.0, 0.0), DiffuseBRDF::new_(checker)));
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 16.0 / 9.0;
    camera.vfov = 28.0;
    camera.look_from = Vec3::new_(8.8, 2.0, 3.0);
    camera.look_at = Vec3::ZERO;
    camera.focal_length = 2.869817807;
    camera.defocus_angle = 2.5;
    camera.environment = EnvironmentType::Color(Vec3::new_(0.85, 0.85, 1.0));
    return {std::move(world), camera, "demo/earth.png"};
}

static void cornell_room(World& world, MatPtr left, MatPtr right, MatPtr white) {   // main.rs:140-169
    world.add_object(Quad::new_(Vec3::new_(555.0, 0.0, 0.0), Vec3::new_(0.0, 555.0, 0.0), Vec3::new_(0.0, 0.0, 555.0), left));
    world.add_object(Quad::new_(Vec3::new_(0.0, 0.0, 0.0), Vec3::new_(0.0, 555.0, 0.0), Vec3::new_(0.0, 0.0, 555.0), right));
    world.add_object(Quad::new_(Vec3::new_(0.0, 0.0, 0.0), Vec3::new_(555.0, 0.0, 0.0), Vec3::new_(0.0, 0.0, 555.0), white));
    world.add_object(Quad::new_(Vec3::new_(555.0, 555.0, 555.0), Vec3::new_(-555.0, 0.0, 0.0), Vec3::new_(0.0, 0.0, -555.0), white));
    world.add_object(Quad::new_(Vec3::new_(0.0, 0.0, 555.0), Vec3::new_(555.0, 0.0, 0.0), Vec3::new_(0.0, 555.0, 0.0), white));
}
static Camera cornell_camera(size_t width, size_t spp) {   // main.rs:217-232
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 1.0;
    camera.vfov = 40.0;
    camera.look_from = Vec3::new_(278.0, 278.0, -800.0);
    camera.look_at = Vec3::new_(278.0, 278.0, 0.0);
    camera.focal_length = 10.0;
    camera.defocus_angle = 0.0;
    camera.environment = EnvironmentType::Color(Vec3::ZERO);
    return camera;
}

SceneSetup cornell_box_scene(size_t width, size_t spp) {   // main.rs:134-236
    World world = World::new_();
    auto red = DiffuseBRDF::from_rgb(Vec3::new_(0.65, 0.05, 0.05));
    auto white = DiffuseBRDF::from_rgb(Vec3::new_(0.73, 0.73, 0.73));
    auto green = DiffuseBRDF::from_rgb(Vec3::new_(0.12, 0.45, 0.15));
    cornell_room(world, green, red, white);
    world.add_light(Quad::new_(Vec3::new_(343.0, 554.0, 332.0), Vec3::new_(-130.0, 0.0, 0.0), Vec3::new_(0.0, 0.0, -105.0),
                               DiffuseLight::from_rgb(Vec3::new_(25.0, 25.0, 25.0))));
    auto mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::ONE),
                                    0.01,   // metallic
                                    0.01,   // roughness
                                    0.01,   // subsurface
                                    0.91,   // specular
                                    0.91,   // specular_tint
                                    1.5,    // ior
                                    0.91,   // spec_trans
                                    0.91,   // sheen
                                    0.91,   // sheen_tint
                                    0.91,   // clearcoat
                                    0.01);  // clearcoat_gloss
    world.add_object(Sphere::new_still(135.0, Vec3::new_(113.0, 170.0, 372.0), mat));
    auto box1 = Cuboid::new_(Vec3::ZERO, Vec3::new_(165.0, 330.0, 165.0), MetalBRDF::from_rgb(Vec3::ONE, 0.1));
    world.add_object(Instance::new_(box1, Vec3::Y, 0.261799, Vec3::new_(265.0, 0.0, 295.0)));
    auto box2 = Cuboid::new_(Vec3::ZERO, Vec3::new_(165.0, 165.0, 165.0), white);
    world.add_object(Instance::new_(box2, Vec3::Y, -0.29, Vec3::new_(130.0, 0.0, 65.0)));
    return {std::move(world), cornell_camera(width, spp), "demo/cornell.png"};
}

SceneSetup environment_map_scene(size_t width, size_t spp) {   // main.rs:238-274
    World world = World::new_();
    world.add_object(Sphere::new_still(9.0, Vec3::new_(4.0, 2.0, 0.0), MetalBRDF::from_rgb(Vec3::ONE, 0.001)));
    world.add_object(Quad::new_(Vec3::new_(-2.0, 6.5, 0.0), Vec3::new_(4.0, 0.0, 0.0), Vec3::new_(0.0, 0.0, 2.0),
                                DiffuseLight::from_rgb(Vec3::new_(10.0, 10.0, 10.0))));
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 16.0 / 9.0;
    camera.vfov = 90.0;
    camera.look_from = Vec3::new_(0.0, 3.0, 17.0);
    camera.look_at = Vec3::new_(0.0, 2.0, 0.0);
    camera.focal_length = 17.0;
    camera.defocus_angle = 1.5;
    camera.environment = EnvironmentType::Map(ImageTexture::new_("assets/grace_probe_latlong.hdr"));
    return {std::move(world), camera, "demo/lights.png"};
}

SceneSetup bsdf_demo_scene(size_t width, size_t spp) {   // main.rs:276-369
    World world = World::new_();
    for (int i = 0; i < 5; ++i) {   // dielectric, varying roughness
        double roughness = 0.1 + 0.2 * (double)i;
        auto mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.65, 0.05, 0.05)), 0.00, roughness, 0.01, 0.01, 0.01, 1.5, 0.01, 0.01, 0.01, 0.01, 0.01);
        world.add_object(Sphere::new_still(0.5, Vec3::new_(-4.0 + (double)i, 1.0, -5.0), mat));
    }
    for (int i = 0; i < 5; ++i) {   // metal
        double roughness = 0.1 + 0.2 * (double)i;
        auto mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.05, 0.65, 0.05)), 0.99, roughness, 0.01, 0.01, 0.01, 1.5, 0.01, 0.01, 0.01, 0.01, 0.01);
        world.add_object(Sphere::new_still(0.5, Vec3::new_(-4.0 + (double)i, 2.0, -5.0), mat));
    }
    for (int i = 0; i < 5; ++i) {   // glass
        double roughness = (0.1 + 0.2 * (double)i) * 0.3;
        auto mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.25, 0.05, 0.65)), 0.01, roughness, 0.01, 0.01, 0.01, 1.5, 0.99, 0.01, 0.01, 0.01, 0.01);
        world.add_object(Sphere::new_still(0.5, Vec3::new_(-4.0 + (double)i, 3.0, -5.0), mat));
    }
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 16.0 / 9.0;
    camera.vfov = 60.0;
    camera.look_from = Vec3::new_(-2.0, 2.0, -1.0);
    camera.look_at = camera.look_from + Vec3::new_(0.0, 0.0, -1000.0);
    camera.vup = Vec3::Y;
    camera.focal_length = 5.0;
    camera.defocus_angle = 0.0;
    camera.environment = EnvironmentType::Map(ImageTexture::new_("assets/envmap.jpg"));
    return {std::move(world), camera, "demo/bsdf.png"};
}

SceneSetup everything_scene(size_t width, size_t spp, const std::string& asset_dir) {   // main.rs:371-532
    World world = World::new_();
    auto checker = CheckerTexture<Vec3>::new_(0.92, SolidTexture<Vec3>::new_(Vec3::new_(0.2, 0.3, 0.1)),
                                              SolidTexture<Vec3>::new_(Vec3::new_(0.9, 0.9, 0.9)));
    world.add_object(Quad::new_(Vec3::new_(-1000.0, 0.0, -1000.0), Vec3::new_(0.0, 0.0, 5000.0), Vec3::new_(5000.0, 0.0, 0.0),
                                DiffuseBRDF::from_textures(checker, nullptr)));
    world.add_object(Sphere::new_still(2.0, Vec3::new_(-4.0, 2.0, 9.8), MetalBRDF::from_rgb(Vec3::ONE, 0.001)));
    world.add_object(Sphere::new_still(1.0, Vec3::new_(4.0, 1.0, 6.0), GlassBSDF::basic(1.5)));
    auto box1 = Cuboid::new_(Vec3::ZERO, Vec3::new_(1.0, 2.0, 1.0), DiffuseBRDF::from_rgb(Vec3::new_(0.0, 0.5, 1.0)));
    world.add_object(Instance::new_(box1, Vec3::Y, 0.5, Vec3::new_(1.2, 0.0, 6.0)));

    auto bunny_models = tobj::load_obj(asset_dir + "/bunny.obj");
    auto bunny_material = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::ONE), 0.91, 0.01, 0.01, 0.01, 0.91, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
    world.add_object(Instance::new_(TriangleMesh::from_obj(10.0, bunny_models[0].mesh, bunny_material), Vec3::Y, 3.14, Vec3::new_(0.1, -0.327, 5.0)));

    auto spot_models = tobj::load_obj(asset_dir + "/spot.obj");
    auto spot_mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.65, 0.05, 0.05)), 0.01, 0.01, 0.91, 0.01, 0.01, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
    world.add_object(Instance::new_(TriangleMesh::from_obj(0.65, spot_models[0].mesh, spot_mat), Vec3::Y, 0.87, Vec3::new_(-1.5, 2.8, 4.3)));

    auto cow_models = tobj::load_obj(asset_dir + "/cow.obj");
    auto cow_mat = PrincipledBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.05, 0.65, 0.05)), 0.91, 0.21, 0.91, 0.01, 0.01, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
    world.add_object(Instance::new_(TriangleMesh::from_obj(0.75, cow_models[0].mesh, cow_mat), Vec3::Y, 0.93, Vec3::new_(2.5, 3.8, 12.0)));

    world.add_object(Sphere::new_still(0.1, Vec3::new_(1.0, 0.1, 3.0), DiffuseLight::from_rgb(Vec3::new_(20.0, 20.0, 10.0))));
    world.add_object(Sphere::new_still(0.2, Vec3::new_(0.0, 0.2, 3.0), MetalBRDF::from_rgb(Vec3::new_(0.6, 0.05, 0.05), 0.1)));
    world.add_object(Sphere::new_still(0.3, Vec3::new_(1.2, 0.3, 3.4),
                                       GlassBSDF::new_(SolidTexture<Vec3>::new_(Vec3::new_(0.7, 0.3, 0.3)), SolidTexture<double>::new_(0.3), 0.0, 1.5)));
    Camera camera = base_camera(width, spp);
    camera.aspect_ratio = 16.0 / 9.0;
    camera.vfov = 60.0;
    camera.look_from = Vec3::new_(0.0, 1.5, 0.0);
    camera.look_at = Vec3::new_(0.0, 1.5, 100000.0);
    camera.vup = Vec3::Y;
    camera.focal_length = 6.0;
    camera.defocus_angle = 1.0;
    camera.environment = EnvironmentType::Map(ImageTexture::new_("assets/grace_probe_latlong.hdr"));
    return {std::move(world), camera, "demo/scene6.png"};
}

SceneSetup normal_demo_scene(size_t width, size_t spp) {   // main.rs:534-618
    World world = World::new_();
    auto bricks_albedo = ImageTexture::new_("assets/bricks/color.png");
    auto bricks_normal = ImageTexture::new_("assets/bricks/normal.png");
    auto material_with_normal = DiffuseBRDF::from_textures(bricks_albedo, bricks_normal);
    auto material_without_normal = DiffuseBRDF::from_textures(bricks_albedo, nullptr);
    auto white = DiffuseBRDF::from_rgb(Vec3::new_(0.73, 0.73, 0.73));
    cornell_room(world, material_without_normal, material_with_normal, white);
    world.add_light(Quad::new_(Vec3::new_(343.0, 554.0, 332.0), Vec3::new_(-130.0, 0.0, 0.0), Vec3::new_(0.0, 0.0, -105.0),
                               DiffuseLight::from_rgb(Vec3::new_(27.0, 28.0, 20.0))));
    auto box1 = Cuboid::new_(Vec3::ZERO, Vec3::new_(165.0, 330.0, 165.0), MetalBRDF::from_rgb(Vec3::splat(0.94), 0.1));
    world.add_object(Instance::new_(box1, Vec3::Y, 0.261799, Vec3::new_(265.0, 0.0, 295.0)));
    world.add_object(Sphere::new_still(100.0, Vec3::new_(130.0, 100.0, 65.0), GlassBSDF::basic(1.5)));
    return {std::move(world), cornell_camera(width, spp), "demo/normals.png"};
}

SceneSetup make_scene(int scene_id, size_t width, size_t spp, const std::string& asset_dir, uint64_t scene_seed) {
    SceneSetup s;
    switch (scene_id) {   // main.rs:635-644
    case 1: s = balls_scene(width, spp, scene_seed); break;
    case 2: s = earth_scene(width, spp); break;
    case 3: s = cornell_box_scene(width, spp); break;
    case 4: s = environment_map_scene(width, spp); break;
    case 5: s = bsdf_demo_scene(width, spp); break;
    case 6: s = everything_scene(width, spp, asset_dir); break;
    case 7: s = normal_demo_scene(width, spp); break;
    default: throw std::runtime_error("unknown scene id (the reference silently does nothing: main.rs:643)");
    }
    s.world.asset_dir = asset_dir;
    return s;
}

// C-ABI convenience: run scene script `scene_id` into an existing pt_scene.
extern "C" int pt_build_scene(pt_scene* scene, int scene_id, uint32_t width, uint32_t spp, const char* asset_dir, uint64_t scene_seed,
                              pt_camera* out_cam) {
    try {
        SceneSetup s = make_scene(scene_id, width, spp, asset_dir ? asset_dir : "assets", scene_seed);
        s.world.emit_into(scene, s.camera.environment.is_map ? s.camera.environment.map : nullptr);
        *out_cam = s.camera.to_c(&s.world);
        return 0;
    } catch (const std::exception& e) {
        return pt_set_error_message(e.what());
    }
}
