// C++ host mirror of the reference's public crate surface (src/lib.rs:1-10) for the render
// path: World / Camera / materials / textures / hittables with the reference's constructor
// names and argument order, so that its scene scripts (src/main.rs) transliterate line by
// line. Everything here is a thin description graph (shared_ptr ~ Arc) that is replayed onto
// the C ABI of include/pt_amd.h when World::build_bvh() runs; the integrator itself
// (Camera::render, camera.rs:79) executes in the HIP kernels behind pt_render().
//
// The reference is Rust; no Rust toolchain exists in this environment, hence C++ here. A Rust
// crate would bind the same C ABI. Rust `T::new(..)` is spelled `T::new_(..)`.
// Error behaviour follows the reference: asset/handle errors are fatal (Rust: unwrap()/panic;
// here: std::runtime_error), a failed image save is only reported (camera.rs:118-123).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"

namespace path_tracer {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    static Vec3 new_(double x, double y, double z) { return Vec3{x, y, z}; }
    static Vec3 splat(double s) { return Vec3{s, s, s}; }
    static const Vec3 ZERO, ONE, X, Y, Z;
    Vec3 operator+(Vec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(Vec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(double s) const { return {x * s, y * s, z * s}; }
};
inline const Vec3 Vec3::ZERO{0, 0, 0}, Vec3::ONE{1, 1, 1}, Vec3::X{1, 0, 0}, Vec3::Y{0, 1, 0}, Vec3::Z{0, 0, 1};

[[noreturn]] inline void panic(const std::string& what) { throw std::runtime_error(what + ": " + pt_last_error()); }

// One replay of the description graph onto a pt_scene; memoises shared nodes (Arc sharing).
struct Emitter {
    pt_scene* scene;
    std::string asset_dir;
    std::map<const void*, int> done;
};

// ---- textures (src/texture.rs) ---------------------------------------------------------
template <class T>
struct Texture {
    virtual ~Texture() = default;
    virtual int emit(Emitter& e) const = 0;
};
template <class T>
using TexPtr = std::shared_ptr<Texture<T>>;

template <class T>
struct SolidTexture;
template <>
struct SolidTexture<Vec3> : Texture<Vec3> {
    Vec3 value;
    static std::shared_ptr<SolidTexture> new_(Vec3 v) { auto t = std::make_shared<SolidTexture>(); t->value = v; return t; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_tex_solid_rgb(e.scene, value.x, value.y, value.z);
        if (h < 0) panic("SolidTexture");
        return e.done[this] = h;
    }
};
template <>
struct SolidTexture<double> : Texture<double> {
    double value;
    static std::shared_ptr<SolidTexture> new_(double v) { auto t = std::make_shared<SolidTexture>(); t->value = v; return t; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_tex_solid_f(e.scene, value);
        if (h < 0) panic("SolidTexture");
        return e.done[this] = h;
    }
};
template <class T>
struct CheckerTexture : Texture<T> {
    double scale;
    TexPtr<T> tex1, tex2;
    static std::shared_ptr<CheckerTexture> new_(double scale, TexPtr<T> a, TexPtr<T> b) {
        auto t = std::make_shared<CheckerTexture>();
        t->scale = scale; t->tex1 = a; t->tex2 = b;
        return t;
    }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_tex_checker(e.scene, scale, tex1->emit(e), tex2->emit(e));
        if (h < 0) panic("CheckerTexture");
        return e.done[this] = h;
    }
};
// ImageTexture::new(filename) texture.rs:62-69. Decoding: .hdr, .png and .jpg natively (pt_load_hdr_rgb8, pt_load_png_rgb8, pt_load_jpeg_rgb8);
// anything else must have been handed over decoded (pt_register_image, keyed by the path
// relative to the asset directory; a registered image always wins) or sit next to the file as a "<file>.rgb8" sidecar
// ("PTRGB8 <w> <h>\n" + w*h*3 bytes; tools/prepare_assets.py writes them with Pillow).
struct ImageTexture : Texture<Vec3> {
    std::string filename;
    static std::shared_ptr<ImageTexture> new_(const std::string& f) { auto t = std::make_shared<ImageTexture>(); t->filename = f; return t; }
    int emit(Emitter& e) const override;
};

// ---- materials (src/bsdf/*.rs, src/material.rs:150-191) ---------------------------------
struct BxDFMaterial {
    virtual ~BxDFMaterial() = default;
    virtual int emit(Emitter& e) const = 0;
};
using MatPtr = std::shared_ptr<BxDFMaterial>;

struct DiffuseBRDF : BxDFMaterial {   // diffuse.rs:21-47
    TexPtr<Vec3> base_color;
    std::shared_ptr<ImageTexture> normal_map;
    static std::shared_ptr<DiffuseBRDF> new_(TexPtr<Vec3> c) { auto m = std::make_shared<DiffuseBRDF>(); m->base_color = c; return m; }
    static std::shared_ptr<DiffuseBRDF> from_rgb(Vec3 c) { return new_(SolidTexture<Vec3>::new_(c)); }
    static std::shared_ptr<DiffuseBRDF> with_normal(Vec3 c, std::shared_ptr<ImageTexture> n) { auto m = from_rgb(c); m->normal_map = n; return m; }
    static std::shared_ptr<DiffuseBRDF> from_textures(TexPtr<Vec3> c, std::shared_ptr<ImageTexture> n) { auto m = new_(c); m->normal_map = n; return m; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_diffuse(e.scene, base_color->emit(e), normal_map ? normal_map->emit(e) : -1);
        if (h < 0) panic("DiffuseBRDF");
        return e.done[this] = h;
    }
};
struct MetalBRDF : BxDFMaterial {   // metal.rs:23-35
    TexPtr<Vec3> base_color;
    TexPtr<double> roughness;
    static std::shared_ptr<MetalBRDF> new_(TexPtr<Vec3> c, TexPtr<double> r) { auto m = std::make_shared<MetalBRDF>(); m->base_color = c; m->roughness = r; return m; }
    static std::shared_ptr<MetalBRDF> from_rgb(Vec3 c, double r) { return new_(SolidTexture<Vec3>::new_(c), SolidTexture<double>::new_(r)); }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_metal(e.scene, base_color->emit(e), roughness->emit(e));
        if (h < 0) panic("MetalBRDF");
        return e.done[this] = h;
    }
};
struct GlassBSDF : BxDFMaterial {   // glass.rs:28-49
    TexPtr<Vec3> base_color;
    TexPtr<double> roughness;
    double anisotropic = 0, ior = 1.5;
    static std::shared_ptr<GlassBSDF> new_(TexPtr<Vec3> c, TexPtr<double> r, double aniso, double ior) {
        auto m = std::make_shared<GlassBSDF>();
        m->base_color = c; m->roughness = r; m->anisotropic = aniso; m->ior = ior;
        return m;
    }
    static std::shared_ptr<GlassBSDF> basic(double ior) { return new_(SolidTexture<Vec3>::new_(Vec3::ONE), SolidTexture<double>::new_(0.001), 0.0, ior); }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_glass(e.scene, base_color->emit(e), roughness->emit(e), anisotropic, ior);
        if (h < 0) panic("GlassBSDF");
        return e.done[this] = h;
    }
};
struct PrincipledBSDF : BxDFMaterial {   // principled.rs:45-73, same argument order
    TexPtr<Vec3> base_color;
    double p[11];
    static std::shared_ptr<PrincipledBSDF> new_(TexPtr<Vec3> base_color, double metallic, double roughness, double subsurface,
                                                double specular, double specular_tint, double ior, double spec_trans,
                                                double sheen, double sheen_tint, double clearcoat, double clearcoat_gloss) {
        auto m = std::make_shared<PrincipledBSDF>();
        m->base_color = base_color;
        const double v[11] = {metallic, roughness, subsurface, specular, specular_tint, ior, spec_trans, sheen, sheen_tint, clearcoat, clearcoat_gloss};
        std::memcpy(m->p, v, sizeof v);
        return m;
    }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_principled(e.scene, base_color->emit(e), p);
        if (h < 0) panic("PrincipledBSDF");
        return e.done[this] = h;
    }
};
struct DiffuseLight : BxDFMaterial {   // material.rs:155-164
    TexPtr<Vec3> emission;
    static std::shared_ptr<DiffuseLight> new_(TexPtr<Vec3> t) { auto m = std::make_shared<DiffuseLight>(); m->emission = t; return m; }
    static std::shared_ptr<DiffuseLight> from_rgb(Vec3 c) { return new_(SolidTexture<Vec3>::new_(c)); }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_light(e.scene, emission->emit(e));
        if (h < 0) panic("DiffuseLight");
        return e.done[this] = h;
    }
};

struct MixBxDf : BxDFMaterial {   // mix.rs:14-20
    double t;
    MatPtr bxdf1, bxdf2;
    static std::shared_ptr<MixBxDf> new_(double t, MatPtr a, MatPtr b) { auto m = std::make_shared<MixBxDf>(); m->t = t; m->bxdf1 = a; m->bxdf2 = b; return m; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_mix(e.scene, t, bxdf1->emit(e), bxdf2->emit(e));
        if (h < 0) panic("MixBxDf");
        return e.done[this] = h;
    }
};
struct SheenBRDF : BxDFMaterial {   // sheen.rs:17-22
    Vec3 base_color;
    double sheen_tint;
    static std::shared_ptr<SheenBRDF> new_(Vec3 c, double tint) { auto m = std::make_shared<SheenBRDF>(); m->base_color = c; m->sheen_tint = tint; return m; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_sheen(e.scene, base_color.x, base_color.y, base_color.z, sheen_tint);
        if (h < 0) panic("SheenBRDF");
        return e.done[this] = h;
    }
};
struct ClearcoatBRDF : BxDFMaterial {   // clearcoat.rs:14-18
    double clearcoat_gloss;
    static std::shared_ptr<ClearcoatBRDF> new_(double gloss) { auto m = std::make_shared<ClearcoatBRDF>(); m->clearcoat_gloss = gloss; return m; }
    int emit(Emitter& e) const override {
        auto it = e.done.find(this);
        if (it != e.done.end()) return it->second;
        int h = pt_mat_clearcoat(e.scene, clearcoat_gloss);
        if (h < 0) panic("ClearcoatBRDF");
        return e.done[this] = h;
    }
};

// ---- hittables (src/hittable/*.rs) -------------------------------------------------------
struct Hittable {
    virtual ~Hittable() = default;
    virtual int emit(Emitter& e) const = 0;
};
using HitPtr = std::shared_ptr<Hittable>;

struct Sphere : Hittable {   // sphere.rs:22-46
    double radius;
    Vec3 position1, position2;
    MatPtr material;
    static std::shared_ptr<Sphere> new_still(double r, Vec3 p, MatPtr m) { return new_moving(r, p, p, m); }
    static std::shared_ptr<Sphere> new_moving(double r, Vec3 p1, Vec3 p2, MatPtr m) {
        auto s = std::make_shared<Sphere>();
        s->radius = r; s->position1 = p1; s->position2 = p2; s->material = m;
        return s;
    }
    int emit(Emitter& e) const override {
        const double a[3] = {position1.x, position1.y, position1.z}, b[3] = {position2.x, position2.y, position2.z};
        int h = pt_sphere(e.scene, radius, a, b, material->emit(e));
        if (h < 0) panic("Sphere");
        return h;
    }
};
struct Quad : Hittable {   // quad.rs:17-36
    Vec3 q, u, v;
    MatPtr material;
    static std::shared_ptr<Quad> new_(Vec3 q, Vec3 u, Vec3 v, MatPtr m) {
        auto s = std::make_shared<Quad>();
        s->q = q; s->u = u; s->v = v; s->material = m;
        return s;
    }
    int emit(Emitter& e) const override {
        const double a[3] = {q.x, q.y, q.z}, b[3] = {u.x, u.y, u.z}, c[3] = {v.x, v.y, v.z};
        int h = pt_quad(e.scene, a, b, c, material->emit(e));
        if (h < 0) panic("Quad");
        return h;
    }
};
struct Cuboid : Hittable {   // cuboid.rs:11-58
    Vec3 a, b;
    MatPtr material;
    static std::shared_ptr<Cuboid> new_(Vec3 a, Vec3 b, MatPtr m) {
        auto s = std::make_shared<Cuboid>();
        s->a = a; s->b = b; s->material = m;
        return s;
    }
    int emit(Emitter& e) const override {
        const double p[3] = {a.x, a.y, a.z}, q[3] = {b.x, b.y, b.z};
        int h = pt_cuboid(e.scene, p, q, material->emit(e));
        if (h < 0) panic("Cuboid");
        return h;
    }
};
// tobj::Mesh as the reference consumes it (mesh.rs:149-170): f32 attributes, u32 position indices
struct Mesh {
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> indices;
};
namespace tobj {
struct Model { Mesh mesh; };
// tobj::load_obj(path, &OFFLINE_RENDERING_LOAD_OPTIONS).unwrap() — main.rs:408
inline std::vector<Model> load_obj(const std::string& path) {
    float *pos, *uv;
    uint32_t *idx, np, ni, nuv;
    if (pt_load_obj(path.c_str(), &pos, &np, &idx, &ni, &uv, &nuv) != 0) panic("tobj::load_obj");
    std::vector<Model> models(1);
    models[0].mesh.positions.assign(pos, pos + 3 * (size_t)np);
    models[0].mesh.indices.assign(idx, idx + ni);
    models[0].mesh.texcoords.assign(uv, uv + 2 * (size_t)nuv);
    pt_free(pos); pt_free(idx); pt_free(uv);
    return models;
}
}  // namespace tobj
struct TriangleMesh : Hittable {   // mesh.rs:149-197
    double scale;
    Mesh mesh;
    MatPtr material;
    static std::shared_ptr<TriangleMesh> from_obj(double scale, const Mesh& mesh, MatPtr m) {
        auto s = std::make_shared<TriangleMesh>();
        s->scale = scale; s->mesh = mesh; s->material = m;
        return s;
    }
    int emit(Emitter& e) const override {
        int h = pt_mesh(e.scene, scale, (uint32_t)(mesh.positions.size() / 3), mesh.positions.data(), (uint32_t)mesh.indices.size(),
                        mesh.indices.data(), (uint32_t)(mesh.normals.size() / 3), mesh.normals.data(),
                        (uint32_t)(mesh.texcoords.size() / 2), mesh.texcoords.data(), material->emit(e));
        if (h < 0) panic("TriangleMesh");
        return h;
    }
};
struct Instance : Hittable {   // instance.rs:20-30 — rotate, then translate
    HitPtr object;
    Vec3 axis, translation;
    double angle;
    static std::shared_ptr<Instance> new_(HitPtr obj, Vec3 axis, double angle, Vec3 translation) {
        auto s = std::make_shared<Instance>();
        s->object = obj; s->axis = axis; s->angle = angle; s->translation = translation;
        return s;
    }
    int emit(Emitter& e) const override {
        const double a[3] = {axis.x, axis.y, axis.z}, t[3] = {translation.x, translation.y, translation.z};
        int h = pt_instance(e.scene, object->emit(e), a, angle, t);
        if (h < 0) panic("Instance");
        return h;
    }
};

// ---- World (src/hittable/world.rs:10-29) -------------------------------------------------
struct World {
    std::vector<HitPtr> objects, lights;
    pt_scene* scene = nullptr;   // set by build_bvh / emit_into
    bool owns_scene = false;
    std::map<const void*, int> handles;   // description -> C-ABI handle (for Camera's env map)
    std::string asset_dir = "assets";
    static World new_() { return World(); }
    template <class T> void add_object(std::shared_ptr<T> o) { objects.push_back(o); }
    template <class T> void add_light(std::shared_ptr<T> o) { lights.push_back(o); }
    // flatten + BVH + upload into an existing scene
    void emit_into(pt_scene* s, std::shared_ptr<ImageTexture> env = nullptr) {
        Emitter e{s, asset_dir, {}};
        for (auto& o : objects) if (pt_world_add_object(s, o->emit(e)) != 0) panic("World::add_object");
        for (auto& l : lights) if (pt_world_add_light(s, l->emit(e)) != 0) panic("World::add_light");
        if (env) env->emit(e);
        if (pt_world_build(s) != 0) panic("World::build_bvh");
        scene = s;
        handles = e.done;
    }
    bool float_hdr = false;   // this build's option: Radiance .hdr images keep their f32 samples (no .to_rgb8(), texture.rs:67)
    void build_bvh(pt_ctx* ctx, std::shared_ptr<ImageTexture> env = nullptr) {
        pt_scene* s = pt_scene_create(ctx);
        if (!s) panic("pt_scene_create");
        owns_scene = true;
        if (float_hdr) pt_scene_set_float_hdr(s, 1);
        emit_into(s, env);
    }
    void release() {
        if (owns_scene && scene) pt_scene_destroy(scene);
        scene = nullptr;
        owns_scene = false;
    }
};

// ---- Camera (src/camera.rs:15-77) --------------------------------------------------------
struct EnvironmentType {
    bool is_map = false;
    Vec3 color;
    std::shared_ptr<ImageTexture> map;
    static EnvironmentType Color(Vec3 c) { EnvironmentType e; e.color = c; return e; }
    static EnvironmentType Map(std::shared_ptr<ImageTexture> t) { EnvironmentType e; e.is_map = true; e.map = t; return e; }
};
struct Camera {
    double aspect_ratio = 0;
    size_t image_width = 0, samples_per_pixel = 0, max_depth = 0;
    double vfov = 0;
    Vec3 look_from, look_at, vup;
    double blur_strength = 0, focal_length = 0, defocus_angle = 0;
    EnvironmentType environment = EnvironmentType::Color(Vec3::ZERO);
    size_t image_height = 0;
    double derived[18] = {0};   // forward,right,up,pixel00,pixel_du,pixel_dv
    static Camera new_() { return Camera(); }

    pt_camera to_c(const World* world) const {
        pt_camera c;
        std::memset(&c, 0, sizeof c);
        c.aspect_ratio = aspect_ratio;
        c.image_width = (uint32_t)image_width;
        c.samples_per_pixel = (uint32_t)samples_per_pixel;
        c.max_depth = (uint32_t)max_depth;
        c.vfov = vfov;
        const Vec3 v[3] = {look_from, look_at, vup};
        double* d[3] = {c.look_from, c.look_at, c.vup};
        for (int i = 0; i < 3; ++i) { d[i][0] = v[i].x; d[i][1] = v[i].y; d[i][2] = v[i].z; }
        c.blur_strength = blur_strength;
        c.focal_length = focal_length;
        c.defocus_angle = defocus_angle;
        c.env_color[0] = environment.color.x; c.env_color[1] = environment.color.y; c.env_color[2] = environment.color.z;
        c.env_tex = -1;
        if (environment.is_map) {
            c.env_is_map = 1;
            if (world) {
                auto it = world->handles.find(environment.map.get());
                if (it == world->handles.end()) throw std::runtime_error("Camera: environment map was not emitted with the world");
                c.env_tex = it->second;
            }
        }
        return c;
    }
    void init() {   // camera.rs:51-77
        pt_camera c = to_c(nullptr);
        uint32_t h = 0;
        if (pt_camera_init(&c, derived, &h) != 0) panic("Camera::init");
        image_height = h;
    }
    // camera.rs:79-126: render, gamma, quantise, save PNG, print the wall-clock seconds
    void render(World& world, const std::string& filename, uint64_t seed = 1, pt_render_stats* stats_out = nullptr) const {
        auto start = std::chrono::steady_clock::now();
        pt_camera c = to_c(&world);
        const size_t n = image_width * image_height;
        std::vector<double> accum(n * 3, 0.0);
        pt_render_stats st;
        if (pt_render(world.scene, &c, seed, 0, (uint32_t)samples_per_pixel, accum.data(), nullptr, &st) != 0) panic("Camera::render");
        std::vector<uint8_t> rgb(n * 3);
        if (pt_resolve_u8(pt_scene_ctx(world.scene), accum.data(), (uint32_t)n, (uint32_t)samples_per_pixel, rgb.data()) != 0) panic("Camera::render");
        if (pt_save_png(filename.c_str(), (uint32_t)image_width, (uint32_t)image_height, rgb.data()) != 0)
            std::fprintf(stderr, "Failed to save image %s\n", pt_last_error());
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        std::fprintf(stderr, "[camera.rs:125] start.elapsed().as_secs_f64() = %.6f (render kernel time %.3f s, %.2f Msamples/s)\n", secs,
                     st.ms_total * 1e-3, (double)st.samples / (st.ms_total * 1e-3) * 1e-6);
        if (stats_out) *stats_out = st;
    }
};

inline int ImageTexture::emit(Emitter& e) const {
    auto it = e.done.find(this);
    if (it != e.done.end()) return it->second;
    // key relative to the asset dir ("assets/bricks/color.png" -> "bricks/color.png")
    std::string key = filename;
    const std::string prefix = "assets/";
    if (key.compare(0, prefix.size(), prefix) == 0) key = key.substr(prefix.size());
    int h = pt_find_registered_image(e.scene, key.c_str());
    if (h >= 0) return e.done[this] = h;
    const std::string path = e.asset_dir + "/" + key;
    uint8_t* rgb = nullptr;
    uint32_t w = 0, hh = 0;
    if (key.size() > 4 && key.substr(key.size() - 4) == ".hdr" && pt_scene_float_hdr(e.scene)) {   // the float-HDR option: no to_rgb8()
        float* rgbf = nullptr;
        if (pt_load_hdr_rgbf32(path.c_str(), &rgbf, &w, &hh) != 0) panic("ImageTexture::new(" + filename + ")");
        h = pt_tex_image_rgbf32(e.scene, w, hh, rgbf);
        pt_free(rgbf);
    } else if (key.size() > 4 && key.substr(key.size() - 4) == ".hdr") {
        if (pt_load_hdr_rgb8(path.c_str(), &rgb, &w, &hh) != 0) panic("ImageTexture::new(" + filename + ")");
        h = pt_tex_image_rgb8(e.scene, w, hh, rgb);
        pt_free(rgb);
    } else if (key.size() > 4 && key.substr(key.size() - 4) == ".png") {
        if (pt_load_png_rgb8(path.c_str(), &rgb, &w, &hh) != 0) panic("ImageTexture::new(" + filename + ")");
        h = pt_tex_image_rgb8(e.scene, w, hh, rgb);
        pt_free(rgb);
    } else if ((key.size() > 4 && key.substr(key.size() - 4) == ".jpg") || (key.size() > 5 && key.substr(key.size() - 5) == ".jpeg")) {
        if (pt_load_jpeg_rgb8(path.c_str(), &rgb, &w, &hh) != 0) panic("ImageTexture::new(" + filename + ")");
        h = pt_tex_image_rgb8(e.scene, w, hh, rgb);
        pt_free(rgb);
    } else {
        std::ifstream in(path + ".rgb8", std::ios::binary);
        std::string magic;
        if (!in || !(in >> magic >> w >> hh) || magic != "PTRGB8")
            throw std::runtime_error("ImageTexture::new(" + filename + "): no decoder for this format in the host library; run tools/prepare_assets.py "
                                     "(writes " + path + ".rgb8) or hand the decoded pixels over with pt_register_image");
        in.get();
        std::vector<uint8_t> px((size_t)w * hh * 3);
        in.read((char*)px.data(), (std::streamsize)px.size());
        if (!in) throw std::runtime_error("ImageTexture::new(" + filename + "): truncated sidecar");
        h = pt_tex_image_rgb8(e.scene, w, hh, px.data());
    }
    if (h < 0) panic("ImageTexture::new");
    return e.done[this] = h;
}

}  // namespace path_tracer
