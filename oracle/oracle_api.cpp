// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h / orc_math.h headers).
// C API, asset ingest and the built-in scenes (literals of the reference's main.rs).
#include "oracle.h"

#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "orc_core.h"

using namespace orc;

int orc::g_math_mode = 0;
extern "C" void orc_set_math_mode(int det) { orc::g_math_mode = det ? 1 : 0; }
extern "C" int orc_get_math_mode(void) { return orc::g_math_mode; }
extern "C" double orc_detmath(int which, double a, double b) {
    switch (which) {
    case 3: return detmath::sin(a);
    case 4: return detmath::cos(a);
    case 5: return detmath::acos(a);
    case 6: return detmath::atan2(a, b);
    case 7: return detmath::pow(a, b);
    case 8: return detmath::log2(a);
    case 10: return detmath::log(a);
    case 11: return detmath::exp(a);
    }
    return std::nan("");
}

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return -1;
}
extern "C" const char* orc_last_error(void) { return g_err.c_str(); }

struct TexSlot {
    std::shared_ptr<TexRGB> rgb;
    std::shared_ptr<TexF> f;
    std::shared_ptr<ImageRGB8> img;  // set when rgb is an image (usable as normal/env map)
};
struct orc_scene {
    std::vector<TexSlot> tex;
    std::vector<std::shared_ptr<Material>> mats;
    std::vector<HitPtr> objs;
    World world;
    bool built = false;
    bool float_hdr = false;   // scene scripts load .hdr files as f32 (orc_scene_set_float_hdr)
    std::map<std::string, std::shared_ptr<ImageRGB8>> images;
};

extern "C" orc_scene* orc_scene_create(void) { return new orc_scene(); }
extern "C" void orc_scene_destroy(orc_scene* s) { delete s; }

#define CHECK_TEX_RGB(s, t) \
    if ((t) < 0 || (size_t)(t) >= (s)->tex.size() || !(s)->tex[t].rgb) return fail("bad rgb texture handle")
#define CHECK_TEX_F(s, t) \
    if ((t) < 0 || (size_t)(t) >= (s)->tex.size() || !(s)->tex[t].f) return fail("bad scalar texture handle")
#define CHECK_MAT(s, m) \
    if ((m) < 0 || (size_t)(m) >= (s)->mats.size()) return fail("bad material handle")
#define CHECK_OBJ(s, o) \
    if ((o) < 0 || (size_t)(o) >= (s)->objs.size()) return fail("bad object handle")

extern "C" int orc_tex_solid_rgb(orc_scene* s, double r, double g, double b) {
    TexSlot t;
    t.rgb = std::make_shared<SolidRGB>(V3{r, g, b});
    s->tex.push_back(t);
    return (int)s->tex.size() - 1;
}
extern "C" int orc_tex_solid_f(orc_scene* s, double v) {
    TexSlot t;
    t.f = std::make_shared<SolidF>(v);
    s->tex.push_back(t);
    return (int)s->tex.size() - 1;
}
extern "C" int orc_tex_checker(orc_scene* s, double scale, int t1, int t2) {
    CHECK_TEX_RGB(s, t1);
    CHECK_TEX_RGB(s, t2);
    TexSlot t;
    t.rgb = std::make_shared<CheckerRGB>(scale, s->tex[t1].rgb, s->tex[t2].rgb);
    s->tex.push_back(t);
    return (int)s->tex.size() - 1;
}
extern "C" int orc_tex_image_rgb8(orc_scene* s, uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (!rgb && w != 0 && h != 0) return fail("null image");
    TexSlot t;
    t.img = std::make_shared<ImageRGB8>(w, h, rgb);
    t.rgb = t.img;
    s->tex.push_back(t);
    return (int)s->tex.size() - 1;
}
extern "C" int orc_tex_image_rgbf32(orc_scene* s, uint32_t w, uint32_t h, const float* rgb) {   // float samples kept (ImageRGB8::pf)
    if (!rgb && w != 0 && h != 0) return fail("null image");
    TexSlot t;
    t.img = std::make_shared<ImageRGB8>(w, h, rgb);
    t.rgb = t.img;
    s->tex.push_back(t);
    return (int)s->tex.size() - 1;
}
extern "C" int orc_scene_set_float_hdr(orc_scene* s, int on) {
    s->float_hdr = on != 0;
    return 0;
}
extern "C" int orc_mat_diffuse(orc_scene* s, int color_tex, int nmap) {
    CHECK_TEX_RGB(s, color_tex);
    auto m = std::make_shared<DiffuseBRDF>();
    m->base_color = s->tex[color_tex].rgb;
    if (nmap >= 0) {
        if ((size_t)nmap >= s->tex.size() || !s->tex[nmap].img) return fail("normal map must be an image texture");
        m->nmap = s->tex[nmap].img;
    }
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_metal(orc_scene* s, int color_tex, int rough_tex) {
    CHECK_TEX_RGB(s, color_tex);
    CHECK_TEX_F(s, rough_tex);
    auto m = std::make_shared<MetalBRDF>();
    m->base_color = s->tex[color_tex].rgb;
    m->roughness = s->tex[rough_tex].f;
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_glass(orc_scene* s, int color_tex, int rough_tex, double, double ior) {
    CHECK_TEX_RGB(s, color_tex);
    CHECK_TEX_F(s, rough_tex);
    auto m = std::make_shared<GlassBSDF>();
    m->base_color = s->tex[color_tex].rgb;
    m->roughness = s->tex[rough_tex].f;
    m->ior = ior;
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_principled(orc_scene* s, int color_tex, const double p[11]) {
    CHECK_TEX_RGB(s, color_tex);
    auto m = std::make_shared<PrincipledBSDF>();
    m->base_color = s->tex[color_tex].rgb;
    m->metallic = p[0]; m->roughness = p[1]; m->subsurface = p[2]; m->specular = p[3];
    m->specular_tint = p[4]; m->ior = p[5]; m->spec_trans = p[6]; m->sheen = p[7];
    m->sheen_tint = p[8]; m->clearcoat = p[9]; m->clearcoat_gloss = p[10];
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_mix(orc_scene* s, double t, int m1, int m2) {
    CHECK_MAT(s, m1);
    CHECK_MAT(s, m2);
    auto m = std::make_shared<MixBxDf>();
    m->t = clampd(t, 0.0, 1.0);   // mix.rs:16
    m->a = s->mats[m1];
    m->b = s->mats[m2];
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_sheen(orc_scene* s, double r, double g, double b, double sheen_tint) {
    auto m = std::make_shared<SheenBRDF>();
    m->base_color = V3{r, g, b};
    m->sheen_tint = sheen_tint;
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
extern "C" int orc_mat_clearcoat(orc_scene* s, double clearcoat_gloss) {
    auto m = std::make_shared<ClearcoatBRDF>();
    m->alpha_g = (1.0 - clearcoat_gloss) * 0.1 + clearcoat_gloss * 0.001;   // clearcoat.rs:16
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
// material-level probe for the known-answer tests: pdf and eval of material `mat` at a hit with unit
// normal n (geometric == shading), uv = (0.5, 0.5), for world-space view / light directions.
static bool g_probe_front_face = true;   // HitInfo::front_face of the synthetic hit of the two probes below (hit_info.rs:24)
extern "C" void orc_mat_probe_front_face(int front) { g_probe_front_face = front != 0; }
extern "C" int orc_mat_probe(orc_scene* s, int mat, const double* n, const double* wo, const double* wi, double* out4) {
    if (mat < 0 || mat >= (int)s->mats.size()) return -1;
    HitInfo info{};
    info.geometric_normal = info.shading_normal = normalize(V3{n[0], n[1], n[2]});
    info.front_face = g_probe_front_face;
    info.u = info.v = 0.5;
    info.mat = s->mats[mat].get();
    V3 v{wo[0], wo[1], wo[2]}, l{wi[0], wi[1], wi[2]};
    out4[0] = info.mat->pdf(v, l, info);
    V3 f = info.mat->eval(v, l, info);
    out4[1] = f.x; out4[2] = f.y; out4[3] = f.z;
    return 0;
}
// material-level sampling probe: n_samples directions drawn by BxDFMaterial::sample at the same synthetic hit
// (normal n, uv (0.5, 0.5)) for view direction wo; out = n_samples x 4 (dir xyz, 1 = Some / 0 = None).
extern "C" int orc_mat_sample_probe(orc_scene* s, int mat, const double* n, const double* wo, uint64_t seed, uint32_t n_samples, double* out) {
    if (mat < 0 || mat >= (int)s->mats.size()) return -1;
    HitInfo info{};
    info.geometric_normal = info.shading_normal = normalize(V3{n[0], n[1], n[2]});
    info.front_face = g_probe_front_face;
    info.u = info.v = 0.5;
    info.mat = s->mats[mat].get();
    V3 v = normalize(V3{wo[0], wo[1], wo[2]});
    Ray ray{info.geometric_normal, -v, 0.0};   // ray.direction() = -view_dir, as trace() calls sample (camera.rs:207)
    for (uint32_t i = 0; i < n_samples; ++i) {
        Rng rng(seed, 0u, i);
        V3 d{0, 0, 0};
        bool ok = info.mat->sample(ray, info, rng, d);
        out[4 * i] = d.x; out[4 * i + 1] = d.y; out[4 * i + 2] = d.z; out[4 * i + 3] = ok ? 1.0 : 0.0;
    }
    return 0;
}
extern "C" int orc_mat_light(orc_scene* s, int tex) {
    CHECK_TEX_RGB(s, tex);
    auto m = std::make_shared<DiffuseLight>();
    m->emission = s->tex[tex].rgb;
    s->mats.push_back(m);
    return (int)s->mats.size() - 1;
}
static V3 V(const double* p) { return V3{p[0], p[1], p[2]}; }
extern "C" int orc_sphere(orc_scene* s, double r, const double p1[3], const double p2[3], int mat) {
    CHECK_MAT(s, mat);
    s->objs.push_back(std::make_shared<Sphere>(r, V(p1), V(p2), s->mats[mat]));
    return (int)s->objs.size() - 1;
}
extern "C" int orc_quad(orc_scene* s, const double q[3], const double u[3], const double v[3], int mat) {
    CHECK_MAT(s, mat);
    s->objs.push_back(std::make_shared<Quad>(V(q), V(u), V(v), s->mats[mat]));
    return (int)s->objs.size() - 1;
}
extern "C" int orc_cuboid(orc_scene* s, const double a[3], const double b[3], int mat) {
    CHECK_MAT(s, mat);
    s->objs.push_back(std::make_shared<Cuboid>(V(a), V(b), s->mats[mat]));
    return (int)s->objs.size() - 1;
}
extern "C" int orc_mesh(orc_scene* s, double scale, uint32_t n_pos, const float* pos, uint32_t n_idx,
                        const uint32_t* idx, uint32_t n_nrm, const float* nrm, uint32_t n_uv,
                        const float* uv, int mat) {
    CHECK_MAT(s, mat);
    for (uint32_t i = 0; i < n_idx; ++i) {
        if (idx[i] >= n_pos) return fail("mesh index out of range");
        if (n_nrm && idx[i] >= n_nrm) return fail("mesh normal index out of range");
        if (n_uv && idx[i] >= n_uv) return fail("mesh uv index out of range");
    }
    s->objs.push_back(std::make_shared<TriangleMesh>(scale, n_pos, pos, n_idx, idx, n_nrm, nrm, n_uv, uv, s->mats[mat]));
    return (int)s->objs.size() - 1;
}
// Instance::new and World::add_object / add_light take an Arc<dyn Hittable> (instance.rs:20-30, world.rs:18-24): an object may be
// shared by any number of instances, may be an instance itself, and may be added to the world directly any number of times as
// well. Canonical primitive ids are per PLACEMENT (orc_core.h Placement / Instance::id_offset).
extern "C" int orc_instance(orc_scene* s, int obj, const double axis[3], double angle, const double tr[3]) {
    CHECK_OBJ(s, obj);
    s->objs.push_back(std::make_shared<Instance>(s->objs[obj], V(axis), angle, V(tr)));
    return (int)s->objs.size() - 1;
}
extern "C" int orc_world_add_object(orc_scene* s, int obj) {
    CHECK_OBJ(s, obj);
    s->world.objects.add(std::make_shared<Placement>(s->objs[obj]));
    return 0;
}
extern "C" int orc_world_add_light(orc_scene* s, int obj) {
    CHECK_OBJ(s, obj);
    s->world.lights.add(std::make_shared<Placement>(s->objs[obj]));
    return 0;
}
extern "C" int orc_world_build(orc_scene* s) {
    s->world.build_bvh();
    s->built = true;
    return 0;
}
extern "C" uint32_t orc_world_prim_count(orc_scene* s) { return s->world.n_prims; }
extern "C" void orc_free(void* p) { free(p); }

// ------------------------------------------------------------------------ asset ingest
// Wavefront OBJ as tobj 4.0.2 reads it for the reference (main.rs:408,433,458): `v` parsed
// as f32, faces fan-triangulated, 1-based (or negative, relative) position indices -> u32,
// `vt` parsed as f32 pairs; only position indices are kept (mesh.rs:173-175 uses them for
// every attribute).
extern "C" int orc_load_obj(const char* path, float** pos, uint32_t* n_pos, uint32_t** idx,
                            uint32_t* n_idx, float** uv, uint32_t* n_uv) {
    FILE* f = fopen(path, "r");
    if (!f) return fail(std::string("cannot open ") + path);
    std::vector<float> P, T;
    std::vector<uint32_t> I;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        char* p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e = p + 1;
            for (int k = 0; k < 3; ++k) P.push_back(strtof(e, &e));
        } else if (p[0] == 'v' && p[1] == 't') {
            char* e = p + 2;
            for (int k = 0; k < 2; ++k) T.push_back(strtof(e, &e));
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::vector<uint32_t> poly;
            char* e = p + 1;
            for (;;) {
                while (*e == ' ' || *e == '\t') ++e;
                if (*e == '\0' || *e == '\n' || *e == '\r') break;
                long v = strtol(e, &e, 10);
                long nv = (long)(P.size() / 3);
                if (v < 0) v = nv + v; else v = v - 1;
                if (v < 0 || v >= nv) { fclose(f); return fail("OBJ face index out of range"); }
                poly.push_back((uint32_t)v);
                while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') ++e;  // skip /vt/vn
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k) {
                I.push_back(poly[0]); I.push_back(poly[k]); I.push_back(poly[k + 1]);
            }
        }
    }
    fclose(f);
    *n_pos = (uint32_t)(P.size() / 3);
    *n_idx = (uint32_t)I.size();
    *n_uv = (uint32_t)(T.size() / 2);
    *pos = (float*)malloc(P.size() * sizeof(float) + 4);
    *idx = (uint32_t*)malloc(I.size() * sizeof(uint32_t) + 4);
    *uv = (float*)malloc(T.size() * sizeof(float) + 4);
    if (!P.empty()) memcpy(*pos, P.data(), P.size() * sizeof(float));     // (an empty vector's data() may be null: UB for memcpy even with n = 0)
    if (!I.empty()) memcpy(*idx, I.data(), I.size() * sizeof(uint32_t));
    if (!T.empty()) memcpy(*uv, T.data(), T.size() * sizeof(float));
    return 0;
}

// Radiance .hdr -> f32 RGB as image 0.25.5's HdrDecoder does (RGBE -> mantissa * 2^(e-136), e == 0 -> 0); the RGB8 form is
// `decode().to_rgb8()` (texture.rs:62-67): round(clamp(x,0,1)*255).
static int load_hdr_f32(const char* path, std::vector<float>& out, uint32_t* w, uint32_t* h) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(std::string("cannot open ") + path);
    char line[512];
    bool magic = false;
    int W = 0, H = 0;
    while (fgets(line, sizeof line, f)) {
        if (!magic) {
            if (strncmp(line, "#?", 2) != 0) { fclose(f); return fail("not a Radiance file"); }
            magic = true;
            continue;
        }
        if (line[0] == '\n') {  // blank line ends the header; next line is the resolution
            if (!fgets(line, sizeof line, f)) break;
            if (sscanf(line, "-Y %d +X %d", &H, &W) != 2) { fclose(f); return fail("unsupported HDR orientation"); }
            break;
        }
    }
    if (W <= 0 || H <= 0 || (uint64_t)W * (uint64_t)H > (1ull << 28)) { fclose(f); return fail("bad HDR header"); }
    std::vector<uint8_t> scan((size_t)W * 4);
    out.assign((size_t)W * H * 3, 0.0f);
    for (int y = 0; y < H; ++y) {
        uint8_t hd[4];
        if (fread(hd, 1, 4, f) != 4) { fclose(f); return fail("truncated HDR"); }
        if (hd[0] == 2 && hd[1] == 2 && (hd[2] & 0x80) == 0 && ((hd[2] << 8) | hd[3]) == W) {
            for (int c = 0; c < 4; ++c) {  // new-style RLE, channel-planar per scanline
                int x = 0;
                while (x < W) {
                    int n = fgetc(f);
                    if (n == EOF) { fclose(f); return fail("truncated HDR"); }
                    if (n > 128) {
                        n -= 128;
                        int v = fgetc(f);
                        if (v == EOF || x + n > W) { fclose(f); return fail("bad HDR run"); }
                        while (n--) scan[(size_t)(x++) * 4 + c] = (uint8_t)v;
                    } else {
                        if (n == 0 || x + n > W) { fclose(f); return fail("bad HDR run"); }
                        while (n--) {
                            int v = fgetc(f);
                            if (v == EOF) { fclose(f); return fail("truncated HDR"); }
                            scan[(size_t)(x++) * 4 + c] = (uint8_t)v;
                        }
                    }
                }
            }
        } else {  // flat RGBE scanline
            memcpy(scan.data(), hd, 4);
            if (fread(scan.data() + 4, 1, (size_t)(W - 1) * 4, f) != (size_t)(W - 1) * 4) { fclose(f); return fail("truncated HDR"); }
        }
        for (int x = 0; x < W; ++x) {
            const uint8_t* p = &scan[(size_t)x * 4];
            for (int c = 0; c < 3; ++c) out[((size_t)y * W + x) * 3 + c] = p[3] != 0 ? (float)p[c] * ldexpf(1.0f, (int)p[3] - 136) : 0.0f;
        }
    }
    fclose(f);
    *w = (uint32_t)W;
    *h = (uint32_t)H;
    return 0;
}
extern "C" int orc_load_hdr_rgbf32(const char* path, float** rgb, uint32_t* w, uint32_t* h) {
    std::vector<float> v;
    if (load_hdr_f32(path, v, w, h) != 0) return -1;
    *rgb = (float*)malloc(v.size() * sizeof(float) + 4);
    memcpy(*rgb, v.data(), v.size() * sizeof(float));
    return 0;
}
extern "C" int orc_load_hdr_rgb8(const char* path, uint8_t** rgb, uint32_t* w, uint32_t* h) {
    std::vector<float> v;
    if (load_hdr_f32(path, v, w, h) != 0) return -1;
    uint8_t* out = (uint8_t*)malloc(v.size() + 4);
    for (size_t i = 0; i < v.size(); ++i) {
        float cl = v[i] < 0.0f ? 0.0f : (v[i] > 1.0f ? 1.0f : v[i]);
        out[i] = (uint8_t)roundf(cl * 255.0f);
    }
    *rgb = out;
    return 0;
}

extern "C" int orc_register_image(orc_scene* s, const char* name, uint32_t w, uint32_t h, const uint8_t* rgb) {
    s->images[name] = std::make_shared<ImageRGB8>(w, h, rgb);
    return 0;
}

// ------------------------------------------------------------------------------ scenes
namespace {
struct SceneBuilder {
    orc_scene* s;
    std::string dir;
    int solid(double r, double g, double b) { return orc_tex_solid_rgb(s, r, g, b); }
    int diffuse_rgb(double r, double g, double b) { return orc_mat_diffuse(s, solid(r, g, b), -1); }
    int metal_rgb(double r, double g, double b, double rough) { return orc_mat_metal(s, solid(r, g, b), orc_tex_solid_f(s, rough)); }
    int glass_basic(double ior) { return orc_mat_glass(s, solid(1, 1, 1), orc_tex_solid_f(s, 0.001), 0.0, ior); }  // glass.rs:42-49
    int light_rgb(double r, double g, double b) { return orc_mat_light(s, solid(r, g, b)); }
    int sphere(double r, double x, double y, double z, int m) {
        double p[3] = {x, y, z};
        return orc_sphere(s, r, p, p, m);
    }
    int quad(double qx, double qy, double qz, double ux, double uy, double uz, double vx, double vy, double vz, int m) {
        double q[3] = {qx, qy, qz}, u[3] = {ux, uy, uz}, v[3] = {vx, vy, vz};
        return orc_quad(s, q, u, v, m);
    }
    int box_instance(double bx, double by, double bz, int m, double angle, double tx, double ty, double tz) {
        double a[3] = {0, 0, 0}, b[3] = {bx, by, bz}, axis[3] = {0, 1, 0}, t[3] = {tx, ty, tz};
        return orc_instance(s, orc_cuboid(s, a, b, m), axis, angle, t);
    }
    int image(const std::string& name) {  // ImageTexture::new, texture.rs:62-69
        auto it = s->images.find(name);
        if (it != s->images.end()) {
            TexSlot t;
            t.img = it->second;
            t.rgb = t.img;
            s->tex.push_back(t);
            return (int)s->tex.size() - 1;
        }
        if (name.size() > 4 && name.substr(name.size() - 4) == ".hdr") {
            uint32_t w, h;
            if (s->float_hdr) {   // the build's option: keep the decoder's f32 samples (no to_rgb8 squash)
                float* rgbf;
                if (orc_load_hdr_rgbf32((dir + "/" + name).c_str(), &rgbf, &w, &h) != 0) return -1;
                int t = orc_tex_image_rgbf32(s, w, h, rgbf);
                free(rgbf);
                return t;
            }
            uint8_t* rgb;
            if (orc_load_hdr_rgb8((dir + "/" + name).c_str(), &rgb, &w, &h) != 0) return -1;
            int t = orc_tex_image_rgb8(s, w, h, rgb);
            free(rgb);
            return t;
        }
        return fail("image '" + name + "' must be registered (orc_register_image) — the oracle decodes only .hdr");
    }
    int obj_instance(const std::string& name, double scale, int mat, double angle, double tx, double ty, double tz) {
        float *pos, *uv;
        uint32_t *idx, np, ni, nuv;
        if (orc_load_obj((dir + "/" + name).c_str(), &pos, &np, &idx, &ni, &uv, &nuv) != 0) return -1;
        int m = orc_mesh(s, scale, np, pos, ni, idx, 0, nullptr, nuv, uv, mat);
        free(pos); free(idx); free(uv);
        if (m < 0) return -1;
        double axis[3] = {0, 1, 0}, t[3] = {tx, ty, tz};
        return orc_instance(s, m, axis, angle, t);
    }
    int principled(int tex, double a, double b, double c, double d, double e, double f, double g, double h, double i, double j, double k) {
        double p[11] = {a, b, c, d, e, f, g, h, i, j, k};
        return orc_mat_principled(s, tex, p);
    }
};
void cam_defaults(orc_camera* c, uint32_t width, uint32_t spp) {
    memset(c, 0, sizeof *c);
    c->image_width = width;
    c->samples_per_pixel = spp;
    c->max_depth = 50;
    c->vup[1] = 1.0;
    c->blur_strength = 0.5;
    c->env_tex = -1;
}
void set3(double* d, double x, double y, double z) { d[0] = x; d[1] = y; d[2] = z; }
}  // namespace

#define ADD(o) do { int _o = (o); if (_o < 0 || orc_world_add_object(s, _o) != 0) return -1; } while (0)
#define ADDL(o) do { int _o = (o); if (_o < 0 || orc_world_add_light(s, _o) != 0) return -1; } while (0)

extern "C" int orc_build_scene(orc_scene* s, int scene_id, uint32_t width, uint32_t spp, const char* asset_dir,
                               const uint8_t* env_rgb8, uint32_t env_w, uint32_t env_h, uint64_t scene_seed,
                               orc_camera* cam) {
    SceneBuilder b{s, asset_dir ? asset_dir : "assets"};
    cam_defaults(cam, width, spp);
    auto env_image = [&](const char* name) -> int {
        int t = env_rgb8 ? orc_tex_image_rgb8(s, env_w, env_h, env_rgb8) : b.image(name);
        if (t >= 0) { cam->env_is_map = 1; cam->env_tex = t; }
        return t;
    };
    switch (scene_id) {
    case 1: {  // balls_scene main.rs:14-82; the reference's unseeded build-time RNG is replaced
               // by Philox(scene_seed) draws in the same order.
        int checker = orc_tex_checker(s, 0.32, b.solid(0.2, 0.3, 0.1), b.solid(0.9, 0.9, 0.9));
        ADD(b.sphere(1000.0, 0.0, -1000.0, 0.0, orc_mat_diffuse(s, checker, -1)));
        ADD(b.sphere(1.0, 0.0, 1.0, 0.0, b.glass_basic(1.5)));
        ADD(b.sphere(1.0, -4.0, 1.0, 0.0, b.diffuse_rgb(0.4, 0.2, 0.1)));
        ADD(b.sphere(1.0, 4.0, 1.0, 0.0, b.metal_rgb(0.7, 0.6, 0.5, 0.0)));
        Rng rng(scene_seed, 0xBA115u, 0u);
        for (int ai = -11; ai < 11; ++ai)
            for (int bi = -11; bi < 11; ++bi) {
                double a = (double)ai, bb = (double)bi;
                double choose = rng.gen();
                double cx = a + 0.9 * rng.gen();
                double cz = bb + 0.9 * rng.gen();
                V3 center{cx, 0.2, cz};
                if (length(center - V3{4.0, 0.2, 0.0}) > 0.9) {
                    if (choose < 0.8) {
                        double r1 = rng.gen(), g1 = rng.gen(), b1 = rng.gen();
                        double r2 = rng.gen(), g2 = rng.gen(), b2 = rng.gen();
                        int m = b.diffuse_rgb(r1 * r2, g1 * g2, b1 * b2);
                        double p1[3] = {cx, 0.2, cz}, p2[3] = {cx, 0.2 + 0.5 * rng.gen(), cz};
                        ADD(orc_sphere(s, 0.2, p1, p2, m));
                    } else if (choose < 0.95) {
                        double r1 = 0.5 + 0.5 * rng.gen(), g1 = 0.5 + 0.5 * rng.gen(), b1 = 0.5 + 0.5 * rng.gen();
                        ADD(b.sphere(0.2, cx, 0.2, cz, b.metal_rgb(r1, g1, b1, 0.0)));
                    } else {
                        ADD(b.sphere(0.2, cx, 0.2, cz, b.glass_basic(1.5)));
                    }
                }
            }
        cam->aspect_ratio = 16.0 / 9.0;
        cam->vfov = 20.0;
        set3(cam->look_from, 13.0, 2.0, 3.0);
        set3(cam->look_at, 0, 0, 0);
        cam->focal_length = 10.0;
        cam->defocus_angle = 0.6;
        set3(cam->env_color, 0.7, 0.8, 1.0);
        break;
    }
    case 2: {  // earth_scene main.rs:84-132
        int earth = b.image("earthmap.jpg");
        if (earth < 0) return -1;
        ADD(b.sphere(1.0, 4.9, 1.0, 3.0, orc_mat_diffuse(s, earth, -1)));
        ADD(b.sphere(1.0, 0.0, 1.0, 0.0, b.diffuse_rgb(0.4, 0.2, 0.1)));
        ADD(b.sphere(1.0, 4.0, 1.0, 0.0, b.metal_rgb(0.7, 0.6, 0.5, 0.1)));
        int checker = orc_tex_checker(s, 0.62, b.solid(0.9, 0.0, 0.1), b.solid(0.9, 0.9, 0.9));
        ADD(b.sphere(1000.0, 0.0, -1000.0, 0.0, orc_mat_diffuse(s, checker, -1)));
        cam->aspect_ratio = 16.0 / 9.0;
        cam->vfov = 28.0;
        set3(cam->look_from, 8.8, 2.0, 3.0);
        set3(cam->look_at, 0, 0, 0);
        cam->focal_length = 2.869817807;
        cam->defocus_angle = 2.5;
        set3(cam->env_color, 0.85, 0.85, 1.0);
        break;
    }
    case 3:    // cornell_box_scene main.rs:134-236
    case 7: {  // normal_demo_scene main.rs:534-618 (same room)
        int white = b.diffuse_rgb(0.73, 0.73, 0.73);
        if (scene_id == 3) {
            int red = b.diffuse_rgb(0.65, 0.05, 0.05), green = b.diffuse_rgb(0.12, 0.45, 0.15);
            ADD(b.quad(555, 0, 0, 0, 555, 0, 0, 0, 555, green));
            ADD(b.quad(0, 0, 0, 0, 555, 0, 0, 0, 555, red));
        } else {
            int albedo = b.image("bricks/color.png");
            int nrm = b.image("bricks/normal.png");
            if (albedo < 0 || nrm < 0) return -1;
            ADD(b.quad(555, 0, 0, 0, 555, 0, 0, 0, 555, orc_mat_diffuse(s, albedo, -1)));
            ADD(b.quad(0, 0, 0, 0, 555, 0, 0, 0, 555, orc_mat_diffuse(s, albedo, nrm)));
        }
        ADD(b.quad(0, 0, 0, 555, 0, 0, 0, 0, 555, white));
        ADD(b.quad(555, 555, 555, -555, 0, 0, 0, 0, -555, white));
        ADD(b.quad(0, 0, 555, 555, 0, 0, 0, 555, 0, white));
        if (scene_id == 3) {
            ADDL(b.quad(343, 554, 332, -130, 0, 0, 0, 0, -105, b.light_rgb(25, 25, 25)));
            int pm = b.principled(b.solid(1, 1, 1), 0.01, 0.01, 0.01, 0.91, 0.91, 1.5, 0.91, 0.91, 0.91, 0.91, 0.01);
            ADD(b.sphere(135.0, 113.0, 170.0, 372.0, pm));
            ADD(b.box_instance(165, 330, 165, b.metal_rgb(1, 1, 1, 0.1), 0.261799, 265, 0, 295));
            ADD(b.box_instance(165, 165, 165, white, -0.29, 130, 0, 65));
        } else {
            ADDL(b.quad(343, 554, 332, -130, 0, 0, 0, 0, -105, b.light_rgb(27, 28, 20)));
            ADD(b.box_instance(165, 330, 165, b.metal_rgb(0.94, 0.94, 0.94, 0.1), 0.261799, 265, 0, 295));
            ADD(b.sphere(100.0, 130.0, 100.0, 65.0, b.glass_basic(1.5)));
        }
        cam->aspect_ratio = 1.0;
        cam->vfov = 40.0;
        set3(cam->look_from, 278, 278, -800);
        set3(cam->look_at, 278, 278, 0);
        cam->focal_length = 10.0;
        cam->defocus_angle = 0.0;
        break;
    }
    case 4: {  // environment_map_scene main.rs:238-274
        ADD(b.sphere(9.0, 4.0, 2.0, 0.0, b.metal_rgb(1, 1, 1, 0.001)));
        ADD(b.quad(-2.0, 6.5, 0.0, 4.0, 0, 0, 0, 0, 2.0, b.light_rgb(10, 10, 10)));
        cam->aspect_ratio = 16.0 / 9.0;
        cam->vfov = 90.0;
        set3(cam->look_from, 0.0, 3.0, 17.0);
        set3(cam->look_at, 0.0, 2.0, 0.0);
        cam->focal_length = 17.0;
        cam->defocus_angle = 1.5;
        if (env_image("grace_probe_latlong.hdr") < 0) return -1;
        break;
    }
    case 5: {  // bsdf_demo_scene main.rs:276-369
        for (int row = 0; row < 3; ++row)
            for (int i = 0; i < 5; ++i) {
                double rough = 0.1 + 0.2 * (double)i;
                int m;
                if (row == 0) m = b.principled(b.solid(0.65, 0.05, 0.05), 0.00, rough, 0.01, 0.01, 0.01, 1.5, 0.01, 0.01, 0.01, 0.01, 0.01);
                else if (row == 1) m = b.principled(b.solid(0.05, 0.65, 0.05), 0.99, rough, 0.01, 0.01, 0.01, 1.5, 0.01, 0.01, 0.01, 0.01, 0.01);
                else m = b.principled(b.solid(0.25, 0.05, 0.65), 0.01, rough * 0.3, 0.01, 0.01, 0.01, 1.5, 0.99, 0.01, 0.01, 0.01, 0.01);
                ADD(b.sphere(0.5, -4.0 + (double)i, 1.0 + (double)row, -5.0, m));
            }
        cam->aspect_ratio = 16.0 / 9.0;
        cam->vfov = 60.0;
        set3(cam->look_from, -2.0, 2.0, -1.0);
        set3(cam->look_at, -2.0 + 0.0, 2.0 + 0.0, -1.0 + -1000.0);
        cam->focal_length = 5.0;
        cam->defocus_angle = 0.0;
        if (env_image("envmap.jpg") < 0) return -1;
        break;
    }
    case 6: {  // everything_scene main.rs:371-532
        int checker = orc_tex_checker(s, 0.92, b.solid(0.2, 0.3, 0.1), b.solid(0.9, 0.9, 0.9));
        ADD(b.quad(-1000, 0, -1000, 0, 0, 5000, 5000, 0, 0, orc_mat_diffuse(s, checker, -1)));
        ADD(b.sphere(2.0, -4.0, 2.0, 9.8, b.metal_rgb(1, 1, 1, 0.001)));
        ADD(b.sphere(1.0, 4.0, 1.0, 6.0, b.glass_basic(1.5)));
        ADD(b.box_instance(1.0, 2.0, 1.0, b.diffuse_rgb(0.0, 0.5, 1.0), 0.5, 1.2, 0.0, 6.0));
        int bunny = b.principled(b.solid(1, 1, 1), 0.91, 0.01, 0.01, 0.01, 0.91, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
        ADD(b.obj_instance("bunny.obj", 10.0, bunny, 3.14, 0.1, -0.327, 5.0));
        int spot = b.principled(b.solid(0.65, 0.05, 0.05), 0.01, 0.01, 0.91, 0.01, 0.01, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
        ADD(b.obj_instance("spot.obj", 0.65, spot, 0.87, -1.5, 2.8, 4.3));
        int cow = b.principled(b.solid(0.05, 0.65, 0.05), 0.91, 0.21, 0.91, 0.01, 0.01, 1.5, 0.01, 0.91, 0.91, 0.91, 0.01);
        ADD(b.obj_instance("cow.obj", 0.75, cow, 0.93, 2.5, 3.8, 12.0));
        ADD(b.sphere(0.1, 1.0, 0.1, 3.0, b.light_rgb(20, 20, 10)));
        ADD(b.sphere(0.2, 0.0, 0.2, 3.0, b.metal_rgb(0.6, 0.05, 0.05, 0.1)));
        ADD(b.sphere(0.3, 1.2, 0.3, 3.4, orc_mat_glass(s, b.solid(0.7, 0.3, 0.3), orc_tex_solid_f(s, 0.3), 0.0, 1.5)));
        cam->aspect_ratio = 16.0 / 9.0;
        cam->vfov = 60.0;
        set3(cam->look_from, 0.0, 1.5, 0.0);
        set3(cam->look_at, 0.0, 1.5, 100000.0);
        cam->focal_length = 6.0;
        cam->defocus_angle = 1.0;
        if (env_image("grace_probe_latlong.hdr") < 0) return -1;
        break;
    }
    default:
        return fail("unknown scene id");
    }
    return orc_world_build(s);
}

// ------------------------------------------------------------------------ camera/render
static int make_camera(orc_scene* s, const orc_camera* c, Camera& cam) {
    cam.aspect_ratio = c->aspect_ratio;
    cam.image_width = c->image_width;
    cam.samples_per_pixel = c->samples_per_pixel;
    cam.max_depth = c->max_depth;
    cam.vfov = c->vfov;
    cam.look_from = V(c->look_from);
    cam.look_at = V(c->look_at);
    cam.vup = V(c->vup);
    cam.blur_strength = c->blur_strength;
    cam.focal_length = c->focal_length;
    cam.defocus_angle = c->defocus_angle;
    cam.env_is_map = c->env_is_map != 0;
    cam.env_color = V(c->env_color);
    if (cam.env_is_map) {
        if (!s || c->env_tex < 0 || (size_t)c->env_tex >= s->tex.size() || !s->tex[c->env_tex].img) return fail("env_tex must be an image texture");
        cam.env_map = s->tex[c->env_tex].img;
    }
    if (cam.image_width == 0 || !(cam.aspect_ratio > 0.0)) return fail("bad camera size");
    cam.init();
    if (cam.image_height == 0) return fail("image height is zero");
    return 0;
}
extern "C" int orc_camera_init(const orc_camera* c, double out[18], uint32_t* image_height) {
    Camera cam;
    orc_camera cc = *c;
    cc.env_is_map = 0;
    if (make_camera(nullptr, &cc, cam) != 0) return -1;
    const V3 v[6] = {cam.forward, cam.right, cam.up, cam.pixel00, cam.pixel_du, cam.pixel_dv};
    for (int i = 0; i < 6; ++i) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
    *image_height = cam.image_height;
    return 0;
}
extern "C" int orc_render(orc_scene* s, const orc_camera* c, uint64_t seed, uint32_t spp_begin, uint32_t spp_end,
                          double* accum, uint64_t counters[4], int nthreads) {
    if (!s->built) return fail("world not built");
    Camera cam;
    if (make_camera(s, c, cam) != 0) return -1;
    const uint32_t W = cam.image_width, H = cam.image_height;
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    Counters total;
#pragma omp parallel num_threads(nthreads)
    {
        Counters cnt;
        // rayon's par_enumerate_pixels_mut analogue (camera.rs:102): dynamic over pixels
#pragma omp for schedule(dynamic, 64)
        for (int64_t p = 0; p < (int64_t)W * H; ++p) {
            uint32_t y = (uint32_t)(p / W), x = (uint32_t)(p % W);
            V3 color{0, 0, 0};
            for (uint32_t sidx = spp_begin; sidx < spp_end; ++sidx) {  // camera.rs:106-108
                Rng rng(seed, (uint32_t)p, sidx);
                color += cam.trace(y, x, s->world, rng, cnt);
                ++cnt.samples;
            }
            accum[3 * p] += color.x;
            accum[3 * p + 1] += color.y;
            accum[3 * p + 2] += color.z;
        }
#pragma omp critical
        {
            total.segments += cnt.segments;
            total.box_tests += cnt.box_tests;
            total.prim_tests += cnt.prim_tests;
            total.samples += cnt.samples;
        }
    }
    if (counters) {
        counters[0] = total.segments; counters[1] = total.box_tests;
        counters[2] = total.prim_tests; counters[3] = total.samples;
    }
    return 0;
}
extern "C" int orc_trace_sample(orc_scene* s, const orc_camera* c, uint64_t seed, uint32_t pixel, uint32_t sample,
                                double radiance[3], double* dump, uint32_t max_rec) {
    if (!s->built) return fail("world not built");
    Camera cam;
    if (make_camera(s, c, cam) != 0) return -1;
    if (pixel >= cam.image_width * cam.image_height) return fail("pixel out of range");
    Counters cnt;
    Rng rng(seed, pixel, sample);
    std::vector<Camera::PathRecord> rec;
    V3 r = cam.trace(pixel / cam.image_width, pixel % cam.image_width, s->world, rng, cnt, dump ? &rec : nullptr);
    radiance[0] = r.x; radiance[1] = r.y; radiance[2] = r.z;
    for (size_t i = 0; dump && i < rec.size() && i < max_rec; ++i) {
        double* d = dump + 8 * i;
        d[0] = rec[i].t; d[1] = (double)rec[i].prim_id;
        d[2] = rec[i].point.x; d[3] = rec[i].point.y; d[4] = rec[i].point.z;
        d[5] = rec[i].throughput.x; d[6] = rec[i].throughput.y; d[7] = rec[i].throughput.z;
    }
    return (int)cnt.segments;
}
extern "C" void orc_resolve_u8(const double* accum, uint32_t n_pixels, uint32_t total_spp, uint8_t* rgb8) {
    double scale = 1.0 / (double)total_spp;  // camera.rs:53,109
    for (size_t i = 0; i < (size_t)n_pixels * 3; ++i) {
        double c = accum[i] * scale;
        double g = std::sqrt(fmax2(c, 0.0));                       // gamma_correct :128-130
        double q = clampd(g, 0.0, 0.999) * 256.0;                  // :111-113
        rgb8[i] = std::isnan(q) ? 0 : (uint8_t)q;                  // `as u8` (NaN -> 0)
    }
}

// ------------------------------------------------------------------------------ probes
extern "C" void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    Philox4 o = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, o.v, sizeof o.v);
}
extern "C" double orc_rng_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw) {
    Rng rng(seed, pixel, sample);
    rng.draw = draw;
    return rng.gen();
}
extern "C" double orc_probe(int which, const double* a) {
    switch (which) {
    case 0: return ggx::D(V3{0, 0, a[0]}, a[1]);
    case 1: return ggx::G1(V3{0, 0, a[0]}, a[1]);
    case 2: return gtr1::D(a[0], a[1]);
    case 3: return fresnel_dielectric(V3{0, 0, a[0]}, V3{0, 0, 1}, a[1], a[2]);
    case 4: {  // principled lobe probability a[3] for (metallic, spec_trans, clearcoat)
        PrincipledBSDF p;
        p.metallic = a[0]; p.spec_trans = a[1]; p.clearcoat = a[2];
        double w[4], q[4];
        p.lobe_weights(w);
        p.lobe_probabilities(w, q);
        return q[(int)a[3]];
    }
    case 5: {  // alpha_g(clearcoat_gloss)
        PrincipledBSDF p;
        p.clearcoat_gloss = a[0];
        return p.alpha_g();
    }
    case 6: {  // to_world(n, to_local(n, x)) round trip error
        V3 n = normalize(V3{a[0], a[1], a[2]}), x{a[3], a[4], a[5]};
        V3 y = to_world(n, to_local(n, x));
        return length(y - x);
    }
    case 7: {  // |to_local(n, n) - z|
        V3 n = normalize(V3{a[0], a[1], a[2]});
        return length(to_local(n, n) - V3{0, 0, 1});
    }
    case 8: return r0_of(a[0]);
    case 9: return schlick_weight(a[0]);
    case 10: {  // rigid inverse identity: |M^-1 (M p) - p|
        Rigid m = rigid_from_rotation_translation(quat_from_axis_angle(normalize(V3{a[0], a[1], a[2]}), a[3]), V3{a[4], a[5], a[6]});
        V3 p{a[7], a[8], a[9]};
        V3 q = xform_point(m.i0, m.i1, m.i2, m.it, xform_point(m.c0, m.c1, m.c2, m.t, p));
        return length(q - p);
    }
    case 11: {  // |reflect| preserved, refract TIR -> 0
        V3 i = normalize(V3{a[0], a[1], a[2]}), n{0, 0, 1};
        return a[3] == 0.0 ? length(reflect(i, n)) : length(refract(i, n, a[3]));
    }
    }
    return std::nan("");
}
extern "C" int orc_intersect(orc_scene* s, const double o[3], const double d[3], double time, double out[15]) {
    if (!s->built) return fail("world not built");
    Counters cnt;
    Ray ray(V(o), V(d), time);
    HitInfo h;
    memset(out, 0, 15 * sizeof(double));
    if (!s->world.intersect_all(ray, Interval{1e-3, INF}, h, cnt)) return 0;
    out[0] = 1.0; out[1] = h.dist; out[2] = (double)h.prim_id; out[3] = h.u; out[4] = h.v;
    out[5] = h.front_face ? 1.0 : 0.0;
    out[6] = h.point.x; out[7] = h.point.y; out[8] = h.point.z;
    out[9] = h.geometric_normal.x; out[10] = h.geometric_normal.y; out[11] = h.geometric_normal.z;
    out[12] = h.shading_normal.x; out[13] = h.shading_normal.y; out[14] = h.shading_normal.z;
    return 0;
}
