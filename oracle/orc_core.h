// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// f64 CPU restatement of the reference's per-pixel integrator: textures, materials,
// geometry + BVH, camera/trace. Structure follows the reference's trait objects so
// that each function can cite the file:line it restates. Two deliberate, documented
// deviations: (1) RNG (orc_rng.h); (2) exact-equal-t ties are
// resolved by a canonical, tree-independent rule — the hit with the LARGER global
// primitive id wins — where the reference's outcome depends on BVH shape
// (bvh.rs:153, list.rs:57-64). Ids are assigned at World::build: lights list first,
// then objects, in insertion order, sub-primitives in their own insertion order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "orc_math.h"
#include "orc_rng.h"

namespace orc {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double EPS = 1e-3;  // bsdf/mod.rs:19

struct Counters {
    uint64_t segments = 0;    // calls to World::intersect_all (camera.rs:179)
    uint64_t box_tests = 0;   // AABB::intersects calls
    uint64_t prim_tests = 0;  // Sphere/Quad/Triangle::intersects calls
    uint64_t samples = 0;
};

struct Ray {  // ray.rs:23-29 — the constructor normalises the direction
    V3 o, d;
    double time;
    Ray(V3 origin, V3 direction, double t) : o(origin), d(normalize(direction)), time(t) {}
    V3 at(double t) const { return o + d * t; }
};

struct Interval {  // interval.rs:2-41
    double min, max;
    // `cull`: distance of the best hit found so far in an enclosing leaf/list loop. The
    // reference shrinks ray_t.max itself there (list.rs:59, bvh.rs:131); the oracle keeps
    // [min,max] intact (so open/closed end-point semantics never depend on visit order)
    // and only drops candidates STRICTLY farther than `cull`; equal-t candidates reach the
    // canonical tie rule.
    double cull = std::numeric_limits<double>::infinity();
    bool contains(double x) const { return min <= x && x <= max; }   // closed
    bool surrounds(double x) const { return min < x && x < max; }    // open
};

// ---------------------------------------------------------------- textures (texture.rs)
struct TexRGB {
    virtual ~TexRGB() = default;
    virtual V3 value(double u, double v, V3 p) const = 0;
};
struct TexF {
    virtual ~TexF() = default;
    virtual double value(double u, double v, V3 p) const = 0;
};
struct SolidRGB : TexRGB {  // texture.rs:11-25
    V3 c;
    explicit SolidRGB(V3 c_) : c(c_) {}
    V3 value(double, double, V3) const override { return c; }
};
struct SolidF : TexF {
    double c;
    explicit SolidF(double c_) : c(c_) {}
    double value(double, double, V3) const override { return c; }
};
// Rust `f64 as i32` saturates, NaN -> 0.
inline int32_t f64_as_i32(double x) {
    if (std::isnan(x)) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
inline uint32_t f64_as_u32(double x) {
    if (std::isnan(x) || x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}
inline bool checker_is_first(double inv_scale, V3 p) {  // texture.rs:43-49
    int32_t x = f64_as_i32(std::floor(p.x * inv_scale));
    int32_t y = f64_as_i32(std::floor(p.y * inv_scale));
    int32_t z = f64_as_i32(std::floor(p.z * inv_scale));
    int32_t s = (int32_t)((uint32_t)x + (uint32_t)y + (uint32_t)z);  // release build wraps
    return s % 2 == 0;
}
struct CheckerRGB : TexRGB {  // texture.rs:27-54
    double inv_scale;
    std::shared_ptr<TexRGB> t1, t2;
    CheckerRGB(double scale, std::shared_ptr<TexRGB> a, std::shared_ptr<TexRGB> b)
        : inv_scale(1.0 / scale), t1(a), t2(b) {}
    V3 value(double u, double v, V3 p) const override {
        return checker_is_first(inv_scale, p) ? t1->value(u, v, p) : t2->value(u, v, p);
    }
};
struct ImageRGB8 : TexRGB {  // texture.rs:56-92 — always RGB8, nearest texel
    uint32_t w, h;
    std::vector<uint8_t> px;
    // The build's float-HDR option ("native float HDR envmaps, bypassing the to_rgb8 squash texture.rs:67,
    // behind a flag"): the image keeps the decoder's f32 samples — `ImageReader::decode()` of a Radiance file yields Rgb32F — and
    // the lookup of texture.rs:73-91 returns them widened to f64 instead of byte / 255. Same texel addressing.
    std::vector<float> pf;
    ImageRGB8(uint32_t w_, uint32_t h_, const uint8_t* data) : w(w_), h(h_), px(data, data + (size_t)w_ * h_ * 3) {}
    ImageRGB8(uint32_t w_, uint32_t h_, const float* data) : w(w_), h(h_), pf(data, data + (size_t)w_ * h_ * 3) {}
    V3 value(double u, double v, V3) const override {
        if (h == 0) return V3{0.0, 1.0, 1.0};
        u = clampd(u, 0.0, 1.0);
        v = 1.0 - clampd(v, 0.0, 1.0);
        uint32_t i = f64_as_u32(u * (double)w);
        uint32_t j = f64_as_u32(v * (double)h);
        // Q6: the reference panics for i==w / j==h (u==1 or v==0); the build clamps.
        if (i > w - 1) i = w - 1;
        if (j > h - 1) j = h - 1;
        if (!pf.empty()) {
            const float* q = &pf[((size_t)j * w + i) * 3];
            return V3{(double)q[0], (double)q[1], (double)q[2]};
        }
        const uint8_t* p = &px[((size_t)j * w + i) * 3];
        const double s = 1.0 / 255.0;
        return V3{s * (double)p[0], s * (double)p[1], s * (double)p[2]};
    }
};

// ------------------------------------------------------------------------- hit record
struct Material;
struct HitInfo {  // hit_info.rs:4-13 (+ prim id for the canonical tie rule)
    V3 point, geometric_normal, shading_normal;
    double dist;
    bool front_face;
    const Material* mat;
    double u, v;
    uint32_t prim_id;
};
inline bool better_hit(double t, uint32_t id, double bt, uint32_t bid) {
    return t < bt || (t == bt && id > bid);
}

// ------------------------------------------------------------------ materials (bsdf/*)
struct Material {  // trait BxDFMaterial, bsdf/mod.rs:21-57
    virtual ~Material() = default;
    virtual bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const = 0;
    virtual double pdf(V3 view_dir, V3 light_dir, const HitInfo& info) const = 0;
    virtual V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const = 0;
    virtual V3 emitted(double, double, V3) const { return V3{0, 0, 0}; }
    virtual const ImageRGB8* normal_map() const { return nullptr; }
};

inline V3 tint(V3 base) {  // bsdf/mod.rs:61-68
    double l = luminance(base);
    return l > 0.0 ? base / l : V3{1.0, 1.0, 1.0};
}
inline double r0_of(double eta) { return powi2((eta - 1.0) / (eta + 1.0)); }  // mod.rs:70
inline double fresnel_dielectric(V3 w, V3 h, double eta_i, double eta_o) {    // mod.rs:77-88, glass.rs:51-62
    double c = std::fabs(dot(w, h));
    double g_squared = powi2(eta_o / eta_i) - 1.0 + c * c;
    if (g_squared < 0.0) return 1.0;
    double g = std::sqrt(g_squared);
    double gmc = g - c, gpc = g + c;
    double x = (c * gpc - 1.0) / (c * gmc + 1.0);
    return 0.5 * (gmc * gmc) / (gpc * gpc) * (1.0 + x * x);
}
inline V3 fresnel_schlick(V3 r0, double angle) {  // mod.rs:90-92, metal.rs:111-113
    return r0 + (1.0 - r0) * powi5(1.0 - angle);
}
inline double schlick_weight(double x) { return powi5(clampd(1.0 - x, 0.0, 1.0)); }  // mod.rs:94-96

inline V3 cosine_sample_hemisphere(Rng& rng) {  // sampling.rs:18-24 (phi first, closed range)
    double phi = rng.gen_range_inclusive(2.0 * PI);
    double r2 = rng.gen();
    double r2s = std::sqrt(r2);
    return V3{r2s * m_cos(phi), r2s * m_sin(phi), std::sqrt(1.0 - r2)};
}
namespace ggx {  // sampling.rs:30-117
inline double D(V3 h, double roughness) {
    double cos_theta = fmax2(h.z, 0.001);
    double alpha2 = fmax2(roughness * roughness, 0.001);
    double denom = (alpha2 - 1.0) * (cos_theta * cos_theta) + 1.0;
    return alpha2 / (PI * denom * denom);
}
inline double G1(V3 w, double roughness) {
    double alpha2 = fmax2(roughness * roughness, 0.001);
    double cos_theta = std::fabs(w.z);
    return 2.0 * cos_theta / (cos_theta + std::sqrt(cos_theta * cos_theta * (1.0 - alpha2) + alpha2));
}
inline double G(V3 v, V3 l, double roughness) { return G1(v, roughness) * G1(l, roughness); }
inline V3 sample_ggx_vndf(V3 v_in, double a2, Rng& rng) {  // sampling.rs:65-94
    V3 v = normalize(V3{v_in.x * a2, v_in.y * a2, v_in.z});
    V3 t1 = v.z < 0.9999 ? normalize(cross(v, V3{0.0, 0.0, 1.0})) : V3{1.0, 0.0, 0.0};
    V3 t2 = cross(t1, v);
    double e1 = rng.gen();
    double e2 = rng.gen();
    double a = 1.0 / (1.0 + v.z);
    double r = std::sqrt(e1);
    double phi = e2 < a ? e2 / a * PI : PI + (e2 - a) / (1.0 - a) * PI;
    double p1 = r * m_cos(phi);
    double p2 = r * m_sin(phi) * (e2 < a ? 1.0 : v.z);
    V3 n = p1 * t1 + p2 * t2 + std::sqrt(fmax2(1.0 - p1 * p1 - p2 * p2, 0.0)) * v;
    return normalize(V3{a2 * n.x, a2 * n.y, fmax2(n.z, 0.0)});
}
inline V3 sample_microfacet_normal(V3 v, double roughness, Rng& rng) {  // sampling.rs:57-63 (Q2)
    V3 h = sample_ggx_vndf(v, roughness * roughness, rng);
    return h.z < 0.0 ? -h : h;
}
}  // namespace ggx
namespace gtr1 {  // sampling.rs:113-143 (Q3: log2, caller passes |l.h|)
inline double D(double abs_cos_theta, double alpha_g) {
    double alpha2 = alpha_g * alpha_g;
    double t = 1.0 + (alpha2 - 1.0) * abs_cos_theta * abs_cos_theta;
    return (alpha2 - 1.0) / (PI * t * m_log2(alpha2));
}
inline V3 sample_microfacet_normal(double alpha, Rng& rng) {
    double e1 = rng.gen();
    double e2 = rng.gen();
    double alpha2 = alpha * alpha;
    double cos_theta = (1.0 - m_pow(alpha2, 1.0 - e1)) / (1.0 - alpha2);
    double sin_theta = std::sqrt(fmax2(1.0 - cos_theta * cos_theta, 0.0));
    double phi = 2.0 * PI * e2;
    V3 h{sin_theta * m_cos(phi), sin_theta * m_sin(phi), cos_theta};
    return h.z < 0.0 ? -h : h;
}
}  // namespace gtr1

struct DiffuseBRDF : Material {  // diffuse.rs:50-84 (uses shading_normal)
    std::shared_ptr<TexRGB> base_color;
    std::shared_ptr<ImageRGB8> nmap;
    bool sample(const Ray&, const HitInfo& info, Rng& rng, V3& out) const override {
        V3 l = cosine_sample_hemisphere(rng);
        out = to_world(info.shading_normal, l);
        return true;
    }
    double pdf(V3, V3 light_dir, const HitInfo& info) const override {
        V3 l = to_local(info.shading_normal, light_dir);
        return std::fabs(l.z) / PI;
    }
    V3 eval(V3, V3 light_dir, const HitInfo& info) const override {
        V3 color = base_color->value(info.u, info.v, info.point);
        V3 l = to_local(info.shading_normal, light_dir);
        return std::fabs(l.z) * (color / PI);
    }
    const ImageRGB8* normal_map() const override { return nmap.get(); }
};

struct MetalBRDF : Material {  // metal.rs:38-80
    std::shared_ptr<TexRGB> base_color;
    std::shared_ptr<TexF> roughness;
    bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const override {
        V3 v = to_local(info.shading_normal, -ray.d);
        double rough = roughness->value(info.u, info.v, info.point);
        V3 h = ggx::sample_microfacet_normal(v, rough, rng);
        V3 dir = to_world(info.shading_normal, reflect(-v, h));
        if (dot(dir, info.shading_normal) <= 0.0) return false;
        out = dir;
        return true;
    }
    double pdf(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        V3 h = normalize(v + l);
        double rough = roughness->value(info.u, info.v, info.point);
        double pdf_h = ggx::G1(v, rough) * std::fabs(dot(v, h)) * ggx::D(h, rough) / std::fabs(v.z);
        double jacobian = 1.0 / (4.0 * std::fabs(dot(l, h)));
        return pdf_h * jacobian;
    }
    V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        V3 h = normalize(v + l);
        double rough = roughness->value(info.u, info.v, info.point);
        V3 base = base_color->value(info.u, info.v, info.point);
        double d = ggx::D(h, rough);
        double g = ggx::G(v, l, rough);
        V3 f = fresnel_schlick(base, dot(l, h));
        return std::fabs(l.z) * (f * g * d / (4.0 * std::fabs(l.z) * std::fabs(v.z)));
    }
};

// Walter-2007 generalized half vector shared by glass.rs:103-107 and principled.rs:295-299
inline V3 generalized_half(V3 v, V3 l, bool is_reflect, double eta_i, double eta_o) {
    if (is_reflect) return normalize(l + v) * signum(v.z);
    return -normalize(l * eta_o + v * eta_i);
}

struct GlassBSDF : Material {  // glass.rs:65-163 (Q4: eval ignores base_color)
    std::shared_ptr<TexRGB> base_color;
    std::shared_ptr<TexF> roughness;
    double ior;
    bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const override {
        V3 v = to_local(info.shading_normal, -ray.d);
        double rough = roughness->value(info.u, info.v, info.point);
        V3 h = ggx::sample_microfacet_normal(v, rough, rng);
        double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
        double f = fresnel_dielectric(v, h, eta_i, eta_o);
        if (rng.gen() < f) {
            out = to_world(info.shading_normal, reflect(-v, h));
        } else {
            V3 t = refract(-v, h, eta_i / eta_o);
            if (is_zero(t)) t = reflect(-v, h);
            out = to_world(info.shading_normal, t);
        }
        return true;
    }
    double pdf(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        bool is_reflect = l.z * v.z > 0.0;
        double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
        V3 h = generalized_half(v, l, is_reflect, eta_i, eta_o);
        double rough = roughness->value(info.u, info.v, info.point);
        double pdf_h = ggx::G1(v, rough) * std::fabs(dot(v, h)) * ggx::D(h, rough) / std::fabs(v.z);
        double f = fresnel_dielectric(v, h, eta_i, eta_o);
        double jacobian;
        if (is_reflect) {
            jacobian = f * 1.0 / (4.0 * std::fabs(dot(l, h)));
        } else {
            double v_dot_h = dot(v, h), l_dot_h = dot(l, h);
            jacobian = (1.0 - f) * (eta_o * eta_o * std::fabs(l_dot_h)) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
        }
        return pdf_h * jacobian;
    }
    V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        bool is_reflect = l.z * v.z > 0.0;
        double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
        V3 h = generalized_half(v, l, is_reflect, eta_i, eta_o);
        double rough = roughness->value(info.u, info.v, info.point);
        double d = ggx::D(h, rough);
        double g = ggx::G(v, l, rough);
        double f = fresnel_dielectric(v, h, eta_i, eta_o);
        double factor;
        if (is_reflect) {
            factor = f * g * d / (4.0 * std::fabs(l.z) * std::fabs(v.z));
        } else {
            double l_dot_h = dot(l, h), v_dot_h = dot(v, h);
            double term1 = std::fabs((l_dot_h * v_dot_h) / (l.z * v.z));
            double term2 = (eta_o * eta_o) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
            factor = term1 * term2 * (1.0 - f) * g * d;
        }
        return splat(factor) * std::fabs(l.z);
    }
};

struct PrincipledBSDF : Material {  // principled.rs (uses geometric_normal everywhere)
    std::shared_ptr<TexRGB> base_color;
    double metallic, roughness, subsurface, specular, specular_tint, ior, spec_trans, sheen,
        sheen_tint, clearcoat, clearcoat_gloss;

    double alpha_g() const { return (1.0 - clearcoat_gloss) * 0.1 + clearcoat_gloss * 0.001; }  // :75-77
    void lobe_weights(double w[4]) const {  // :79-85
        w[0] = (1.0 - metallic) * (1.0 - spec_trans);
        w[1] = 1.0 - spec_trans * (1.0 - metallic);
        w[2] = spec_trans * (1.0 - metallic);
        w[3] = 0.25 * clearcoat;
    }
    void lobe_probabilities(const double w[4], double p[4]) const {  // :87-100
        double inv_total = 1.0 / (w[0] + w[1] + w[2] + w[3]);
        for (int i = 0; i < 4; ++i) p[i] = w[i] * inv_total;
    }
    bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const override {  // :262-276
        double w[4], p[4];
        lobe_weights(w);
        lobe_probabilities(w, p);
        double r = rng.gen();
        V3 n = info.geometric_normal;
        if (r < p[0]) {  // sample_diffuse :102-104
            out = to_world(n, cosine_sample_hemisphere(rng));
            return true;
        } else if (r < p[0] + p[1]) {  // sample_specular :106-118
            V3 v = to_local(n, -ray.d);
            V3 h = ggx::sample_microfacet_normal(v, roughness, rng);
            V3 dir = to_world(n, reflect(-v, h));
            if (dot(dir, n) <= 0.0) return false;
            out = dir;
            return true;
        } else if (r < p[0] + p[1] + p[2]) {  // sample_glass :120-142
            V3 v = to_local(n, -ray.d);
            V3 h = ggx::sample_microfacet_normal(v, roughness, rng);
            double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
            double f = fresnel_dielectric(v, h, eta_i, eta_o);
            if (rng.gen() < f) {
                out = to_world(n, reflect(-v, h));
            } else {
                V3 t = refract(-v, h, eta_i / eta_o);
                if (is_zero(t)) t = reflect(-v, h);
                out = to_world(n, t);
            }
            return true;
        } else {  // sample_clearcoat :144-155 (sampler alpha fixed at 0.25)
            V3 v = to_local(n, -ray.d);
            V3 h = gtr1::sample_microfacet_normal(0.25, rng);
            V3 dir = to_world(n, reflect(-v, h));
            if (dot(dir, n) <= 0.0) return false;
            out = dir;
            return true;
        }
    }
    double specular_pdf(V3 v, V3 l, V3 h) const {  // :161-168
        double pdf_h = ggx::G1(v, roughness) * std::fabs(dot(v, h)) * ggx::D(h, roughness) / std::fabs(v.z);
        double jacobian = 1.0 / (4.0 * std::fabs(dot(l, h)));
        return pdf_h * jacobian;
    }
    double glass_pdf(V3 v, V3 l, V3 h, double eta_i, double eta_o, bool is_reflect) const {  // :170-185
        double pdf_h = ggx::G1(v, roughness) * std::fabs(dot(v, h)) * ggx::D(h, roughness) / std::fabs(v.z);
        double f = fresnel_dielectric(v, h, eta_i, eta_o);
        double jacobian;
        if (is_reflect) {
            jacobian = f * 1.0 / (4.0 * std::fabs(dot(l, h)));
        } else {
            double v_dot_h = dot(v, h), l_dot_h = dot(l, h);
            jacobian = (1.0 - f) * (eta_o * eta_o * std::fabs(l_dot_h)) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
        }
        return pdf_h * jacobian;
    }
    double clearcoat_pdf(V3 v, V3 l, V3 h) const {  // :187-192
        double pdf_h = ggx::G1(v, 0.25) * std::fabs(dot(v, h)) * gtr1::D(std::fabs(dot(l, h)), alpha_g()) / std::fabs(v.z);
        double jacobian = 1.0 / (4.0 * std::fabs(dot(l, h)));
        return pdf_h * jacobian;
    }
    double pdf(V3 view_dir, V3 light_dir, const HitInfo& info) const override {  // :278-315
        double w[4], p[4];
        lobe_weights(w);
        lobe_probabilities(w, p);
        V3 v = to_local(info.geometric_normal, view_dir);
        V3 l = to_local(info.geometric_normal, light_dir);
        bool is_reflect = l.z * v.z > 0.0;
        double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
        V3 h = generalized_half(v, l, is_reflect, eta_i, eta_o);
        double pdf = 0.0;
        if (p[0] > 0.0 && is_reflect) pdf += p[0] * (std::fabs(l.z) / PI);  // diffuse_pdf :157-159
        if (p[1] > 0.0 && is_reflect) pdf += p[1] * specular_pdf(v, l, h);
        if (p[2] > 0.0) pdf += p[2] * glass_pdf(v, l, h, eta_i, eta_o, is_reflect);
        if (p[3] > 0.0 && is_reflect) pdf += p[3] * clearcoat_pdf(v, l, h);
        return pdf;
    }
    V3 eval_diffuse(V3 color, V3 v, V3 l, V3 h) const {  // :196-214
        double l_dot_h = dot(l, h);
        double rr = 2.0 * roughness * l_dot_h * l_dot_h;
        double fl = schlick_weight(l.z), fv = schlick_weight(v.z);
        double f_retro = rr * (fl + fv + fl * fv * (rr - 1.0));
        double f_d = (1.0 - 0.5 * fl) * (1.0 - 0.5 * fv);
        double fss90 = 0.5 * rr;
        double f_ss = flerp(1.0, fss90, fl) * flerp(1.0, fss90, fv);
        double ss = 1.25 * (f_ss * (1.0 / (l.z + v.z) - 0.5) + 0.5);
        return color / PI * flerp(f_d + f_retro, ss, subsurface);
    }
    V3 eval_specular(V3 fresnel, V3 v, V3 l, V3 h) const {  // :216-225
        double d = ggx::D(h, roughness);
        double g = ggx::G(v, l, roughness);
        return fresnel * g * d / (4.0 * std::fabs(l.z) * std::fabs(v.z));
    }
    V3 eval_glass(V3 v, V3 l, V3 h, double eta_i, double eta_o, bool is_reflect) const {  // :227-246
        double d = ggx::D(h, roughness);
        double g = ggx::G(v, l, roughness);
        double f = fresnel_dielectric(v, h, eta_i, eta_o);
        if (is_reflect) return splat(f * g * d / (4.0 * std::fabs(l.z) * std::fabs(v.z)));
        double l_dot_h = dot(l, h), v_dot_h = dot(v, h);
        double term1 = std::fabs((l_dot_h * v_dot_h) / (l.z * v.z));
        double term2 = (eta_o * eta_o) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
        return splat(term1 * term2 * (1.0 - f) * g * d);
    }
    V3 eval_clearcoat(V3 v, V3 l, V3 h) const {  // :248-258
        double d = gtr1::D(std::fabs(dot(l, h)), alpha_g());
        double g = ggx::G(v, l, 0.25);
        V3 f = fresnel_schlick(splat(r0_of(1.5)), dot(l, h));
        return std::fabs(l.z) * (f * d * g / (4.0 * std::fabs(l.z) * std::fabs(v.z)));
    }
    V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const override {  // :317-366
        V3 base = base_color->value(info.u, info.v, info.point);
        double w[4], p[4];
        lobe_weights(w);
        lobe_probabilities(w, p);
        V3 v = to_local(info.geometric_normal, view_dir);
        V3 l = to_local(info.geometric_normal, light_dir);
        bool is_reflect = l.z * v.z > 0.0;
        double eta_i = info.front_face ? 1.0 : ior, eta_o = info.front_face ? ior : 1.0;
        V3 h = generalized_half(v, l, is_reflect, eta_i, eta_o);
        V3 brdf{0, 0, 0};
        if (p[0] > 0.0 && is_reflect) {
            V3 c_tint = tint(base);
            V3 c_sheen = vlerp(splat(1.0), c_tint, sheen_tint);
            V3 sheen_term = sheen * c_sheen * schlick_weight(std::fabs(dot(l, h)));
            V3 diffuse_term = eval_diffuse(base, v, l, h);
            brdf += w[0] * (diffuse_term + sheen_term);
        }
        if (p[1] > 0.0 && is_reflect) {
            V3 c_tint = tint(base);
            V3 ks = vlerp(splat(1.0), c_tint, specular_tint);
            V3 c0 = vlerp(specular * r0_of(eta_i / eta_o) * ks, base, metallic);
            V3 metallic_fresnel = fresnel_schlick(c0, dot(l, h));
            V3 dielectric = splat(fresnel_dielectric(v, h, eta_i, eta_o));
            V3 fresnel = vlerp(dielectric, metallic_fresnel, metallic);
            brdf += w[1] * eval_specular(fresnel, v, l, h);
        }
        if (p[2] > 0.0) brdf += w[2] * eval_glass(v, l, h, eta_i, eta_o, is_reflect);
        if (p[3] > 0.0 && is_reflect) brdf += w[3] * eval_clearcoat(v, l, h);
        return brdf * std::fabs(l.z);
    }
};

struct MixBxDf : Material {  // mix.rs (never instantiated by the reference's scenes; part of bsdf/)
    double t;
    std::shared_ptr<Material> a, b;
    bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const override {
        double p = rng.gen();
        return t < p ? a->sample(ray, info, rng, out) : b->sample(ray, info, rng, out);
    }
    double pdf(V3 v, V3 l, const HitInfo& info) const override {
        double p1 = (1.0 - t) * a->pdf(v, l, info);
        double p2 = t * b->pdf(v, l, info);
        return p1 + p2;
    }
    V3 eval(V3 v, V3 l, const HitInfo& info) const override {
        V3 w1 = (1.0 - t) * a->eval(v, l, info);
        V3 w2 = t * b->eval(v, l, info);
        return w1 + w2;
    }
};
struct SheenBRDF : Material {  // sheen.rs (geometric normal, colour is a plain Vec3)
    V3 base_color;
    double sheen_tint;
    bool sample(const Ray&, const HitInfo& info, Rng& rng, V3& out) const override {
        out = to_world(info.geometric_normal, cosine_sample_hemisphere(rng));
        return true;
    }
    double pdf(V3, V3 light_dir, const HitInfo& info) const override {
        V3 l = to_local(info.geometric_normal, light_dir);
        return std::fabs(l.z) / PI;
    }
    V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.geometric_normal, view_dir);
        V3 l = to_local(info.geometric_normal, light_dir);
        V3 h = normalize(v + l);
        V3 c_sheen = vlerp(splat(1.0), tint(base_color), sheen_tint);
        return c_sheen * powi5(1.0 - std::fabs(dot(l, h))) * std::fabs(l.z);
    }
};
struct ClearcoatBRDF : Material {  // clearcoat.rs (shading normal)
    double alpha_g;
    bool sample(const Ray& ray, const HitInfo& info, Rng& rng, V3& out) const override {
        V3 v = to_local(info.shading_normal, -ray.d);
        V3 h = gtr1::sample_microfacet_normal(0.25, rng);
        V3 dir = to_world(info.shading_normal, reflect(-v, h));
        if (dot(dir, info.shading_normal) <= 0.0) return false;
        out = dir;
        return true;
    }
    double pdf(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        V3 h = normalize(v + l);
        double pdf_h = ggx::G1(v, 0.25) * std::fabs(dot(v, h)) * gtr1::D(std::fabs(dot(l, h)), alpha_g) / std::fabs(v.z);
        double jacobian = 1.0 / (4.0 * std::fabs(dot(l, h)));
        return pdf_h * jacobian;
    }
    V3 eval(V3 view_dir, V3 light_dir, const HitInfo& info) const override {
        V3 v = to_local(info.shading_normal, view_dir);
        V3 l = to_local(info.shading_normal, light_dir);
        V3 h = normalize(v + l);
        double d = gtr1::D(std::fabs(dot(l, h)), alpha_g);
        double g = ggx::G(v, l, 0.25);
        V3 f = fresnel_schlick(splat(r0_of(1.5)), dot(l, h));
        return std::fabs(l.z) * (f * d * g / (4.0 * std::fabs(l.z) * std::fabs(v.z)));
    }
};

struct DiffuseLight : Material {  // material.rs:150-191 (Q5)
    std::shared_ptr<TexRGB> emission;
    bool sample(const Ray&, const HitInfo&, Rng&, V3&) const override { return false; }
    double pdf(V3, V3, const HitInfo&) const override { return 1.0; }
    V3 eval(V3, V3, const HitInfo&) const override { return V3{1.0, 1.0, 1.0}; }
    V3 emitted(double u, double v, V3 p) const override { return emission->value(u, v, p); }
};

// hit_info.rs:16-67
inline void tangent_basis(V3 n, V3& tangent, V3& bitangent) {
    V3 a = std::fabs(n.x) > 0.9 ? V3{0.0, 1.0, 0.0} : V3{1.0, 0.0, 0.0};
    tangent = normalize(cross(n, a));
    bitangent = cross(n, tangent);
}
inline HitInfo make_hit(const Ray& ray, V3 point, V3 normal, double dist, const Material* mat,
                        double u, double v, uint32_t prim_id) {
    HitInfo h;
    h.front_face = dot(ray.d, normal) < 0.0;
    h.geometric_normal = h.front_face ? normalize(normal) : -normalize(normal);
    if (const ImageRGB8* nm = mat->normal_map()) {
        V3 m = 2.0 * nm->value(u, v, point) - splat(1.0);
        V3 t, b;
        tangent_basis(h.geometric_normal, t, b);
        h.shading_normal = normalize(m.x * t + m.y * b + m.z * h.geometric_normal);
    } else {
        h.shading_normal = h.geometric_normal;
    }
    h.point = point;
    h.dist = dist;
    h.mat = mat;
    h.u = u;
    h.v = v;
    h.prim_id = prim_id;
    return h;
}

// --------------------------------------------------------------- geometry (hittable/*)
struct AABB {  // aabb.rs — every constructor pads by 1e-3 (and union re-pads)
    V3 mn{INF, INF, INF}, mx{-INF, -INF, -INF};
    static AABB make(V3 a, V3 b) {
        AABB r;
        r.mn = vmin(a, b) - splat(1e-3);
        r.mx = vmax(a, b) + splat(1e-3);
        return r;
    }
    static AABB unite(AABB a, AABB b) { return make(vmin(a.mn, b.mn), vmax(a.mx, b.mx)); }
    V3 centroid() const { return 0.5 * (mn + mx); }
    double surface_area() const {  // half area, aabb.rs:48-52
        V3 e = mx - mn;
        return e.x * e.y + e.x * e.z + e.y * e.z;
    }
    bool intersects(const Ray& ray, Interval ray_t, Counters& c) const {  // aabb.rs:31-42
        ++c.box_tests;
        V3 m{1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z};
        V3 t1 = (mn - ray.o) * m;
        V3 t2 = (mx - ray.o) * m;
        double t_near = max_element(vmin(t1, t2));
        double t_far = min_element(vmax(t1, t2));
        return t_near <= t_far && t_far >= ray_t.min && t_near <= ray_t.max;
    }
    AABB transformed(const Rigid& m) const {  // aabb.rs:54-79
        V3 c[8] = {mn, {mn.x, mn.y, mx.z}, {mn.x, mx.y, mn.z}, {mn.x, mx.y, mx.z},
                   {mx.x, mn.y, mn.z}, {mx.x, mn.y, mx.z}, {mx.x, mx.y, mn.z}, mx};
        V3 lo = splat(INF), hi = splat(-INF);
        for (auto& p : c) {
            V3 q = xform_point(m.c0, m.c1, m.c2, m.t, p);
            lo = vmin(lo, q);
            hi = vmax(hi, q);
        }
        return make(lo, hi);
    }
};

struct Hittable {  // trait Hittable, hittable/mod.rs:38-48
    virtual ~Hittable() = default;
    // `best`: t/id of the best hit found so far along this ray (canonical tie rule); a
    // candidate is returned only if it is inside ray_t AND better than `best`.
    virtual bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const = 0;
    virtual AABB bounding_box() const = 0;
    virtual bool sample(V3 origin, double time, Rng& rng, V3& dir) const = 0;
    virtual double pdf(V3 origin, V3 direction, double time, Counters& c) const = 0;
    virtual uint32_t assign_ids(uint32_t first) = 0;  // returns next free id
    virtual uint32_t prim_count() const = 0;
};
using HitPtr = std::shared_ptr<Hittable>;

struct Sphere : Hittable {  // sphere.rs
    double radius;
    V3 p1, p2;
    std::shared_ptr<Material> mat;
    AABB bbox;
    uint32_t id = 0;
    Sphere(double r, V3 a, V3 b, std::shared_ptr<Material> m) : radius(fmax2(r, 0.0)), p1(a), p2(b), mat(m) {
        V3 rv = splat(r);
        bbox = AABB::unite(AABB::make(a - rv, a + rv), AABB::make(b - rv, b + rv));
        if (a.x == b.x && a.y == b.y && a.z == b.z) bbox = AABB::make(a - rv, a + rv);  // new_still :22-32
    }
    V3 position(double t) const { return p1 + (p2 - p1) * t; }  // :58-60
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {  // :64-100
        ++c.prim_tests;
        V3 center = position(ray.time);
        V3 l = center - ray.o;
        double s = dot(l, ray.d);
        double l2 = length_squared(l);
        double r2 = radius * radius;
        if (s < 0.0 && l2 > r2) return false;
        double d2 = l2 - s * s;
        if (d2 > r2) return false;
        double q = std::sqrt(r2 - d2);
        double t = l2 > r2 ? s - q : s + q;
        if (t <= ray_t.min || t >= ray_t.max) return false;  // open interval
        if (t > ray_t.cull) return false;
        V3 point = ray.at(t);
        V3 normal = normalize(point - center);
        double theta = m_acos(-normal.y);                  // get_uv :52-56
        double phi = m_atan2(-normal.z, normal.x) + PI;
        out = make_hit(ray, point, normal, t, mat.get(), phi / (2.0 * PI), theta / PI, id);
        return true;
    }
    AABB bounding_box() const override { return bbox; }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override {  // :110-122
        double u = rng.gen(), v = rng.gen();
        double theta = 2.0 * PI * u;
        double phi = m_acos(2.0 * v - 1.0);
        double x = m_sin(phi) * m_cos(theta), y = m_sin(phi) * m_sin(theta), z = m_cos(phi);
        V3 point = position(time) + V3{x, y, z} * radius;
        dir = normalize(point - origin);
        return true;
    }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override {  // :124-135
        HitInfo h;
        if (!intersects(Ray(origin, direction, time), Interval{0.0, INF}, h, c)) return 0.0;
        double r2 = radius * radius;
        double solid_angle = 2.0 * PI * std::sqrt(1.0 - r2 / length_squared(position(time) - origin));
        return 1.0 / solid_angle;
    }
    uint32_t assign_ids(uint32_t first) override { id = first; return first + 1; }
    uint32_t prim_count() const override { return 1; }
};

struct Quad : Hittable {  // quad.rs
    V3 q, u, v, w, normal;
    double d;
    AABB bbox;
    std::shared_ptr<Material> mat;
    uint32_t id = 0;
    Quad(V3 q_, V3 u_, V3 v_, std::shared_ptr<Material> m) : q(q_), u(u_), v(v_), mat(m) {  // :17-36
        bbox = AABB::unite(AABB::make(q, q + u + v), AABB::make(q + u, q + v));
        V3 n = cross(u, v);
        normal = normalize(n);
        d = dot(normal, q);
        w = n / length_squared(n);
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {  // :40-70
        ++c.prim_tests;
        double nd = dot(normal, ray.d);
        if (std::fabs(nd) < 1e-8) return false;
        double t = (d - dot(normal, ray.o)) / nd;
        if (!ray_t.contains(t)) return false;  // closed interval
        if (t > ray_t.cull) return false;
        V3 p = ray.at(t) - q;
        double alpha = dot(w, cross(p, v));
        double beta = dot(w, cross(u, p));
        if (!(alpha >= 0.0 && alpha <= 1.0) || !(beta >= 0.0 && beta <= 1.0)) return false;
        out = make_hit(ray, ray.at(t), normal, t, mat.get(), alpha, beta, id);
        return true;
    }
    AABB bounding_box() const override { return bbox; }
    bool sample(V3 origin, double, Rng& rng, V3& dir) const override {  // :80-86
        double a = rng.gen(), b = rng.gen();
        V3 point = q + u * a + v * b;
        dir = normalize(point - origin);
        return true;
    }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override {  // :88-98
        Ray ray(origin, direction, time);
        HitInfo h;
        if (!intersects(ray, Interval{0.0, INF}, h, c)) return 0.0;
        double area = length(cross(u, v));
        double cos_theta = std::fabs(dot(ray.d, h.shading_normal));
        return (h.dist * h.dist) / (cos_theta * area);
    }
    uint32_t assign_ids(uint32_t first) override { id = first; return first + 1; }
    uint32_t prim_count() const override { return 1; }
};

struct Triangle : Hittable {  // mesh.rs:13-141
    V3 vtx[3];
    bool has_n = false, has_uv = false;
    V3 nrm[3];
    double uvs[3][2];
    std::shared_ptr<Material> mat;
    AABB bbox;
    uint32_t id = 0;
    Triangle(V3 a, V3 b, V3 c, std::shared_ptr<Material> m) : mat(m) {
        vtx[0] = a; vtx[1] = b; vtx[2] = c;
        bbox = AABB::make(vmin(vmin(a, b), c), vmax(vmax(a, b), c));
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {  // :50-112
        ++c.prim_tests;
        V3 v0 = vtx[0];
        V3 edge1 = vtx[1] - v0, edge2 = vtx[2] - v0;
        V3 h = cross(ray.d, edge2);
        double a = dot(edge1, h);
        if (std::fabs(a) < 1e-8) return false;
        double f = 1.0 / a;
        V3 s = ray.o - v0;
        double u = f * dot(s, h);
        if (!(u >= 0.0 && u <= 1.0)) return false;
        V3 q = cross(s, edge1);
        double v = f * dot(ray.d, q);
        if (v < 0.0 || u + v > 1.0) return false;
        double t = f * dot(edge2, q);
        if (!ray_t.contains(t)) return false;  // closed interval
        if (t > ray_t.cull) return false;
        double w = 1.0 - u - v;
        V3 normal = has_n ? normalize(nrm[0] * w + nrm[1] * u + nrm[2] * v) : normalize(cross(edge1, edge2));
        double tu = u, tv = v;
        if (has_uv) {
            tu = uvs[0][0] * w + uvs[1][0] * u + uvs[2][0] * v;
            tv = uvs[0][1] * w + uvs[1][1] * u + uvs[2][1] * v;
        }
        out = make_hit(ray, ray.at(t), normal, t, mat.get(), tu, tv, id);
        return true;
    }
    AABB bounding_box() const override { return bbox; }
    double area() const { return 0.5 * length(cross(vtx[1] - vtx[0], vtx[2] - vtx[0])); }
    bool sample(V3 origin, double, Rng& rng, V3& dir) const override {  // :122-129
        double u = rng.gen(), v = rng.gen();
        double w = 1.0 - u - v;
        V3 point = vtx[0] * w + vtx[1] * u + vtx[2] * v;
        dir = normalize(point - origin);
        return true;
    }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override {  // :131-141
        Ray ray(origin, direction, time);
        HitInfo h;
        if (!intersects(ray, Interval{0.0, INF}, h, c)) return 0.0;
        double cos_theta = std::fabs(dot(direction, h.shading_normal));
        return h.dist * h.dist / (cos_theta * area());
    }
    uint32_t assign_ids(uint32_t first) override { id = first; return first + 1; }
    uint32_t prim_count() const override { return 1; }
};

struct BVHNode {  // bvh.rs:6-16
    AABB bbox;
    std::vector<HitPtr> leaf;  // non-empty for leaves
    std::unique_ptr<BVHNode> left, right;
    bool is_leaf() const { return !left; }
};

struct BVH {  // bvh.rs:20-121 — full-sweep SAH, O(n^2) per node
    static AABB fold_boxes(const std::vector<HitPtr>& h) {
        AABB acc;
        for (auto& o : h) acc = AABB::unite(acc, o->bounding_box());
        return acc;
    }
    static double centroid_axis(const HitPtr& o, int axis) {
        V3 c = o->bounding_box().centroid();
        return axis == 0 ? c.x : axis == 1 ? c.y : c.z;
    }
    static double evaluate_sah(int axis, double split_pos, const AABB& parent, const std::vector<HitPtr>& h) {  // :86-120
        AABB lb, rb;
        size_t lc = 0, rc = 0;
        for (auto& o : h) {
            if (centroid_axis(o, axis) < split_pos) { lb = AABB::unite(lb, o->bounding_box()); ++lc; }
            else { rb = AABB::unite(rb, o->bounding_box()); ++rc; }
        }
        if (lc == 0 || rc == 0) return INF;
        double cost = lb.surface_area() * (double)lc + rb.surface_area() * (double)rc;
        double parent_cost = parent.surface_area() * (double)h.size();
        return (cost > 0.0 && cost < parent_cost) ? cost : INF;
    }
    static std::unique_ptr<BVHNode> build(std::vector<HitPtr> h) {  // :28-52
        auto node = std::make_unique<BVHNode>();
        if (h.size() <= 4) {
            node->bbox = fold_boxes(h);
            node->leaf = std::move(h);
            return node;
        }
        AABB parent = fold_boxes(h);
        double best_cost = INF, best_pos = 0.0;
        int best_axis = 0;
        for (int axis = 0; axis < 3; ++axis) {  // :62-76
            std::vector<double> pos(h.size());
            for (size_t i = 0; i < h.size(); ++i) pos[i] = centroid_axis(h[i], axis);
            std::stable_sort(pos.begin(), pos.end());
            // every candidate is evaluated exactly as the reference does (O(n) each); the
            // evaluations are independent, so they run in parallel and the first strict
            // minimum is then picked in the reference's order — same tree, less waiting.
            std::vector<double> cost(pos.size());
            const int64_t nc = (int64_t)pos.size();
#pragma omp parallel for schedule(static) if (nc > 512)
            for (int64_t i = 0; i < nc; ++i) cost[i] = evaluate_sah(axis, pos[i], parent, h);
            for (int64_t i = 0; i < nc; ++i)
                if (cost[i] < best_cost) { best_cost = cost[i]; best_axis = axis; best_pos = pos[i]; }
        }
        std::vector<HitPtr> l, r;
        for (auto& o : h) (centroid_axis(o, best_axis) < best_pos ? l : r).push_back(o);  // :78-81
        if (l.empty() || r.empty()) {
            node->bbox = fold_boxes(h);
            node->leaf = std::move(h);
            return node;
        }
        node->left = build(std::move(l));
        node->right = build(std::move(r));
        node->bbox = AABB::unite(node->left->bbox, node->right->bbox);
        return node;
    }
    // bvh.rs:123-164 — box re-tested on entry, both children descended when both boxes
    // are hit, no t-trimming across siblings. Tie rule: canonical (see file header).
    static bool intersects(const BVHNode& n, const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) {
        if (!n.bbox.intersects(ray, ray_t, c)) return false;
        if (n.is_leaf()) {
            bool found = false;
            double closest = ray_t.cull;
            HitInfo cand;
            for (auto& p : n.leaf) {
                if (p->intersects(ray, Interval{ray_t.min, ray_t.max, closest}, cand, c)) {
                    if (!found || better_hit(cand.dist, cand.prim_id, out.dist, out.prim_id)) {
                        out = cand;
                        found = true;
                        closest = cand.dist;
                    }
                }
            }
            return found;
        }
        bool lh = n.left->bbox.intersects(ray, ray_t, c);
        bool rh = n.right->bbox.intersects(ray, ray_t, c);
        if (!lh && !rh) return false;
        if (!lh) return intersects(*n.right, ray, ray_t, out, c);
        if (!rh) return intersects(*n.left, ray, ray_t, out, c);
        HitInfo a, b;
        bool ha = intersects(*n.left, ray, ray_t, a, c);
        bool hb = intersects(*n.right, ray, ray_t, b, c);
        if (!ha && !hb) return false;
        if (ha && (!hb || better_hit(a.dist, a.prim_id, b.dist, b.prim_id))) out = a;
        else out = b;
        return true;
    }
};

struct HittableList : Hittable {  // list.rs
    std::vector<HitPtr> objects;
    AABB bbox;
    std::unique_ptr<BVHNode> bvh;
    void add(HitPtr o) {
        bbox = AABB::unite(bbox, o->bounding_box());
        objects.push_back(o);
    }
    void build_bvh() {
        if (!objects.empty()) bvh = BVH::build(objects);
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {  // :49-68
        if (bvh) return BVH::intersects(*bvh, ray, ray_t, out, c);
        bool found = false;
        double closest = ray_t.cull;
        HitInfo cand;
        for (auto& o : objects) {
            if (o->intersects(ray, Interval{ray_t.min, ray_t.max, closest}, cand, c)) {
                if (!found || better_hit(cand.dist, cand.prim_id, out.dist, out.prim_id)) {
                    out = cand;
                    found = true;
                    closest = cand.dist;
                }
            }
        }
        return found;
    }
    AABB bounding_box() const override { return bbox; }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override {  // :78-84
        if (objects.empty()) return false;
        uint32_t i = rng.gen_index((uint32_t)objects.size());
        return objects[i]->sample(origin, time, rng, dir);
    }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override {  // :86-96
        if (objects.empty()) return 0.0;
        double sum = 0.0;
        for (auto& o : objects) sum += o->pdf(origin, direction, time, c);
        return sum / (double)objects.size();
    }
    uint32_t assign_ids(uint32_t first) override {
        for (auto& o : objects) first = o->assign_ids(first);
        return first;
    }
    uint32_t prim_count() const override {
        uint32_t n = 0;
        for (auto& o : objects) n += o->prim_count();
        return n;
    }
};

struct Cuboid : Hittable {  // cuboid.rs:11-85 — six quads in a list WITHOUT a BVH
    HittableList sides;
    Cuboid(V3 a, V3 b, std::shared_ptr<Material> m) {
        V3 mn = vmin(a, b), mx = vmax(a, b);
        V3 dx{mx.x - mn.x, 0.0, 0.0}, dy{0.0, mx.y - mn.y, 0.0}, dz{0.0, 0.0, mx.z - mn.z};
        sides.add(std::make_shared<Quad>(V3{mn.x, mn.y, mx.z}, dx, dy, m));   // front
        sides.add(std::make_shared<Quad>(V3{mx.x, mn.y, mx.z}, -dz, dy, m));  // right
        sides.add(std::make_shared<Quad>(V3{mx.x, mn.y, mn.z}, -dx, dy, m));  // back
        sides.add(std::make_shared<Quad>(V3{mn.x, mn.y, mn.z}, dz, dy, m));   // left
        sides.add(std::make_shared<Quad>(V3{mn.x, mx.y, mx.z}, dx, -dz, m));  // top
        sides.add(std::make_shared<Quad>(V3{mn.x, mn.y, mn.z}, dx, dz, m));   // bottom
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override { return sides.intersects(ray, ray_t, out, c); }
    AABB bounding_box() const override { return sides.bounding_box(); }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override { return sides.sample(origin, time, rng, dir); }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override { return sides.pdf(origin, direction, time, c); }
    uint32_t assign_ids(uint32_t first) override { return sides.assign_ids(first); }
    uint32_t prim_count() const override { return 6; }
};

struct TriangleMesh : Hittable {  // mesh.rs:144-220
    HittableList triangles;
    // from_obj :149-197 — positions are f32 (tobj) widened to f64 and scaled
    TriangleMesh(double scale, size_t n_pos, const float* pos, size_t n_idx, const uint32_t* idx,
                 size_t n_nrm, const float* nrm, size_t n_uv, const float* uv, std::shared_ptr<Material> m) {
        std::vector<V3> vertices(n_pos), normals(n_nrm);
        for (size_t i = 0; i < n_pos; ++i)
            vertices[i] = V3{(double)pos[3 * i], (double)pos[3 * i + 1], (double)pos[3 * i + 2]} * scale;
        for (size_t i = 0; i < n_nrm; ++i)
            normals[i] = V3{(double)nrm[3 * i], (double)nrm[3 * i + 1], (double)nrm[3 * i + 2]};
        for (size_t f = 0; f + 2 < n_idx; f += 3) {
            uint32_t i0 = idx[f], i1 = idx[f + 1], i2 = idx[f + 2];
            auto t = std::make_shared<Triangle>(vertices[i0], vertices[i1], vertices[i2], m);
            if (n_nrm) {
                t->has_n = true;
                t->nrm[0] = normals[i0]; t->nrm[1] = normals[i1]; t->nrm[2] = normals[i2];
            }
            if (n_uv) {
                t->has_uv = true;
                const uint32_t ii[3] = {i0, i1, i2};
                for (int k = 0; k < 3; ++k) {
                    t->uvs[k][0] = (double)uv[2 * ii[k]];
                    t->uvs[k][1] = (double)uv[2 * ii[k] + 1];
                }
            }
            triangles.add(t);
        }
        triangles.build_bvh();
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override { return triangles.intersects(ray, ray_t, out, c); }
    AABB bounding_box() const override { return triangles.bounding_box(); }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override { return triangles.sample(origin, time, rng, dir); }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override { return triangles.pdf(origin, direction, time, c); }
    uint32_t assign_ids(uint32_t first) override { return triangles.assign_ids(first); }
    uint32_t prim_count() const override { return triangles.prim_count(); }
};

struct Instance : Hittable {  // instance.rs — rotate then translate
    HitPtr object;   // Arc<dyn Hittable>: may be shared by several instances and may itself be an Instance (instance.rs:20-30)
    AABB bbox;
    Rigid m;
    // Canonical primitive ids (the build's tie rule, not the reference's): the wrapped object numbers its primitives
    // locally from 0 and every instance adds the offset of ITS placement, so an object shared by several instances gets
    // distinct ids per placement while comparisons inside the object see one consistent order.
    uint32_t id_offset = 0;
    Instance(HitPtr obj, V3 axis, double angle, V3 translation) : object(obj) {
        m = rigid_from_rotation_translation(quat_from_axis_angle(axis, angle), translation);
        bbox = obj->bounding_box().transformed(m);
    }
    Ray to_local_ray(V3 o, V3 d, double time) const {  // :36-38 (Ray ctor re-normalises)
        return Ray(xform_point(m.i0, m.i1, m.i2, m.it, o), xform_vector(m.i0, m.i1, m.i2, d), time);
    }
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {  // :34-54
        Ray local = to_local_ray(ray.o, ray.d, ray.time);
        if (!object->intersects(local, ray_t, out, c)) return false;
        out.prim_id += id_offset;
        out.point = xform_point(m.c0, m.c1, m.c2, m.t, out.point);
        out.geometric_normal = normalize(xform_vector(m.c0, m.c1, m.c2, out.geometric_normal));
        // Q1: shading_normal, front_face, u, v stay as computed in LOCAL space.
        return true;
    }
    AABB bounding_box() const override { return bbox; }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override {  // :64-69
        V3 lo = xform_point(m.i0, m.i1, m.i2, m.it, origin);
        V3 ld;
        if (!object->sample(lo, time, rng, ld)) return false;
        dir = xform_vector(m.c0, m.c1, m.c2, ld);
        return true;
    }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override {  // :71-75
        return object->pdf(xform_point(m.i0, m.i1, m.i2, m.it, origin), xform_vector(m.i0, m.i1, m.i2, direction), time, c);
    }
    uint32_t assign_ids(uint32_t first) override {
        id_offset = first;
        return first + object->assign_ids(0);
    }
    uint32_t prim_count() const override { return object->prim_count(); }
};

// One entry of a world list (World::add_object / add_light take any Arc<dyn Hittable>, world.rs:18-24 — the same object may be
// added several times and may also sit under instances). Only the build's canonical primitive ids need this wrapper: the
// object numbers its primitives locally from 0 and every PLACEMENT adds its own offset, exactly as Instance does; every other
// call is forwarded unchanged (no arithmetic of its own).
struct Placement : Hittable {
    HitPtr object;
    uint32_t id_offset = 0;
    explicit Placement(HitPtr obj) : object(std::move(obj)) {}
    bool intersects(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const override {
        if (!object->intersects(ray, ray_t, out, c)) return false;
        out.prim_id += id_offset;
        return true;
    }
    AABB bounding_box() const override { return object->bounding_box(); }
    bool sample(V3 origin, double time, Rng& rng, V3& dir) const override { return object->sample(origin, time, rng, dir); }
    double pdf(V3 origin, V3 direction, double time, Counters& c) const override { return object->pdf(origin, direction, time, c); }
    uint32_t assign_ids(uint32_t first) override {
        id_offset = first;
        return first + object->assign_ids(0);
    }
    uint32_t prim_count() const override { return object->prim_count(); }
};

struct World {  // world.rs
    HittableList objects, lights;
    uint32_t n_prims = 0;
    void build_bvh() {  // :26-29 (+ canonical id assignment, lights first)
        uint32_t next = lights.assign_ids(0);
        n_prims = objects.assign_ids(next);
        objects.build_bvh();
        lights.build_bvh();
    }
    bool intersect_all(const Ray& ray, Interval ray_t, HitInfo& out, Counters& c) const {  // :47-62
        ++c.segments;
        HitInfo lh, oh;
        bool hl = lights.intersects(ray, ray_t, lh, c);
        bool ho = objects.intersects(ray, ray_t, oh, c);
        if (!hl && !ho) return false;
        if (hl && (!ho || lh.dist < oh.dist)) out = lh;  // tie -> object (object ids > light ids)
        else out = oh;
        return true;
    }
};

// ---------------------------------------------------------------------- camera.rs
struct Camera {
    // public fields, camera.rs:23-36
    double aspect_ratio = 1.0;
    uint32_t image_width = 0, samples_per_pixel = 0, max_depth = 0;
    double vfov = 0;
    V3 look_from{0, 0, 0}, look_at{0, 0, 0}, vup{0, 0, 0};
    double blur_strength = 0, focal_length = 0, defocus_angle = 0;
    bool env_is_map = false;
    V3 env_color{0, 0, 0};
    std::shared_ptr<ImageRGB8> env_map;
    // derived, camera.rs:38-47
    V3 forward, right, up, center, pixel00, pixel_du, pixel_dv;
    uint32_t image_height = 0;

    static double to_radians(double deg) { return deg * (PI / 180.0); }
    void init() {  // camera.rs:51-77
        image_height = (uint32_t)((double)image_width / aspect_ratio);
        center = look_from;
        double theta = to_radians(vfov);
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h * focal_length;
        double viewport_width = viewport_height * ((double)image_width / (double)image_height);
        forward = normalize(look_from - look_at);
        right = normalize(cross(vup, forward));
        up = cross(forward, right);
        V3 viewport_u = right * viewport_width;
        V3 viewport_v = up * -viewport_height;
        pixel_du = viewport_u / (double)image_width;
        pixel_dv = viewport_v / (double)image_height;
        V3 upperleft = center - (forward * focal_length) - (viewport_u / 2.0) - (viewport_v / 2.0);
        pixel00 = upperleft + (pixel_du + pixel_dv) * 0.5;
    }
    static void random_offsets(Rng& rng, double& ox, double& oy) {  // :133-138
        double radius = std::sqrt(rng.gen());
        double angle = rng.gen() * 2.0 * PI;
        ox = radius * m_cos(angle);
        oy = radius * m_sin(angle);
    }
    V3 sample_environment(const Ray& ray) const {  // :140-151
        if (!env_is_map) return env_color;
        double theta = m_acos(ray.d.y);
        double phi = m_atan2(ray.d.z, ray.d.x);
        double u = (phi + PI) / (2.0 * PI);
        double v = 1.0 - theta / PI;
        return env_map->value(u, v, V3{0, 0, 0});
    }
    Ray generate_ray(uint32_t r, uint32_t c, Rng& rng) const {  // :153-168
        double bx, by;
        random_offsets(rng, bx, by);
        bx = bx * blur_strength;
        by = by * blur_strength;
        // NB: the x offset moves along rows (pixel_dv), the y offset along columns (:156-157)
        V3 sample_location = pixel00 + (pixel_dv * ((double)r + bx)) + (pixel_du * ((double)c + by));
        double radius = std::tan(to_radians(defocus_angle / 2.0)) * focal_length;
        V3 dof_right = right * radius, dof_up = up * radius;
        double px, py;
        random_offsets(rng, px, py);  // drawn even when defocus_angle == 0
        V3 origin = center + (dof_right * px) + (dof_up * py);
        V3 direction = sample_location - origin;
        double time = rng.gen();
        return Ray(origin, direction, time);
    }
    struct PathRecord {  // optional per-bounce trace dump for debugging GPU divergence
        double t;
        uint32_t prim_id;
        V3 point, throughput;
    };
    V3 trace(uint32_t r, uint32_t c, const World& world, Rng& rng, Counters& cnt,
             std::vector<PathRecord>* dump = nullptr) const {  // :170-228
        const double eps = 1e-3;
        const uint32_t min_bounces = 5;
        V3 radiance{0, 0, 0}, throughput{1, 1, 1};
        Ray ray = generate_ray(r, c, rng);
        for (uint32_t bounces = 0; bounces < max_depth; ++bounces) {
            HitInfo hit;
            if (!world.intersect_all(ray, Interval{eps, INF}, hit, cnt)) {
                radiance += throughput * sample_environment(ray);
                break;
            }
            if (dump) dump->push_back({hit.dist, hit.prim_id, hit.point, throughput});
            radiance += throughput * hit.mat->emitted(hit.u, hit.v, hit.point);
            if (bounces > min_bounces) {  // russian roulette :190-196
                double p = clampd(luminance(throughput), 0.01, 1.0);
                if (rng.gen() > p) break;
                throughput /= p;
            }
            double p_light = world.lights.objects.empty() ? 0.0 : 0.5;  // :199-200
            double p_bsdf = 1.0 - p_light;
            double rsel = rng.gen();  // drawn even when p_light == 0
            V3 dir;
            bool ok = rsel < p_light ? world.lights.sample(hit.point, ray.time, rng, dir)
                                     : hit.mat->sample(ray, hit, rng, dir);
            if (!ok) break;
            double bsdf_pdf = hit.mat->pdf(-ray.d, dir, hit);
            double light_pdf = world.lights.pdf(hit.point, dir, ray.time, cnt);
            double pdf = p_bsdf * bsdf_pdf + p_light * light_pdf;
            V3 brdf = hit.mat->eval(-ray.d, dir, hit);
            V3 attenuation = brdf / pdf;
            double e = EPS * signum(dot(dir, hit.geometric_normal));
            Ray next(hit.point + e * hit.geometric_normal, dir, ray.time);
            throughput *= attenuation;
            ray = next;
        }
        return radiance;
    }
};

}  // namespace orc
