// Device-side textures and BSDFs: Lambert, GGX metal, GGX rough glass, the 4-lobe
// Disney-style principled BSDF and the diffuse emitter. One function per reference
// method; the reference's sample()/pdf()/eval() triple is evaluated in one pass per bounce
// with common sub-expressions (frame, local v/l, half vector, D, G1) computed once — the
// reference recomputes them with identical inputs, so sharing changes no result bit.
// Quirks kept on purpose: Q2 (sampler stretches by roughness^2 while D/G use
// alpha^2 = roughness^2), Q3 (GTR1 with log2 and |l.h|), Q4 (glass ignores base colour).
#pragma once
#include "pt_dev_math.h"
#include "pt_types.h"

namespace pt {

struct HitD {
    V3 point, gn, sn;      // world point, geometric normal (world), shading normal (LOCAL under an instance: Q1)
    double dist, u, v;
    bool front;
    uint32_t mat;
};

// ---- textures (texture.rs) -------------------------------------------------------------
PT_DEV int32_t f64_as_i32(double x) {          // Rust `as i32`: saturating, NaN -> 0
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
PT_DEV uint32_t f64_as_u32(double x) {
    if (x != x || x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
PT_DEV bool checker_is_first(double inv_scale, V3 p) {   // texture.rs:43-49
    int32_t x = f64_as_i32(floor(p.x * inv_scale));
    int32_t y = f64_as_i32(floor(p.y * inv_scale));
    int32_t z = f64_as_i32(floor(p.z * inv_scale));
    int32_t s = (int32_t)((uint32_t)x + (uint32_t)y + (uint32_t)z);
    return s % 2 == 0;
}
PT_DEV V3 tex_image(const SceneD& sc, const TexD& T, double u, double v) {   // texture.rs:72-91
    if (T.h == 0) return V3{0.0, 1.0, 1.0};
    u = clampd(u, 0.0, 1.0);
    v = 1.0 - clampd(v, 0.0, 1.0);
    uint32_t i = f64_as_u32(u * (double)T.w);
    uint32_t j = f64_as_u32(v * (double)T.h);
    if (i > T.w - 1) i = T.w - 1;      // Q6: the reference would panic at u==1 / v==0
    if (j > T.h - 1) j = T.h - 1;
    if (T.kind == TEX_IMAGE_F32) {     // float samples, no RGB8 squash (pt_types.h)
        const float* q = sc.atlas_f + T.ofs + ((size_t)j * T.w + i) * 3;
        return V3{(double)q[0], (double)q[1], (double)q[2]};
    }
    const uint8_t* p = sc.atlas + T.ofs + ((size_t)j * T.w + i) * 3;
    const double s = 1.0 / 255.0;
    return V3{s * (double)p[0], s * (double)p[1], s * (double)p[2]};
}
PT_DEV V3 tex_rgb(const SceneD& sc, int32_t t, double u, double v, V3 p) {
    for (int depth = 0; depth < 16; ++depth) {
        const TexD& T = sc.tex[t];
        if (T.kind == TEX_CHECKER) {
            const bool first = checker_is_first(T.inv_scale, p);
            if (T.flat) return first ? V3{T.c1[0], T.c1[1], T.c1[2]} : V3{T.c2[0], T.c2[1], T.c2[2]};
            t = (int32_t)(first ? T.t1 : T.t2);
            continue;
        }
        if (T.kind == TEX_IMAGE || T.kind == TEX_IMAGE_F32) return tex_image(sc, T, u, v);
        return V3{T.v[0], T.v[1], T.v[2]};
    }
    return V3{0.0, 0.0, 0.0};
}
PT_DEV double tex_f(const SceneD& sc, int32_t t, V3 p) {
    for (int depth = 0; depth < 16; ++depth) {
        const TexD& T = sc.tex[t];
        if (T.kind == TEX_CHECKER) {
            const bool first = checker_is_first(T.inv_scale, p);
            if (T.flat) return first ? T.c1[0] : T.c2[0];
            t = (int32_t)(first ? T.t1 : T.t2);
            continue;
        }
        return T.v[0];
    }
    return 0.0;
}

// ---- microfacet helpers (bsdf/mod.rs:61-97, sampling.rs) --------------------------------
PT_DEV V3 tint(V3 base) {
    double l = luminance(base);
    return l > 0.0 ? base / l : V3{1.0, 1.0, 1.0};
}
PT_DEV double r0_of(double eta) { return powi2((eta - 1.0) / (eta + 1.0)); }
PT_DEV double fresnel_dielectric(V3 w, V3 h, double eta_i, double eta_o) {
    double c = fabs(dot(w, h));
    double g_squared = powi2(eta_o / eta_i) - 1.0 + c * c;
    if (g_squared < 0.0) return 1.0;
    double g = sqrt(g_squared);
    double gmc = g - c, gpc = g + c;
    double x = (c * gpc - 1.0) / (c * gmc + 1.0);
    return 0.5 * (gmc * gmc) / (gpc * gpc) * (1.0 + x * x);
}
PT_DEV V3 fresnel_schlick(V3 r0, double angle) { return r0 + (1.0 - r0) * powi5(1.0 - angle); }
PT_DEV double schlick_weight(double x) { return powi5(clampd(1.0 - x, 0.0, 1.0)); }
PT_DEV double ggx_D(V3 h, double roughness) {
    double cos_theta = fmax(h.z, 0.001);
    double alpha2 = fmax(roughness * roughness, 0.001);
    double denom = (alpha2 - 1.0) * (cos_theta * cos_theta) + 1.0;
    return alpha2 / (D_PI * denom * denom);
}
PT_DEV double ggx_G1(V3 w, double roughness) {
    double alpha2 = fmax(roughness * roughness, 0.001);
    double cos_theta = fabs(w.z);
    return 2.0 * cos_theta / (cos_theta + sqrt(cos_theta * cos_theta * (1.0 - alpha2) + alpha2));
}
PT_DEV double gtr1_D(double abs_cos_theta, double alpha_g) {
    double alpha2 = alpha_g * alpha_g;
    double t = 1.0 + (alpha2 - 1.0) * abs_cos_theta * abs_cos_theta;
    return (alpha2 - 1.0) / (D_PI * t * dev_log2(alpha2));
}
PT_DEV V3 cosine_sample_hemisphere(Rng& rng, double two_pi_scale) {   // sampling.rs:18-24
    uint64_t a, b;
    rng_u64x2(rng, a, b);
    double phi = ((double)(a >> 12) * (1.0 / 4503599627370496.0)) * two_pi_scale;
    double r2 = u64_to_unit(b);
    double r2s = sqrt(r2);
    const SinCos sc_phi = dev_sincos(phi);
    const double sn = sc_phi.s, cs = sc_phi.c;
    return V3{r2s * cs, r2s * sn, sqrt(1.0 - r2)};
}
PT_DEV V3 ggx_sample_microfacet_normal(V3 v_in, double roughness, Rng& rng) {   // sampling.rs:57-94
    double a2 = roughness * roughness;
    V3 v = normalize(V3{v_in.x * a2, v_in.y * a2, v_in.z});
    V3 t1 = v.z < 0.9999 ? normalize(cross(v, V3{0.0, 0.0, 1.0})) : V3{1.0, 0.0, 0.0};
    V3 t2 = cross(t1, v);
    uint64_t ua, ub;
    rng_u64x2(rng, ua, ub);
    double e1 = u64_to_unit(ua), e2 = u64_to_unit(ub);
    double a = 1.0 / (1.0 + v.z);
    double r = sqrt(e1);
    double phi = e2 < a ? e2 / a * D_PI : D_PI + (e2 - a) / (1.0 - a) * D_PI;
    const SinCos sc_phi = dev_sincos(phi);
    const double sn = sc_phi.s, cs = sc_phi.c;
    double p1 = r * cs;
    double p2 = r * sn * (e2 < a ? 1.0 : v.z);
    V3 n = p1 * t1 + p2 * t2 + sqrt(fmax(1.0 - p1 * p1 - p2 * p2, 0.0)) * v;
    V3 h = normalize(V3{a2 * n.x, a2 * n.y, fmax(n.z, 0.0)});
    return h.z < 0.0 ? -h : h;
}
PT_DEV V3 gtr1_sample_microfacet_normal(double alpha, Rng& rng) {   // sampling.rs:126-142
    uint64_t ua, ub;
    rng_u64x2(rng, ua, ub);
    double e1 = u64_to_unit(ua), e2 = u64_to_unit(ub);
    double alpha2 = alpha * alpha;
    double cos_theta = (1.0 - dev_pow(alpha2, 1.0 - e1)) / (1.0 - alpha2);
    double sin_theta = sqrt(fmax(1.0 - cos_theta * cos_theta, 0.0));
    double phi = 2.0 * D_PI * e2;
    const SinCos sc_phi = dev_sincos(phi);
    const double sn = sc_phi.s, cs = sc_phi.c;
    V3 h{sin_theta * cs, sin_theta * sn, cos_theta};
    return h.z < 0.0 ? -h : h;
}
PT_DEV V3 generalized_half(V3 v, V3 l, bool is_reflect, double eta_i, double eta_o) {   // glass.rs:103-107
    if (is_reflect) return normalize(l + v) * signum(v.z);
    return -normalize(l * eta_o + v * eta_i);
}
// Walter 2007 reflection/refraction jacobian shared by glass.rs:113-121 and principled.rs:175-182
PT_DEV double glass_jacobian(double f, V3 v, V3 l, V3 h, double eta_i, double eta_o, bool is_reflect) {
    if (is_reflect) return f * 1.0 / (4.0 * fabs(dot(l, h)));
    double v_dot_h = dot(v, h), l_dot_h = dot(l, h);
    return (1.0 - f) * (eta_o * eta_o * fabs(l_dot_h)) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
}
PT_DEV double glass_factor(double f, double g, double d, V3 v, V3 l, V3 h, double eta_i, double eta_o,
                           bool is_reflect) {   // glass.rs:141-150, principled.rs:235-245
    if (is_reflect) return f * g * d / (4.0 * fabs(l.z) * fabs(v.z));
    double l_dot_h = dot(l, h), v_dot_h = dot(v, h);
    double term1 = fabs((l_dot_h * v_dot_h) / (l.z * v.z));
    double term2 = (eta_o * eta_o) / powi2(eta_i * v_dot_h + eta_o * l_dot_h);
    return term1 * term2 * (1.0 - f) * g * d;
}
PT_DEV V3 sample_dielectric(V3 v, V3 h, double eta_i, double eta_o, Rng& rng) {   // glass.rs:79-89
    double f = fresnel_dielectric(v, h, eta_i, eta_o);
    if (rng_f64(rng) < f) return reflect(-v, h);
    V3 t = refract(-v, h, eta_i / eta_o);
    if (is_zero(t)) t = reflect(-v, h);
    return t;
}

// The texture values a material's sample / pdf / eval read (base colour, roughness; texture.rs `value`), fetched ONCE,
// up front: k_shade keeps every global-memory read of a bounce in its first phase so that the pure arithmetic that
// follows — sample, pdf, eval, next ray — can hide the asynchronous fetch of the wave's next group of path records.
// The reference looks the same textures up again in each of sample / pdf / eval with the same arguments: same values.
constexpr int MIX_MAX_DEPTH = 2;     // levels of MixBxDf the host admits (pt_mat_mix): a mix of mixes of leaves
struct TexVals {
    V3 color;
    double rough;
};
// The shading frame of a bounce and the view vector in it. sample(), pdf() and eval() of the reference each rebuild the same
// quaternion from the same normal and rotate the same -ray.direction (sampling.rs:8-16 via diffuse.rs / metal.rs / glass.rs /
// principled.rs): k_shade forms them ONCE per bounce for the hit's material (make_local_frame) — same inputs, same bits — and the
// sampler and the pdf/eval code below take them from here (LF = true). The children of a MixBxDf build their own (LF = false).
struct LocalFrame {
    Frame f;
    V3 v;
};
PT_DEV LocalFrame make_local_frame(const MatD& m, const HitD& h, V3 wo) {
    LocalFrame lf{};
    const uint32_t k = m.kind;
    if (k == MAT_MIX || k == MAT_LIGHT) return lf;
    lf.f = frame_to_z((k == MAT_PRINCIPLED || k == MAT_SHEEN) ? h.gn : h.sn);    // principled.rs / sheen.rs use the geometric normal
    if (k != MAT_DIFFUSE) lf.v = to_local(lf.f, wo);                              // (Lambert needs no view vector)
    return lf;
}
PT_DEV TexVals fetch_tex(const SceneD& sc, const MatD& m, const HitD& h) {
    TexVals tv{V3{0.0, 0.0, 0.0}, 0.0};
    const uint32_t k = m.kind;
    if (k == MAT_DIFFUSE || k == MAT_METAL || k == MAT_PRINCIPLED || k == MAT_LIGHT)
        tv.color = m.color_solid ? V3{m.color_v[0], m.color_v[1], m.color_v[2]} : tex_rgb(sc, m.color_tex, h.u, h.v, h.point);
    if (k == MAT_METAL || k == MAT_GLASS) tv.rough = m.rough_solid ? m.rough_v : tex_f(sc, m.rough_tex, h.point);
    return tv;   // MAT_MIX: its children fetch their own (mat_sample / mat_pdf_eval)
}

// ---- BxDFMaterial::sample (bsdf/mod.rs:23) ----------------------------------------------
// wo = -ray.direction. Returns false where the reference returns None.
PT_DEV bool mat_sample(const SceneD& sc, const MatD& mat, const HitD& h, V3 wo, Rng& rng, double two_pi_scale, const TexVals& tv, const LocalFrame& lf, V3& dir) {
    const MatD* leaf = &mat;
    double rough = tv.rough;
    const bool own = mat.kind == MAT_MIX;   // a mix's child builds its own frame; everything else uses the bounce's (make_local_frame)
    if (mat.kind == MAT_MIX) {   // mix.rs:25-32: the selector is drawn first, then the chosen child samples — which may be a mix again
#pragma unroll 1
        for (int depth = 0; depth < MIX_MAX_DEPTH && leaf->kind == MAT_MIX; ++depth) {
            double p = rng_f64(rng);
            leaf = &sc.mats[leaf->p[0] < p ? leaf->color_tex : leaf->rough_tex];
        }
        if (leaf->kind == MAT_METAL || leaf->kind == MAT_GLASS) rough = tex_f(sc, leaf->rough_tex, h.point);
    }
    const MatD& m = *leaf;
    LocalFrame cf = lf;
    if (own) cf = make_local_frame(m, h, wo);
    const Frame f = cf.f;
    const V3 v = cf.v;
    switch (m.kind) {
    case MAT_DIFFUSE: {   // diffuse.rs:51-54
        dir = to_world(f, cosine_sample_hemisphere(rng, two_pi_scale));
        return true;
    }
    case MAT_METAL: {     // metal.rs:39-54
        V3 hv = ggx_sample_microfacet_normal(v, rough, rng);
        V3 d = to_world(f, reflect(-v, hv));
        if (dot(d, h.sn) <= 0.0) return false;
        dir = d;
        return true;
    }
    case MAT_GLASS: {     // glass.rs:66-90
        V3 hv = ggx_sample_microfacet_normal(v, rough, rng);
        double eta_i = h.front ? 1.0 : m.ior, eta_o = h.front ? m.ior : 1.0;
        dir = to_world(f, sample_dielectric(v, hv, eta_i, eta_o, rng));
        return true;
    }
    case MAT_PRINCIPLED: {   // principled.rs:262-276 (lobe pick drawn before the lobe's own draws)
        double r = rng_f64(rng);
        const double p0 = m.lobe_p[0], p1 = m.lobe_p[1], p2 = m.lobe_p[2];
        if (r < p0) {
            dir = to_world(f, cosine_sample_hemisphere(rng, two_pi_scale));
            return true;
        }
        const double roughness = m.p[1], ior = m.p[5];
        if (r < p0 + p1) {
            V3 hv = ggx_sample_microfacet_normal(v, roughness, rng);
            V3 d = to_world(f, reflect(-v, hv));
            if (dot(d, h.gn) <= 0.0) return false;
            dir = d;
            return true;
        }
        if (r < p0 + p1 + p2) {
            V3 hv = ggx_sample_microfacet_normal(v, roughness, rng);
            double eta_i = h.front ? 1.0 : ior, eta_o = h.front ? ior : 1.0;
            dir = to_world(f, sample_dielectric(v, hv, eta_i, eta_o, rng));
            return true;
        }
        V3 hv = gtr1_sample_microfacet_normal(0.25, rng);
        V3 d = to_world(f, reflect(-v, hv));
        if (dot(d, h.gn) <= 0.0) return false;
        dir = d;
        return true;
    }
    case MAT_SHEEN: {     // sheen.rs:27-30
        dir = to_world(f, cosine_sample_hemisphere(rng, two_pi_scale));
        return true;
    }
    case MAT_CLEARCOAT: { // clearcoat.rs:23-35
        V3 hv = gtr1_sample_microfacet_normal(0.25, rng);
        V3 d = to_world(f, reflect(-v, hv));
        if (dot(d, h.sn) <= 0.0) return false;
        dir = d;
        return true;
    }
    default:   // MAT_LIGHT: material.rs:168-170
        return false;
    }
}

// ---- BxDFMaterial::pdf + eval (cosine included in eval) -----------------------------------
PT_DEV void leaf_pdf_eval(const SceneD& sc, const MatD& m, const HitD& h, V3 wo, V3 wi, const TexVals& tv, const LocalFrame& lf, double& pdf, V3& brdf) {
    const Frame f = lf.f;     // the leaf's frame and local view vector (make_local_frame)
    const V3 v = lf.v;
    switch (m.kind) {
    case MAT_DIFFUSE: {   // diffuse.rs:56-65
        V3 l = to_local(f, wi);
        V3 color = tv.color;
        pdf = fabs(l.z) / D_PI;
        brdf = fabs(l.z) * (color / D_PI);
        return;
    }
    case MAT_METAL: {     // metal.rs:56-80
        V3 l = to_local(f, wi);
        V3 hv = normalize(v + l);
        double rough = tv.rough;
        V3 base = tv.color;
        double g1v = ggx_G1(v, rough), d = ggx_D(hv, rough);
        double pdf_h = g1v * fabs(dot(v, hv)) * d / fabs(v.z);
        pdf = pdf_h * (1.0 / (4.0 * fabs(dot(l, hv))));
        double g = g1v * ggx_G1(l, rough);
        V3 fr = fresnel_schlick(base, dot(l, hv));
        brdf = fabs(l.z) * (fr * g * d / (4.0 * fabs(l.z) * fabs(v.z)));
        return;
    }
    case MAT_GLASS: {     // glass.rs:92-163
        V3 l = to_local(f, wi);
        bool is_reflect = l.z * v.z > 0.0;
        double eta_i = h.front ? 1.0 : m.ior, eta_o = h.front ? m.ior : 1.0;
        V3 hv = generalized_half(v, l, is_reflect, eta_i, eta_o);
        double rough = tv.rough;
        double g1v = ggx_G1(v, rough), d = ggx_D(hv, rough);
        double pdf_h = g1v * fabs(dot(v, hv)) * d / fabs(v.z);
        double fr = fresnel_dielectric(v, hv, eta_i, eta_o);
        pdf = pdf_h * glass_jacobian(fr, v, l, hv, eta_i, eta_o, is_reflect);
        double g = g1v * ggx_G1(l, rough);
        brdf = splat(glass_factor(fr, g, d, v, l, hv, eta_i, eta_o, is_reflect)) * fabs(l.z);
        return;
    }
    case MAT_PRINCIPLED: {   // principled.rs:278-366, frame = geometric normal
        V3 l = to_local(f, wi);
        bool is_reflect = l.z * v.z > 0.0;
        const double metallic = m.p[0], roughness = m.p[1], subsurface = m.p[2], specular = m.p[3],
                     specular_tint = m.p[4], ior = m.p[5], sheen = m.p[7], sheen_tint = m.p[8];
        double eta_i = h.front ? 1.0 : ior, eta_o = h.front ? ior : 1.0;
        V3 hv = generalized_half(v, l, is_reflect, eta_i, eta_o);
        V3 base = tv.color;
        double acc_pdf = 0.0;
        V3 acc{0.0, 0.0, 0.0};
        const double d = ggx_D(hv, roughness), g1v = ggx_G1(v, roughness);
        const double g = g1v * ggx_G1(l, roughness);
        const double pdf_h = g1v * fabs(dot(v, hv)) * d / fabs(v.z);
        if (m.lobe_p[0] > 0.0 && is_reflect) {   // diffuse + sheen (+ fake subsurface) :196-214,:342-347
            acc_pdf += m.lobe_p[0] * (fabs(l.z) / D_PI);
            V3 c_sheen = vlerp(splat(1.0), tint(base), sheen_tint);
            V3 sheen_term = sheen * c_sheen * schlick_weight(fabs(dot(l, hv)));
            double l_dot_h = dot(l, hv);
            double rr = 2.0 * roughness * l_dot_h * l_dot_h;
            double fl = schlick_weight(l.z), fv = schlick_weight(v.z);
            double f_retro = rr * (fl + fv + fl * fv * (rr - 1.0));
            double f_d = (1.0 - 0.5 * fl) * (1.0 - 0.5 * fv);
            double fss90 = 0.5 * rr;
            double f_ss = flerp(1.0, fss90, fl) * flerp(1.0, fss90, fv);
            double ss = 1.25 * (f_ss * (1.0 / (l.z + v.z) - 0.5) + 0.5);
            V3 diffuse_term = base / D_PI * flerp(f_d + f_retro, ss, subsurface);
            acc = acc + m.lobe_w[0] * (diffuse_term + sheen_term);
        }
        if (m.lobe_p[1] > 0.0 && is_reflect) {   // GGX specular :161-168,:216-225,:348-358
            acc_pdf += m.lobe_p[1] * (pdf_h * (1.0 / (4.0 * fabs(dot(l, hv)))));
            V3 ks = vlerp(splat(1.0), tint(base), specular_tint);
            V3 c0 = vlerp(specular * r0_of(eta_i / eta_o) * ks, base, metallic);
            V3 metallic_fresnel = fresnel_schlick(c0, dot(l, hv));
            V3 dielectric = splat(fresnel_dielectric(v, hv, eta_i, eta_o));
            V3 fresnel = vlerp(dielectric, metallic_fresnel, metallic);
            acc = acc + m.lobe_w[1] * (fresnel * g * d / (4.0 * fabs(l.z) * fabs(v.z)));
        }
        if (m.lobe_p[2] > 0.0) {                 // GGX glass :170-185,:227-246
            double fr = fresnel_dielectric(v, hv, eta_i, eta_o);
            acc_pdf += m.lobe_p[2] * (pdf_h * glass_jacobian(fr, v, l, hv, eta_i, eta_o, is_reflect));
            acc = acc + m.lobe_w[2] * splat(glass_factor(fr, g, d, v, l, hv, eta_i, eta_o, is_reflect));
        }
        if (m.lobe_p[3] > 0.0 && is_reflect) {   // GTR1 clearcoat :187-192,:248-258 (Q3)
            double l_h = fabs(dot(l, hv));
            double dc = gtr1_D(l_h, m.alpha_g);
            double g1v_c = ggx_G1(v, 0.25);
            double pdf_hc = g1v_c * fabs(dot(v, hv)) * dc / fabs(v.z);
            acc_pdf += m.lobe_p[3] * (pdf_hc * (1.0 / (4.0 * l_h)));
            double gc = g1v_c * ggx_G1(l, 0.25);
            V3 fc = fresnel_schlick(splat(r0_of(1.5)), dot(l, hv));
            acc = acc + m.lobe_w[3] * (fabs(l.z) * (fc * dc * gc / (4.0 * fabs(l.z) * fabs(v.z))));
        }
        pdf = acc_pdf;
        brdf = acc * fabs(l.z);
        return;
    }
    case MAT_SHEEN: {     // sheen.rs:32-44
        V3 l = to_local(f, wi);
        V3 hv = normalize(v + l);
        V3 c_sheen = vlerp(splat(1.0), tint(V3{m.p[0], m.p[1], m.p[2]}), m.p[3]);
        pdf = fabs(l.z) / D_PI;
        brdf = c_sheen * powi5(1.0 - fabs(dot(l, hv))) * fabs(l.z);
        return;
    }
    case MAT_CLEARCOAT: { // clearcoat.rs:37-60
        V3 l = to_local(f, wi);
        V3 hv = normalize(v + l);
        double l_h = fabs(dot(l, hv));
        double dc = gtr1_D(l_h, m.alpha_g);
        double g1v = ggx_G1(v, 0.25);
        double pdf_h = g1v * fabs(dot(v, hv)) * dc / fabs(v.z);
        pdf = pdf_h * (1.0 / (4.0 * l_h));
        double gc = g1v * ggx_G1(l, 0.25);
        V3 fc = fresnel_schlick(splat(r0_of(1.5)), dot(l, hv));
        brdf = fabs(l.z) * (fc * dc * gc / (4.0 * fabs(l.z) * fabs(v.z)));
        return;
    }
    default:   // MAT_LIGHT: material.rs:172-178
        pdf = 1.0;
        brdf = V3{1.0, 1.0, 1.0};
        return;
    }
}
// BxDFMaterial::pdf + eval incl. MixBxDf (mix.rs:34-44): (1-t)*child1 + t*child2, where a child may itself be a mix (MixBxDf::new takes
// any Arc<dyn BxDFMaterial>, mix.rs:14-20) — the host admits MIX_MAX_DEPTH = 2 levels. The reference's recursion rounds every level's
// two products and their sum separately, so the weights are NOT multiplied through: the (at most four) leaves are visited in
// the recursion's order by ONE non-unrolled loop (the leaf code above is instantiated once), an inner accumulator holding the
// current child's value while its second leaf is evaluated and an outer one holding the first child's weighted value.
PT_DEV void mat_pdf_eval(const SceneD& sc, const MatD& m, const HitD& h, V3 wo, V3 wi, const TexVals& tv, const LocalFrame& lf, double& pdf, V3& brdf) {
    const bool mix = m.kind == MAT_MIX;
    pdf = 0.0;
    brdf = V3{0.0, 0.0, 0.0};
    double cp = 0.0;                       // value of the child being evaluated
    V3 cf{0.0, 0.0, 0.0};
    const int n_leaves = mix ? 4 : 1;      // slots (child, grandchild); a leaf child uses one of its two
#pragma unroll 1
    for (int slot = 0; slot < n_leaves; ++slot) {
        const int c = slot >> 1, g = slot & 1;
        const MatD& child = mix ? sc.mats[c == 0 ? m.color_tex : m.rough_tex] : m;
        const bool inner = mix && child.kind == MAT_MIX;
        if (mix && !inner && g == 1) continue;                      // a leaf child has no second slot
        const MatD& lm = inner ? sc.mats[g == 0 ? child.color_tex : child.rough_tex] : child;
        double lp;
        V3 lfv;
        const TexVals ltv = mix ? fetch_tex(sc, lm, h) : tv;
        const LocalFrame llf = mix ? make_local_frame(lm, h, wo) : lf;
        leaf_pdf_eval(sc, lm, h, wo, wi, ltv, llf, lp, lfv);
        if (!mix) {
            pdf = lp;
            brdf = lfv;
            break;
        }
        if (inner) {                                                // mix.rs:34-44 one level down
            const double wi_ = g == 0 ? 1.0 - child.p[0] : child.p[0];
            const double wp = wi_ * lp;
            const V3 wf = wi_ * lfv;
            cp = g == 0 ? wp : cp + wp;
            cf = g == 0 ? wf : cf + wf;
            if (g == 0) continue;                                   // the child's second leaf comes next
        } else {
            cp = lp;
            cf = lfv;
        }
        const double w = c == 0 ? 1.0 - m.p[0] : m.p[0];
        const double wp = w * cp;
        const V3 wf = w * cf;
        pdf = c == 0 ? wp : pdf + wp;
        brdf = c == 0 ? wf : brdf + wf;
    }
}

}  // namespace pt
