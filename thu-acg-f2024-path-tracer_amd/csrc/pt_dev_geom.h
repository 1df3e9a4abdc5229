// Device-side primitive intersection (sphere / quad / triangle, optionally under a rigid
// instance), hit-record reconstruction, light-list sampling and camera-ray generation.
// Formulas and comparison strictness follow hittable/{sphere,quad,mesh,instance,hit_info}.rs
// and camera.rs:133-168 operation by operation (see pt_dev_math.h for the contract).
#pragma once
#include "pt_dev_bsdf.h"

namespace pt {

struct RayD {
    V3 o, d;   // d normalised (ray.rs:23-29)
    double time;
};
PT_DEV V3 ray_at(const RayD& r, double t) { return r.o + r.d * t; }
PT_DEV RayD make_ray(V3 o, V3 d, double time) { return RayD{o, normalize(d), time}; }
// instance.rs:36-38 with the analytic rigid inverse
PT_DEV RayD ray_to_local(const InstD& m, const RayD& r) {
    return make_ray(xform_point(m.i0, m.i1, m.i2, m.it, r.o), xform_vector(m.i0, m.i1, m.i2, r.d), r.time);
}

// Ray into the object space of a placement: Instance::intersects' transform (instance.rs:36-38, Ray::new re-normalises)
// of every instance of the chain, outermost first. U: the chain is wave-uniform (scalar loads). `innermost` receives the
// last index visited (the start of the way back: to_world_chain).
template <bool U = false>
PT_DEV RayD ray_to_local_chain(const SceneD& sc, int32_t outermost, RayD r, int32_t* innermost = nullptr) {
    int32_t last = -1;
    for (int32_t i = outermost; i >= 0;) {
        last = i;
        if constexpr (U) {
            const InstD m = ldu(&sc.insts[i]);
            r = ray_to_local(m, r);
            i = m.inner;
        } else {
            const InstD& m = sc.insts[i];
            r = ray_to_local(m, r);
            i = m.inner;
        }
    }
    if (innermost) *innermost = last;
    return r;
}

// sphere.rs:64-87 — open interval (t_min, +inf)
PT_DEV bool hit_sphere(const SphereD& s, const RayD& r, double t_min, double& t, V3& center) {
    center = ld3(s.p1) + (ld3(s.p2) - ld3(s.p1)) * r.time;
    V3 l = center - r.o;
    double sd = dot(l, r.d);
    double l2 = length_squared(l);
    double r2 = s.r * s.r;
    if (sd < 0.0 && l2 > r2) return false;
    double d2 = l2 - sd * sd;
    if (d2 > r2) return false;
    double q = sqrt(r2 - d2);
    t = l2 > r2 ? sd - q : sd + q;
    if (t <= t_min || t >= D_INF) return false;
    return true;
}
// quad.rs:40-59 — closed interval [t_min, +inf]
PT_DEV bool hit_quad(const QuadD& qd, const RayD& r, double t_min, double& t, double& alpha, double& beta) {
    V3 n = ld3(qd.n);
    double nd = dot(n, r.d);
    if (fabs(nd) < 1e-8) return false;
    t = (qd.d - dot(n, r.o)) / nd;
    if (!(t_min <= t && t <= D_INF)) return false;
    V3 p = ray_at(r, t) - ld3(qd.q);
    V3 w = ld3(qd.w);
    alpha = dot(w, cross(p, ld3(qd.v)));
    beta = dot(w, cross(ld3(qd.u), p));
    if (!(alpha >= 0.0 && alpha <= 1.0) || !(beta >= 0.0 && beta <= 1.0)) return false;
    return true;
}
// mesh.rs:50-82 Moeller-Trumbore — closed interval
PT_DEV bool hit_tri(const TriD& tr, const RayD& r, double t_min, double& t, double& u, double& v) {
    V3 v0 = ld3(tr.v0);
    V3 edge1 = ld3(tr.v1) - v0, edge2 = ld3(tr.v2) - v0;
    V3 h = cross(r.d, edge2);
    double a = dot(edge1, h);
    if (fabs(a) < 1e-8) return false;
    double f = 1.0 / a;
    V3 s = r.o - v0;
    u = f * dot(s, h);
    if (!(u >= 0.0 && u <= 1.0)) return false;
    V3 q = cross(s, edge1);
    v = f * dot(r.d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    t = f * dot(edge2, q);
    if (!(t_min <= t && t <= D_INF)) return false;
    return true;
}

// hit_info.rs:57-67
PT_DEV void tangent_basis(V3 n, V3& tangent, V3& bitangent) {
    V3 a = fabs(n.x) > 0.9 ? V3{0.0, 1.0, 0.0} : V3{1.0, 0.0, 0.0};
    tangent = normalize(cross(n, a));
    bitangent = cross(n, tangent);
}
// HitInfo::new hit_info.rs:16-55
// UM: `mat` is wave-uniform (lights.pdf walks the lights list in step): the material's normal-map handle arrives by a scalar load —
// a vector load there would make the pure-arithmetic stretch of k_shade wait for the record prefetch it is meant to hide.
template <bool UM = false>
PT_DEV void finish_hit(const SceneD& sc, const RayD& r, V3 point, V3 normal, double dist, uint32_t mat, double u,
                       double v, HitD& h) {
    h.front = dot(r.d, normal) < 0.0;
    V3 nn = normalize(normal);
    h.gn = h.front ? nn : -nn;
    int32_t nm;
    if constexpr (UM) nm = ldu(&sc.mats[mat].nmap_tex); else nm = sc.mats[mat].nmap_tex;
    if (nm >= 0) {
        V3 m = 2.0 * tex_image(sc, sc.tex[nm], u, v) - splat(1.0);
        V3 t, b;
        tangent_basis(h.gn, t, b);
        h.sn = normalize(m.x * t + m.y * b + m.z * h.gn);
    } else {
        h.sn = h.gn;
    }
    h.point = point;
    h.dist = dist;
    h.mat = mat;
    h.u = u;
    h.v = v;
}

// Rebuilds the reference's HitInfo for primitive `gid` known to be hit by `world_ray`.
// Re-runs that one primitive's intersection (same arithmetic as the traversal kernel, so the
// same t/u/v bits), then applies Instance::intersects' world transform (instance.rs:43-53, Q1).
PT_DEV bool reconstruct_hit(const SceneD& sc, const RayD& world_ray, uint32_t gid, double t_min, HitD& h) {
    const PrimRef pr = sc.prims[gid];
    int32_t innermost = -1;
    const RayD r = ray_to_local_chain(sc, pr.inst, world_ray, &innermost);
    const uint32_t kind = pr.kind & 0xFFu;
    if (kind == PRIM_SPHERE) {
        double t;
        V3 c;
        if (!hit_sphere(sc.spheres[pr.index], r, t_min, t, c)) return false;
        V3 point = ray_at(r, t);
        V3 normal = normalize(point - c);
        double theta = dev_acos(-normal.y);                       // sphere.rs:52-56
        double phi = dev_atan2(-normal.z, normal.x) + D_PI;
        finish_hit(sc, r, point, normal, t, pr.mat, phi / (2.0 * D_PI), theta / D_PI, h);
    } else if (kind == PRIM_QUAD) {
        double t, a, b;
        const QuadD& q = sc.quads[pr.index];
        if (!hit_quad(q, r, t_min, t, a, b)) return false;
        finish_hit(sc, r, ray_at(r, t), ld3(q.n), t, pr.mat, a, b, h);
    } else {
        double t, u, v;
        const TriD& tr = sc.tris[pr.index];
        if (!hit_tri(tr, r, t_min, t, u, v)) return false;
        V3 v0 = ld3(tr.v0);
        V3 edge1 = ld3(tr.v1) - v0, edge2 = ld3(tr.v2) - v0;
        double w = 1.0 - u - v;
        V3 normal;
        double tu = u, tv = v;
        if (pr.kind & PRIM_HAS_NORMALS) {                    // mesh.rs:85-87
            const TriAttr& at = sc.tri_attr[pr.index];
            normal = normalize(ld3(at.n[0]) * w + ld3(at.n[1]) * u + ld3(at.n[2]) * v);
        } else {
            normal = normalize(cross(edge1, edge2));          // flat shading :88
        }
        if (pr.kind & PRIM_HAS_UVS) {                        // mesh.rs:91-99
            const TriAttr& at = sc.tri_attr[pr.index];
            tu = at.uv[0][0] * w + at.uv[1][0] * u + at.uv[2][0] * v;
            tv = at.uv[0][1] * w + at.uv[1][1] * u + at.uv[2][1] * v;
        }
        finish_hit(sc, r, ray_at(r, t), normal, t, pr.mat, tu, tv, h);
    }
    for (int32_t i = innermost; i >= 0;) {                                       // instance.rs:43-53, innermost instance first
        const InstD& m = sc.insts[i];
        h.point = xform_point(m.c0, m.c1, m.c2, m.t, h.point);
        h.gn = normalize(xform_vector(m.c0, m.c1, m.c2, h.gn));
        i = m.outer;
    }
    return true;
}

// ---- lights list: Hittable::sample / pdf for every kind of object (list.rs:78-96, quad.rs:80-98,
// sphere.rs:110-135, mesh.rs:122-141, cuboid.rs:78-84, instance.rs:64-75) ---------------------------
PT_DEV V3 sample_quad_dir(const QuadD& q, V3 origin, Rng& rng) {                  // quad.rs:80-86
    uint64_t ua, ub;
    rng_u64x2(rng, ua, ub);                                                        // two consecutive draws: one Philox block when they share it
    double a = u64_to_unit(ua), b = u64_to_unit(ub);
    V3 point = ld3(q.q) + ld3(q.u) * a + ld3(q.v) * b;
    return normalize(point - origin);
}
PT_DEV double pdf_quad(const SceneD& sc, const QuadD& q, uint32_t mat, V3 origin, V3 direction, double time) {   // quad.rs:88-98
    RayD r = make_ray(origin, direction, time);
    double t, al, be;
    if (!hit_quad(q, r, 0.0, t, al, be)) return 0.0;
    HitD h;
    finish_hit<true>(sc, r, ray_at(r, t), ld3(q.n), t, mat, al, be, h);   // (every caller hands a wave-uniform material: lights_pdf)
    double area = length(cross(ld3(q.u), ld3(q.v)));
    double cos_theta = fabs(dot(r.d, h.sn));
    return (h.dist * h.dist) / (cos_theta * area);
}
// ONE: the lights list has a single entry (the Cornell box, scene 7): the index draw still happens (list.rs:82 draws it), but the
// light is the same for every lane, so its entry, transform chain and quad / sphere record arrive by scalar loads instead of four
// DEPENDENT vector gathers (lights[i] -> entries -> prims -> quads, ~700 cycles each in k_shade).
template <bool ONE>
PT_DEV V3 lights_sample_impl(const SceneD& sc, V3 origin_w, double time, Rng& rng) {
    uint32_t i = rng_index(rng, sc.n_lights);
    Entry e;
    if constexpr (ONE) e = ldu(&sc.entries[ldu(&sc.lights[0])]); else e = sc.entries[sc.lights[i]];
    V3 origin = origin_w;
    int32_t innermost = -1;
    for (int32_t k = e.inst; k >= 0;) {                                           // instance.rs:64-66, outermost instance first
        if constexpr (ONE) {
            const InstD m = ldu(&sc.insts[k]);
            origin = xform_point(m.i0, m.i1, m.i2, m.it, origin);
            innermost = k;
            k = m.inner;
        } else {
            const InstD& m = sc.insts[k];
            origin = xform_point(m.i0, m.i1, m.i2, m.it, origin);
            innermost = k;
            k = m.inner;
        }
    }
    V3 dir;
    if (e.kind == ENTRY_QUAD) {
        if constexpr (ONE) {
            const PrimRef pr = ldu(&sc.prims[e.first_prim]);
            const QuadD qd = ldu(&sc.quads[pr.index]);
            dir = sample_quad_dir(qd, origin, rng);
        } else {
            dir = sample_quad_dir(sc.quads[sc.prims[e.first_prim].index], origin, rng);
        }
    } else if (e.kind == ENTRY_CUBOID) {                                          // cuboid.rs:78-80 -> list.rs:78-84
        uint32_t j = rng_index(rng, 6u);
        dir = sample_quad_dir(sc.quads[sc.prims[e.first_prim + j].index], origin, rng);
    } else if (e.kind == ENTRY_MESH) {                                            // mesh.rs:207-209 -> :122-129
        uint32_t j = rng_index(rng, e.n_prims);
        const TriD& tr = sc.tris[sc.prims[e.first_prim + j].index];
        double u = rng_f64(rng), v = rng_f64(rng);
        double w = 1.0 - u - v;
        V3 point = ld3(tr.v0) * w + ld3(tr.v1) * u + ld3(tr.v2) * v;
        dir = normalize(point - origin);
    } else {                                                                      // sphere.rs:110-122
        const SphereD& s = sc.spheres[sc.prims[e.first_prim].index];
        double a = rng_f64(rng), b = rng_f64(rng);
        double theta = 2.0 * D_PI * a;
        double phi = dev_acos(2.0 * b - 1.0);
        const SinCos scp = dev_sincos(phi), sct = dev_sincos(theta);
        const double sp = scp.s, cp = scp.c, st = sct.s, ct = sct.c;
        V3 center = ld3(s.p1) + (ld3(s.p2) - ld3(s.p1)) * time;
        V3 point = center + V3{sp * ct, sp * st, cp} * s.r;
        dir = normalize(point - origin);
    }
    for (int32_t k = innermost; k >= 0;) {                                        // instance.rs:67-68 (not re-normalised)
        const InstD& m = sc.insts[k];
        dir = xform_vector(m.c0, m.c1, m.c2, dir);
        k = m.outer;
    }
    return dir;
}
PT_DEV V3 lights_sample(const SceneD& sc, V3 origin_w, double time, Rng& rng) {
    if (sc.n_lights == 1u) return lights_sample_impl<true>(sc, origin_w, time, rng);
    return lights_sample_impl<false>(sc, origin_w, time, rng);
}
PT_DEV double lights_pdf(const SceneD& sc, V3 origin_w, V3 direction_w, double time) {
    if (sc.n_lights == 0) return 0.0;
    double sum = 0.0;
    for (uint32_t i = 0; i < sc.n_lights; ++i) {                                  // the light index is wave-uniform: scalar loads (ldu)
        const Entry e = ldu(&sc.entries[ldu(&sc.lights[i])]);
        V3 origin = origin_w, direction = direction_w;
        for (int32_t k = e.inst; k >= 0;) {                                       // instance.rs:71-75, outermost instance first
            const InstD m = ldu(&sc.insts[k]);
            origin = xform_point(m.i0, m.i1, m.i2, m.it, origin);
            direction = xform_vector(m.i0, m.i1, m.i2, direction);
            k = m.inner;
        }
        double pdf = 0.0;
        if (e.kind == ENTRY_QUAD) {
            const PrimRef pr = ldu(&sc.prims[e.first_prim]);
            const QuadD qd = ldu(&sc.quads[pr.index]);
            pdf = pdf_quad(sc, qd, pr.mat, origin, direction, time);
        } else if (e.kind == ENTRY_CUBOID) {                                      // list.rs:86-96 over the six sides
            double s6 = 0.0;
            for (uint32_t j = 0; j < 6u; ++j) {
                const PrimRef pr = ldu(&sc.prims[e.first_prim + j]);
                const QuadD qd = ldu(&sc.quads[pr.index]);
                s6 += pdf_quad(sc, qd, pr.mat, origin, direction, time);
            }
            pdf = s6 / 6.0;
        } else if (e.kind == ENTRY_MESH) {                                        // list.rs:86-96 over ALL triangles (O(n), like the reference)
            double sn = 0.0;
            RayD r = make_ray(origin, direction, time);
            for (uint32_t j = 0; j < e.n_prims; ++j) {
                const PrimRef pr = sc.prims[e.first_prim + j];
                const TriD& tr = sc.tris[pr.index];
                double t, u, v, p = 0.0;
                if (hit_tri(tr, r, 0.0, t, u, v)) {                               // mesh.rs:131-141
                    V3 v0 = ld3(tr.v0);
                    V3 edge1 = ld3(tr.v1) - v0, edge2 = ld3(tr.v2) - v0;
                    double w = 1.0 - u - v;
                    V3 normal;
                    double tu = u, tv = v;
                    if (pr.kind & PRIM_HAS_NORMALS) {
                        const TriAttr& at = sc.tri_attr[pr.index];
                        normal = normalize(ld3(at.n[0]) * w + ld3(at.n[1]) * u + ld3(at.n[2]) * v);
                    } else {
                        normal = normalize(cross(edge1, edge2));
                    }
                    if (pr.kind & PRIM_HAS_UVS) {
                        const TriAttr& at = sc.tri_attr[pr.index];
                        tu = at.uv[0][0] * w + at.uv[1][0] * u + at.uv[2][0] * v;
                        tv = at.uv[0][1] * w + at.uv[1][1] * u + at.uv[2][1] * v;
                    }
                    HitD h;
                    finish_hit(sc, r, ray_at(r, t), normal, t, pr.mat, tu, tv, h);
                    double area = 0.5 * length(cross(edge1, edge2));
                    double cos_theta = fabs(dot(direction, h.sn));            // `direction`, not the normalised ray (mesh.rs:136)
                    p = h.dist * h.dist / (cos_theta * area);
                }
                sn += p;
            }
            pdf = sn / (double)e.n_prims;
        } else {                                                                  // sphere.rs:124-135
            const SphereD& s = sc.spheres[sc.prims[e.first_prim].index];
            RayD r = make_ray(origin, direction, time);
            double t;
            V3 c;
            if (hit_sphere(s, r, 0.0, t, c)) {
                double r2 = s.r * s.r;
                V3 center = ld3(s.p1) + (ld3(s.p2) - ld3(s.p1)) * time;
                double solid_angle = 2.0 * D_PI * sqrt(1.0 - r2 / length_squared(center - origin));
                pdf = 1.0 / solid_angle;
            }
        }
        sum += pdf;
    }
    return sum / (double)sc.n_lights;
}

// ---- camera.rs:133-168 -----------------------------------------------------------------
PT_DEV void random_offsets(Rng& rng, double& ox, double& oy) {
    uint64_t a, b;
    rng_u64x2(rng, a, b);
    double radius = sqrt(u64_to_unit(a));
    double angle = u64_to_unit(b) * 2.0 * D_PI;
    const SinCos sc_angle = dev_sincos(angle);
    const double sn = sc_angle.s, cs = sc_angle.c;
    ox = radius * cs;
    oy = radius * sn;
}
PT_DEV RayD generate_ray(const CamD& cam, uint32_t row, uint32_t col, Rng& rng) {
    double bx, by;
    random_offsets(rng, bx, by);
    bx = bx * cam.blur_strength;
    by = by * cam.blur_strength;
    V3 sample_location = ld3(cam.pixel00) + (ld3(cam.pixel_dv) * ((double)row + bx)) + (ld3(cam.pixel_du) * ((double)col + by));
    V3 origin = ld3(cam.center);
    if (cam.lens_zero) {
        rng.draw += 2;   // the two lens draws are made all the same (camera.rs:160); their products with a zero radius add nothing
    } else {
        double px, py;
        random_offsets(rng, px, py);
        origin = origin + (ld3(cam.dof_right) * px) + (ld3(cam.dof_up) * py);
    }
    double time = 0.0;
    if (cam.motionless) ++rng.draw;   // drawn all the same (camera.rs:165); no result depends on it (CamD::motionless), so its Philox block is not computed
    else time = rng_f64(rng);
    return make_ray(origin, sample_location - origin, time);
}
// camera.rs:140-151
PT_DEV V3 sample_environment(const SceneD& sc, const CamD& cam, V3 d) {
    if (!cam.env_is_map) return ld3(cam.env_color);
    double theta = dev_acos(d.y);
    double phi = dev_atan2(d.z, d.x);
    double u = (phi + D_PI) / (2.0 * D_PI);
    double v = 1.0 - theta / D_PI;
    return tex_image(sc, ldu(&sc.tex[cam.env_tex]), u, v);   // the environment's descriptor is the same for every lane
}

}  // namespace pt
