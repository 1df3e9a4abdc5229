// Scene builder + flattener + BVH builder (host). See pt_scene.h.
#include "pt_scene.h"
#include "pt_bvh_device.h"
#include "pt_detmath.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/pt_amd.h"

using namespace pt;
using namespace pt::host;

namespace pt {
static thread_local std::string g_error;
int set_error(const std::string& msg) {
    g_error = msg;
    return -1;
}
const char* last_error() { return g_error.c_str(); }
bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
const char* exp_env(const char* name) {
    const char* on = getenv("PT_EXPERIMENT");
    return (on && on[0] == '1') ? getenv(name) : nullptr;
}
void DeviceBuffers::release() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    view = SceneD{};
}
}  // namespace pt

pt_scene::~pt_scene() {
    dev.release();
    if (pool_mem) (void)hipFree(pool_mem);
    if (tile_accum) (void)hipFree(tile_accum);
    if (compact_scratch) (void)hipFree(compact_scratch);
    if (d_counters) (void)hipFree(d_counters);
    if (h_counters) (void)hipHostFree(h_counters);
}

#define TEX_RGB_OK(s, t) ((t) >= 0 && (size_t)(t) < (s)->tex.size() && (s)->tex[t].is_rgb)
#define TEX_F_OK(s, t) ((t) >= 0 && (size_t)(t) < (s)->tex.size() && !(s)->tex[t].is_rgb)
#define MAT_OK(s, m) ((m) >= 0 && (size_t)(m) < (s)->mats.size())
#define OBJ_OK(s, o) ((o) >= 0 && (size_t)(o) < (s)->objs.size())

static int push_tex(pt_scene* s, HostTex&& t) {
    s->tex.push_back(std::move(t));
    s->built = false;
    return (int)s->tex.size() - 1;
}
extern "C" int pt_tex_solid_rgb(pt_scene* s, double r, double g, double b) {
    HostTex t;
    t.d.kind = TEX_SOLID_RGB;
    t.d.v[0] = r; t.d.v[1] = g; t.d.v[2] = b;
    t.is_rgb = true;
    return push_tex(s, std::move(t));
}
extern "C" int pt_tex_solid_f(pt_scene* s, double v) {
    HostTex t;
    t.d.kind = TEX_SOLID_F;
    t.d.v[0] = v;
    return push_tex(s, std::move(t));
}
extern "C" int pt_tex_checker(pt_scene* s, double scale, int t1, int t2) {
    if (!((TEX_RGB_OK(s, t1) && TEX_RGB_OK(s, t2)) || (TEX_F_OK(s, t1) && TEX_F_OK(s, t2))))
        return set_error("pt_tex_checker: both child textures must exist and have the same value type");
    HostTex t;
    t.d.kind = TEX_CHECKER;
    t.d.t1 = (uint32_t)t1; t.d.t2 = (uint32_t)t2;
    t.d.inv_scale = 1.0 / scale;   // scale.recip() texture.rs:36
    t.is_rgb = s->tex[t1].is_rgb;
    return push_tex(s, std::move(t));
}
extern "C" int pt_tex_image_rgb8(pt_scene* s, uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (!rgb && w != 0 && h != 0) return set_error("pt_tex_image_rgb8: null pixels");
    HostTex t;
    t.d.kind = TEX_IMAGE;
    t.d.w = w; t.d.h = h;
    t.image.assign(rgb, rgb + (size_t)w * h * 3);
    t.is_rgb = true;
    return push_tex(s, std::move(t));
}
extern "C" int pt_tex_image_rgbf32(pt_scene* s, uint32_t w, uint32_t h, const float* rgb) {
    if (!rgb && w != 0 && h != 0) return set_error("pt_tex_image_rgbf32: null pixels");
    HostTex t;
    t.d.kind = TEX_IMAGE_F32;
    t.d.w = w; t.d.h = h;
    t.image_f.assign(rgb, rgb + (size_t)w * h * 3);
    t.is_rgb = true;
    return push_tex(s, std::move(t));
}
extern "C" int pt_scene_set_float_hdr(pt_scene* s, int on) {
    if (!s) return set_error("pt_scene_set_float_hdr: null scene");
    s->float_hdr = on != 0;
    return 0;
}
extern "C" int pt_scene_float_hdr(pt_scene* s) { return s && s->float_hdr ? 1 : 0; }
extern "C" int pt_register_image(pt_scene* s, const char* name, uint32_t w, uint32_t h, const uint8_t* rgb) {
    int t = pt_tex_image_rgb8(s, w, h, rgb);
    if (t < 0) return -1;
    s->images[name] = t;
    return 0;
}

static int push_mat(pt_scene* s, const MatD& m) {
    s->mats.push_back(m);
    s->built = false;
    return (int)s->mats.size() - 1;
}
static MatD blank_mat(uint32_t kind) {
    MatD m;
    memset(&m, 0, sizeof m);
    m.kind = kind;
    m.color_tex = m.rough_tex = m.nmap_tex = -1;
    return m;
}
extern "C" int pt_mat_diffuse(pt_scene* s, int color_tex, int nmap) {
    if (!TEX_RGB_OK(s, color_tex)) return set_error("pt_mat_diffuse: bad colour texture");
    MatD m = blank_mat(MAT_DIFFUSE);
    m.color_tex = color_tex;
    if (nmap >= 0) {
        if (!TEX_RGB_OK(s, nmap) || (s->tex[nmap].d.kind != TEX_IMAGE && s->tex[nmap].d.kind != TEX_IMAGE_F32)) return set_error("pt_mat_diffuse: normal map must be an image texture");
        m.nmap_tex = nmap;
    }
    return push_mat(s, m);
}
extern "C" int pt_mat_metal(pt_scene* s, int color_tex, int rough_tex) {
    if (!TEX_RGB_OK(s, color_tex) || !TEX_F_OK(s, rough_tex)) return set_error("pt_mat_metal: bad texture handle");
    MatD m = blank_mat(MAT_METAL);
    m.color_tex = color_tex;
    m.rough_tex = rough_tex;
    return push_mat(s, m);
}
extern "C" int pt_mat_glass(pt_scene* s, int color_tex, int rough_tex, double, double ior) {
    if (!TEX_RGB_OK(s, color_tex) || !TEX_F_OK(s, rough_tex)) return set_error("pt_mat_glass: bad texture handle");
    MatD m = blank_mat(MAT_GLASS);
    m.color_tex = color_tex;
    m.rough_tex = rough_tex;
    m.ior = ior;
    return push_mat(s, m);
}
extern "C" int pt_mat_principled(pt_scene* s, int color_tex, const double p[11]) {
    if (!TEX_RGB_OK(s, color_tex)) return set_error("pt_mat_principled: bad colour texture");
    MatD m = blank_mat(MAT_PRINCIPLED);
    m.color_tex = color_tex;
    for (int i = 0; i < 11; ++i) m.p[i] = p[i];
    m.ior = p[5];
    const double metallic = p[0], spec_trans = p[6], clearcoat = p[9], gloss = p[10];
    m.lobe_w[0] = (1.0 - metallic) * (1.0 - spec_trans);   // principled.rs:79-85
    m.lobe_w[1] = 1.0 - spec_trans * (1.0 - metallic);
    m.lobe_w[2] = spec_trans * (1.0 - metallic);
    m.lobe_w[3] = 0.25 * clearcoat;
    double inv_total = 1.0 / (m.lobe_w[0] + m.lobe_w[1] + m.lobe_w[2] + m.lobe_w[3]);   // :87-100
    for (int i = 0; i < 4; ++i) m.lobe_p[i] = m.lobe_w[i] * inv_total;
    m.alpha_g = (1.0 - gloss) * 0.1 + gloss * 0.001;       // :75-77
    return push_mat(s, m);
}
extern "C" int pt_mat_mix(pt_scene* s, double t, int m1, int m2) {   // MixBxDf::new mix.rs:14-20
    if (!MAT_OK(s, m1) || !MAT_OK(s, m2)) return set_error("pt_mat_mix: bad material handle");
    // a child may be a mix (MixBxDf::new takes any Arc<dyn BxDFMaterial>, mix.rs:14-20) — of leaves: the kernels evaluate two levels
    for (int c : {m1, m2})
        if (s->mats[c].kind == MAT_MIX && (s->mats[s->mats[c].color_tex].kind == MAT_MIX || s->mats[s->mats[c].rough_tex].kind == MAT_MIX))
            return set_error("pt_mat_mix: MixBxDf nested deeper than two levels is not supported");
    MatD m = blank_mat(MAT_MIX);
    m.p[0] = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);   // t.clamp(0, 1)
    m.color_tex = m1;
    m.rough_tex = m2;
    return push_mat(s, m);
}
extern "C" int pt_mat_sheen(pt_scene* s, double r, double g, double b, double sheen_tint) {   // SheenBRDF::new sheen.rs:17-22
    MatD m = blank_mat(MAT_SHEEN);
    m.p[0] = r; m.p[1] = g; m.p[2] = b; m.p[3] = sheen_tint;
    return push_mat(s, m);
}
extern "C" int pt_mat_clearcoat(pt_scene* s, double clearcoat_gloss) {   // ClearcoatBRDF::new clearcoat.rs:14-18
    MatD m = blank_mat(MAT_CLEARCOAT);
    m.alpha_g = (1.0 - clearcoat_gloss) * 0.1 + clearcoat_gloss * 0.001;
    return push_mat(s, m);
}
extern "C" int pt_mat_light(pt_scene* s, int tex) {
    if (!TEX_RGB_OK(s, tex)) return set_error("pt_mat_light: bad emission texture");
    MatD m = blank_mat(MAT_LIGHT);
    m.color_tex = tex;
    return push_mat(s, m);
}

static int push_obj(pt_scene* s, HostObj&& o) {
    s->objs.push_back(std::move(o));
    s->built = false;
    return (int)s->objs.size() - 1;
}
static QuadD make_quad(D3 q, D3 u, D3 v) {   // Quad::new quad.rs:17-36
    QuadD r;
    D3 n = cross(u, v);
    D3 normal = normalize(n);
    st3(r.q, q); st3(r.u, u); st3(r.v, v);
    st3(r.n, normal);
    r.d = dot(normal, q);
    st3(r.w, n / dot(n, n));
    return r;
}
extern "C" int pt_sphere(pt_scene* s, double radius, const double p1[3], const double p2[3], int mat) {
    if (!MAT_OK(s, mat)) return set_error("pt_sphere: bad material");
    HostObj o;
    o.kind = OBJ_SPHERE;
    o.mat = mat;
    o.sphere.r = std::fmax(radius, 0.0);   // sphere.rs:26
    for (int i = 0; i < 3; ++i) { o.sphere.p1[i] = p1[i]; o.sphere.p2[i] = p2[i]; }
    return push_obj(s, std::move(o));
}
extern "C" int pt_quad(pt_scene* s, const double q[3], const double u[3], const double v[3], int mat) {
    if (!MAT_OK(s, mat)) return set_error("pt_quad: bad material");
    HostObj o;
    o.kind = OBJ_QUAD;
    o.mat = mat;
    o.quads.push_back(make_quad(d3(q), d3(u), d3(v)));
    return push_obj(s, std::move(o));
}
extern "C" int pt_cuboid(pt_scene* s, const double a[3], const double b[3], int mat) {   // cuboid.rs:11-58
    if (!MAT_OK(s, mat)) return set_error("pt_cuboid: bad material");
    HostObj o;
    o.kind = OBJ_CUBOID;
    o.mat = mat;
    D3 mn = vmin(d3(a), d3(b)), mx = vmax(d3(a), d3(b));
    D3 dx{mx.x - mn.x, 0.0, 0.0}, dy{0.0, mx.y - mn.y, 0.0}, dz{0.0, 0.0, mx.z - mn.z};
    o.quads.push_back(make_quad(D3{mn.x, mn.y, mx.z}, dx, dy));    // front
    o.quads.push_back(make_quad(D3{mx.x, mn.y, mx.z}, -dz, dy));   // right
    o.quads.push_back(make_quad(D3{mx.x, mn.y, mn.z}, -dx, dy));   // back
    o.quads.push_back(make_quad(D3{mn.x, mn.y, mn.z}, dz, dy));    // left
    o.quads.push_back(make_quad(D3{mn.x, mx.y, mx.z}, dx, -dz));   // top
    o.quads.push_back(make_quad(D3{mn.x, mn.y, mn.z}, dx, dz));    // bottom
    return push_obj(s, std::move(o));
}
extern "C" int pt_mesh(pt_scene* s, double scale, uint32_t n_pos, const float* pos, uint32_t n_idx, const uint32_t* idx,
                       uint32_t n_nrm, const float* nrm, uint32_t n_uv, const float* uv, int mat) {   // mesh.rs:149-197
    if (!MAT_OK(s, mat)) return set_error("pt_mesh: bad material");
    if (n_idx % 3 != 0) return set_error("pt_mesh: index count must be a multiple of 3");
    if (n_idx / 3 > 0x07FFFFFFu) return set_error("pt_mesh: too many triangles");
    for (uint32_t i = 0; i < n_idx; ++i) {
        if (idx[i] >= n_pos) return set_error("pt_mesh: position index out of range");
        if (n_nrm && idx[i] >= n_nrm) return set_error("pt_mesh: normal index out of range");
        if (n_uv && idx[i] >= n_uv) return set_error("pt_mesh: texcoord index out of range");
    }
    HostObj o;
    o.kind = OBJ_MESH;
    o.mat = mat;
    o.has_normals = n_nrm != 0;
    o.has_uvs = n_uv != 0;
    auto vertex = [&](uint32_t i) { return D3{(double)pos[3 * i], (double)pos[3 * i + 1], (double)pos[3 * i + 2]} * scale; };
    o.tris.resize(n_idx / 3);
    if (o.has_normals || o.has_uvs) o.tri_attr.resize(n_idx / 3);
    for (uint32_t f = 0; f < n_idx / 3; ++f) {
        const uint32_t ii[3] = {idx[3 * f], idx[3 * f + 1], idx[3 * f + 2]};
        st3(o.tris[f].v0, vertex(ii[0]));
        st3(o.tris[f].v1, vertex(ii[1]));
        st3(o.tris[f].v2, vertex(ii[2]));
        if (!o.tri_attr.empty()) {
            TriAttr& a = o.tri_attr[f];
            memset(&a, 0, sizeof a);
            for (int k = 0; k < 3; ++k) {
                if (o.has_normals) for (int c = 0; c < 3; ++c) a.n[k][c] = (double)nrm[3 * ii[k] + c];
                if (o.has_uvs) for (int c = 0; c < 2; ++c) a.uv[k][c] = (double)uv[2 * ii[k] + c];
            }
        }
    }
    return push_obj(s, std::move(o));
}
extern "C" int pt_instance(pt_scene* s, int obj, const double axis[3], double angle, const double tr[3]) {   // instance.rs:20-30
    if (!OBJ_OK(s, obj)) return set_error("pt_instance: bad object handle");
    // the wrapped object may be wrapped again (shared geometry), may itself be an instance (nesting) and may also be placed directly
    HostObj o;
    o.kind = OBJ_INSTANCE;
    o.child = obj;
    // DQuat::from_axis_angle, DMat4::from_rotation_translation (glam 0.29 quat_to_axes)
    // detmath, not libm: g++ turns a sin/cos pair into one sincos() call, clang keeps two calls, and glibc's sincos can
    // differ from its sin and cos in the last bit on some CPUs — found by fuzzing (one instance angle in 40 scenes gave
    // a rotation matrix one ulp off the oracle's). The shared implementation makes the transform machine-independent.
    double sn, cs;
    detmath::sincos(angle * 0.5, sn, cs);
    D3 v = d3(axis) * sn;
    double qx = v.x, qy = v.y, qz = v.z, qw = cs;
    double x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    double xx = qx * x2, xy = qx * y2, xz = qx * z2;
    double yy = qy * y2, yz = qy * z2, zz = qz * z2;
    double wx = qw * x2, wy = qw * y2, wz = qw * z2;
    D3 c0{1.0 - (yy + zz), xy + wz, xz - wy};
    D3 c1{xy - wz, 1.0 - (xx + zz), yz + wx};
    D3 c2{xz + wy, yz - wx, 1.0 - (xx + yy)};
    D3 t = d3(tr);
    // analytic rigid inverse: R^T and -(R^T t)  
    D3 i0{c0.x, c1.x, c2.x}, i1{c0.y, c1.y, c2.y}, i2{c0.z, c1.z, c2.z};
    D3 it = -xform_vector(i0, i1, i2, t);
    st3(o.xf.c0, c0); st3(o.xf.c1, c1); st3(o.xf.c2, c2); st3(o.xf.t, t);
    st3(o.xf.i0, i0); st3(o.xf.i1, i1); st3(o.xf.i2, i2); st3(o.xf.it, it);
    o.xf.inner = o.xf.outer = -1;
    return push_obj(s, std::move(o));
}
static int place(pt_scene* s, int obj, std::vector<int>& list, const char* who) {
    if (!OBJ_OK(s, obj)) return set_error(std::string(who) + ": bad object handle");
    // World::add_object / add_light take any Arc<dyn Hittable> (world.rs:18-24): the same object any number of times, under instances
    // as well. Every entry of the two lists is one PLACEMENT with its own primitive ids (scene_build).
    list.push_back(obj);
    s->built = false;
    return 0;
}
extern "C" int pt_world_add_object(pt_scene* s, int obj) { return place(s, obj, s->world_objects, "pt_world_add_object"); }
extern "C" int pt_world_add_light(pt_scene* s, int obj) { return place(s, obj, s->world_lights, "pt_world_add_light"); }
extern "C" uint32_t pt_world_prim_count(pt_scene* s) { return s->n_prims; }
extern "C" int pt_world_set_device_bvh_threshold(pt_scene* s, uint32_t min_triangles) {
    if (!s) return set_error("pt_world_set_device_bvh_threshold: null scene");
    s->device_bvh_min_tris = min_triangles;
    s->built = false;
    return 0;
}
extern "C" int pt_world_device_bvh_info(pt_scene* s, uint32_t* n_meshes, uint32_t* deepest) {
    if (!s || !s->built) return set_error("pt_world_device_bvh_info: world not built");
    if (n_meshes) *n_meshes = s->n_device_blas;
    if (deepest) *deepest = s->device_blas_depth;
    return 0;
}
extern "C" int pt_world_build(pt_scene* s) { return scene_build(s); }

// ----------------------------------------------------------------------------------------
// BVH builder: top-down binned SAH (16 bins) over item boxes, depth-limited (falls back to
// object-median splits when the remaining depth budget is tight) so that the traversal
// kernel's LDS stack (TRAVERSAL_STACK levels) can never overflow. The reference's O(n^2)
// full-sweep SAH (bvh.rs:54-120) is NOT reproduced: closest-hit results do not depend on the
// tree, only on the canonical tie rule both sides share.
// ----------------------------------------------------------------------------------------
namespace {
struct BuildItem {
    Box box;
    D3 c;
    uint32_t ref_payload;   // original index
};
struct Builder {
    std::vector<BvhNode>& nodes;
    std::vector<BuildItem>& items;
    int leaf_max, max_depth;
    bool tri_leaves;                       // leaf = REF_TRIS range, else REF_ENTRY single
    std::vector<uint32_t>* order;          // BLAS: output permutation (leaf order)
    uint32_t leaf_base = 0;                // BLAS: index of this mesh's first triangle in the global array
    int depth_reached = 0;
    int n_bins = 16;                       // SAH bins per axis (<= MAX_BINS)
    size_t sweep_below = 0;                // ranges of at most this many items are split by the exact SAH sweep (every split position of every axis)
    static constexpr int MAX_BINS = 64;

    static float down(double v) {
        float f = (float)v;
        if ((double)f > v) f = std::nextafterf(f, -INFINITY);
        return f;
    }
    static float up(double v) {
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, INFINITY);
        return f;
    }
    static void store_box(const Box& b, float* lo, float* hi) {
        // pad, then round outward to f32: strictly conservative for the f64 slab test
        double m = 1e-3;
        const double v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
        for (double x : v) m = std::fmax(m, std::fabs(x));
        double pad = 1e-7 * m;
        lo[0] = down(b.lo.x - pad); lo[1] = down(b.lo.y - pad); lo[2] = down(b.lo.z - pad);
        hi[0] = up(b.hi.x + pad); hi[1] = up(b.hi.y + pad); hi[2] = up(b.hi.z + pad);
    }
    static double axis_of(D3 v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

    uint32_t make_leaf(size_t begin, size_t end) {
        if (tri_leaves) {
            uint32_t first = leaf_base + (uint32_t)order->size();
            for (size_t i = begin; i < end; ++i) order->push_back(items[i].ref_payload);
            return REF_TRIS | ((uint32_t)(end - begin - 1) << 27) | first;
        }
        return REF_ENTRY | items[begin].ref_payload;
    }
    // returns child reference; `out_box` = bounds of the subtree
    uint32_t build(size_t begin, size_t end, int depth, Box& out_box) {
        depth_reached = std::max(depth_reached, depth);
        Box box, cbox;
        for (size_t i = begin; i < end; ++i) { box.grow(items[i].box); cbox.grow(items[i].c); }
        out_box = box;
        const size_t n = end - begin;
        if (n <= (size_t)leaf_max) return make_leaf(begin, end);
        // depth budget: levels still needed with perfect median splits
        int needed = 0;
        for (size_t cap = (size_t)leaf_max; cap < n; cap *= 2) ++needed;
        const bool force_median = needed >= max_depth - depth;
        D3 ext = cbox.hi - cbox.lo;
        int axis = ext.x >= ext.y && ext.x >= ext.z ? 0 : (ext.y >= ext.z ? 1 : 2);
        size_t mid = begin + n / 2;
        bool done = false;
        if (!force_median && n <= sweep_below) {
            // exact sweep: the items sorted along each axis, every position between two neighbours a candidate
            double best_cost = INFINITY;
            int best_axis = -1;
            size_t best_pos = 0;
            std::vector<BuildItem> tmp(items.begin() + begin, items.begin() + end);
            std::vector<double> right_area(n);
            for (int a = 0; a < 3; ++a) {
                std::stable_sort(tmp.begin(), tmp.end(), [&](const BuildItem& x, const BuildItem& y) { return axis_of(x.c, a) < axis_of(y.c, a); });
                Box acc;
                for (size_t i = n; i-- > 1;) { acc.grow(tmp[i].box); right_area[i] = acc.half_area(); }
                acc = Box();
                for (size_t i = 0; i + 1 < n; ++i) {
                    acc.grow(tmp[i].box);
                    const double cost = acc.half_area() * (double)(i + 1) + right_area[i + 1] * (double)(n - i - 1);
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_pos = i + 1; }
                }
            }
            if (best_axis >= 0) {
                std::stable_sort(items.begin() + begin, items.begin() + end,
                                 [&](const BuildItem& x, const BuildItem& y) { return axis_of(x.c, best_axis) < axis_of(y.c, best_axis); });
                mid = begin + best_pos;
                done = true;
            }
        }
        if (!force_median && !done) {
            const int NB = n_bins;
            double best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                double lo = axis_of(cbox.lo, a), hi = axis_of(cbox.hi, a);
                if (!(hi > lo)) continue;
                Box bb[MAX_BINS];
                size_t bc[MAX_BINS] = {0};
                double k = (double)NB / (hi - lo);
                for (size_t i = begin; i < end; ++i) {
                    int b = (int)((axis_of(items[i].c, a) - lo) * k);
                    b = std::min(std::max(b, 0), NB - 1);
                    bb[b].grow(items[i].box);
                    ++bc[b];
                }
                Box right_acc[MAX_BINS];
                size_t right_cnt[MAX_BINS];
                Box acc;
                size_t cnt = 0;
                for (int b = NB - 1; b > 0; --b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    right_acc[b] = acc;
                    right_cnt[b] = cnt;
                }
                acc = Box();
                cnt = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = acc.half_area() * (double)cnt + right_acc[b + 1].half_area() * (double)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                double lo = axis_of(cbox.lo, best_axis), hi = axis_of(cbox.hi, best_axis);
                double k = (double)NB / (hi - lo);
                auto it = std::stable_partition(items.begin() + begin, items.begin() + end, [&](const BuildItem& it_) {
                    int b = (int)((axis_of(it_.c, best_axis) - lo) * k);
                    b = std::min(std::max(b, 0), NB - 1);
                    return b <= best_bin;
                });
                mid = (size_t)(it - items.begin());
                done = mid > begin && mid < end;
            }
        }
        if (!done) {
            mid = begin + n / 2;
            std::stable_sort(items.begin() + begin, items.begin() + end,
                             [&](const BuildItem& a, const BuildItem& b) { return axis_of(a.c, axis) < axis_of(b.c, axis); });
        }
        uint32_t me = (uint32_t)nodes.size();
        nodes.emplace_back();
        Box lb, rb;
        uint32_t l = build(begin, mid, depth + 1, lb);
        uint32_t r = build(mid, end, depth + 1, rb);
        BvhNode& nd = nodes[me];
        store_box(lb, nd.lo0, nd.hi0);
        store_box(rb, nd.lo1, nd.hi1);
        nd.child0 = l;
        nd.child1 = r;
        nd.pad0 = nd.pad1 = 0;
        return REF_NODE | me;
    }
};

template <class T>
bool upload(DeviceBuffers& dev, const std::vector<T>& v, const T*& out) {
    out = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void* p = nullptr;
    if (!hip_ok(hipMalloc(&p, bytes), "hipMalloc(scene)")) return false;
    dev.allocs.push_back(p);
    if (!v.empty() && !hip_ok(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(scene)")) return false;
    out = (const T*)p;
    return true;
}
// max |coordinate| of a subtree's bounds, as the f32 slab test's error margin needs it: padded like
// the stored boxes and rounded up
float box_extent(const Box& b) {
    double m = 1e-3;
    const double v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
    for (double x : v) m = std::fmax(m, std::fabs(x));
    return std::nextafterf((float)(m * 1.000001), INFINITY);
}
Box sphere_box(const SphereD& s) {
    Box b;
    D3 r{s.r, s.r, s.r};
    b.grow(d3(s.p1) - r); b.grow(d3(s.p1) + r);
    b.grow(d3(s.p2) - r); b.grow(d3(s.p2) + r);
    return b;
}
Box quad_box(const QuadD& q) {
    Box b;
    D3 o = d3(q.q), u = d3(q.u), v = d3(q.v);
    b.grow(o); b.grow(o + u); b.grow(o + v); b.grow(o + u + v);
    return b;
}
Box xform_box(const Box& b, const InstD& m) {
    Box r;
    for (int i = 0; i < 8; ++i) {
        D3 p{(i & 1) ? b.hi.x : b.lo.x, (i & 2) ? b.hi.y : b.lo.y, (i & 4) ? b.hi.z : b.lo.z};
        r.grow(xform_point(d3(m.c0), d3(m.c1), d3(m.c2), d3(m.t), p));
    }
    return r;
}
}  // namespace

int pt::scene_build(pt_scene* s) {
    if (!s->ctx) return set_error("scene has no context");
    if (!hip_ok(hipSetDevice(s->ctx->device), "hipSetDevice")) return -1;
    s->dev.release();
    s->built = false;

    std::vector<BvhNode> nodes;
    std::vector<Entry> entries;
    std::vector<PrimRef> prims;
    std::vector<SphereD> spheres;
    std::vector<QuadD> quads;
    std::vector<TriD> tris;
    std::vector<TriAttr> tri_attr;
    std::vector<uint32_t> tri_gid;
    std::vector<InstD> insts;
    std::vector<uint32_t> lights;
    std::vector<BuildItem> tlas_items;
    bool any_attr = false;
    for (auto& o : s->objs) any_attr = any_attr || !o.tri_attr.empty();

    // canonical global primitive ids: lights list first, then objects, insertion order
    std::vector<int> order = s->world_lights;
    order.insert(order.end(), s->world_objects.begin(), s->world_objects.end());
    if (order.empty()) return set_error("pt_world_build: the world is empty");
    int max_blas_depth = 0;
    s->n_device_blas = s->device_blas_depth = 0;
    struct SharedBlas {   // one tree and one triangle range per MESH OBJECT, however many placements it has
        uint32_t root, tri_base;
        float extent;
        Box local;
        std::vector<uint32_t> face_pos;   // face -> position in BLAS (leaf) order
    };
    std::map<int, SharedBlas> blas_of;
    for (size_t wi = 0; wi < order.size(); ++wi) {
        // the placement's instance chain, outermost first, down to the object it finally wraps
        int oi = order[wi];
        int inst = -1;
        std::vector<int> chain;
        for (int guard = 0; s->objs[oi].kind == OBJ_INSTANCE; ++guard) {
            if (guard > 64) return set_error("pt_world_build: instance chain too deep");
            chain.push_back((int)insts.size());
            insts.push_back(s->objs[oi].xf);
            oi = s->objs[oi].child;
        }
        for (size_t k = 0; k < chain.size(); ++k) {
            insts[chain[k]].outer = k == 0 ? -1 : chain[k - 1];
            insts[chain[k]].inner = k + 1 < chain.size() ? chain[k + 1] : -1;
        }
        if (!chain.empty()) inst = chain.front();
        const HostObj* o = &s->objs[oi];
        auto to_world = [&](D3 p) {   // object space -> world: innermost instance first
            for (size_t k = chain.size(); k-- > 0;) {
                const InstD& m = insts[chain[k]];
                p = xform_point(d3(m.c0), d3(m.c1), d3(m.c2), d3(m.t), p);
            }
            return p;
        };
        const bool is_light = wi < s->world_lights.size();
        if (is_light) lights.push_back((uint32_t)entries.size());   // any hittable may be a light (world.rs:18-20)
        Entry e{};
        memset(&e, 0, sizeof e);
        e.first_prim = (uint32_t)prims.size();
        e.inst = inst;
        e.blas_root = REF_EMPTY;
        Box local;
        switch (o->kind) {
        case OBJ_SPHERE:
            e.kind = ENTRY_SPHERE;
            prims.push_back(PrimRef{PRIM_SPHERE, (uint32_t)spheres.size(), (uint32_t)o->mat, inst});
            spheres.push_back(o->sphere);
            local = sphere_box(o->sphere);
            break;
        case OBJ_QUAD:
        case OBJ_CUBOID:
            e.kind = o->kind == OBJ_QUAD ? ENTRY_QUAD : ENTRY_CUBOID;
            for (const QuadD& q : o->quads) {
                prims.push_back(PrimRef{PRIM_QUAD, (uint32_t)quads.size(), (uint32_t)o->mat, inst});
                quads.push_back(q);
                local.grow(quad_box(q));
            }
            break;
        case OBJ_MESH: {
            e.kind = ENTRY_MESH;
            if (o->tris.empty()) return set_error("pt_world_build: empty mesh");
            auto it = blas_of.find(oi);
            if (it == blas_of.end()) {
                std::vector<BuildItem> items(o->tris.size());
                SharedBlas sb;
                for (size_t i = 0; i < o->tris.size(); ++i) {
                    Box b;
                    b.grow(d3(o->tris[i].v0)); b.grow(d3(o->tris[i].v1)); b.grow(d3(o->tris[i].v2));
                    items[i] = BuildItem{b, b.centroid(), (uint32_t)i};
                    sb.local.grow(b);
                }
                std::vector<uint32_t> perm;
                sb.tri_base = (uint32_t)tris.size();
                if ((size_t)sb.tri_base + o->tris.size() > 0x07FFFFFFu) return set_error("pt_world_build: too many triangles");
                int leaf_max = 4;   // PT_LEAF_MAX: experiments only (smaller leaves were slower on scene 6)
                if (const char* ev = exp_env("PT_LEAF_MAX")) leaf_max = std::min(8, std::max(1, atoi(ev)));
                Box bb = sb.local;
                bool on_device = false;
                if (s->device_bvh_min_tris != 0 && o->tris.size() >= s->device_bvh_min_tris) {
                    // large mesh: depth-bounded LBVH on the GPU (pt_bvh_device.hip); only a HIP failure falls back to the host builder
                    DeviceBlas db;
                    const double lo[3] = {sb.local.lo.x, sb.local.lo.y, sb.local.lo.z}, hi[3] = {sb.local.hi.x, sb.local.hi.y, sb.local.hi.z};
                    // depth budget: eight levels above a perfectly balanced tree — the radix tree of 1.3 M triangles comes out 27 deep, and
                    // every level the bound takes away bends Morton splits (measured on that mesh, K2 per launch: 27 levels 1.78 ms,
                    // 23: 1.83, 21: 2.19, 20: 2.50; the host's SAH tree 1.50) — within what the traversal stacks cover (k_extend2's
                    // largest LDS stack holds 32 entries; TRAVERSAL_STACK bounds top level + mesh tree)
                    int bal = 0;
                    for (size_t cap = (size_t)leaf_max; cap < o->tris.size(); cap *= 2) ++bal;
                    int dev_depth = std::min(29, std::max((int)MAX_BLAS_DEPTH, bal + 8));
                    if (const char* ev = exp_env("PT_LBVH_DEPTH")) dev_depth = std::min(29, std::max(bal, atoi(ev)));
                    uint32_t median_below = 0;
                    if (const char* ev = exp_env("PT_LBVH_MEDIAN")) median_below = (uint32_t)atoi(ev);
                    if (build_blas_device(o->tris.data(), (uint32_t)o->tris.size(), lo, hi, (uint32_t)leaf_max, (uint32_t)dev_depth, median_below, db, s->ctx->stream)) {
                        const uint32_t node_base = (uint32_t)nodes.size();
                        auto fix = [&](uint32_t ref) {
                            if ((ref & REF_TYPE_MASK) == REF_NODE) return REF_NODE | (node_base + ref);
                            return (ref & ~0x07FFFFFFu) | (sb.tri_base + (ref & 0x07FFFFFFu));     // REF_TRIS: count bits kept
                        };
                        for (BvhNode nd : db.nodes) {
                            nd.child0 = fix(nd.child0);
                            nd.child1 = fix(nd.child1);
                            nodes.push_back(nd);
                        }
                        sb.root = REF_NODE | node_base;
                        perm = db.order;
                        max_blas_depth = std::max(max_blas_depth, db.depth);
                        ++s->n_device_blas;
                        s->device_blas_depth = std::max<uint32_t>(s->device_blas_depth, (uint32_t)db.depth);
                        on_device = true;
                    }
                }
                if (!on_device) {
                    Builder bl{nodes, items, leaf_max, MAX_BLAS_DEPTH, true, &perm, sb.tri_base};
                    if (const char* ev = exp_env("PT_SAH_BINS")) bl.n_bins = std::min((int)Builder::MAX_BINS, std::max(2, atoi(ev)));
                    if (const char* ev = exp_env("PT_SAH_SWEEP")) bl.sweep_below = (size_t)std::max(0, atoi(ev));
                    sb.root = bl.build(0, items.size(), 0, bb);
                    max_blas_depth = std::max(max_blas_depth, bl.depth_reached);
                }
                sb.extent = box_extent(bb);
                sb.face_pos.resize(perm.size());
                for (size_t k = 0; k < perm.size(); ++k) {
                    const uint32_t face = perm[k];
                    sb.face_pos[face] = (uint32_t)k;
                    tris.push_back(o->tris[face]);
                    tri_gid.push_back(face);                 // face index inside the mesh; the kernels add Entry::first_prim
                    if (any_attr) tri_attr.push_back(o->tri_attr.empty() ? TriAttr{} : o->tri_attr[face]);
                }
                it = blas_of.emplace(oi, std::move(sb)).first;
            }
            const SharedBlas& sb = it->second;
            e.blas_root = sb.root;
            e.extent = sb.extent;
            local = sb.local;
            const uint32_t flags = PRIM_TRI | (o->has_normals ? PRIM_HAS_NORMALS : 0u) | (o->has_uvs ? PRIM_HAS_UVS : 0u);
            for (size_t face = 0; face < o->tris.size(); ++face)   // one PrimRef per PLACED triangle: ids are per placement
                prims.push_back(PrimRef{flags, sb.tri_base + sb.face_pos[face], (uint32_t)o->mat, inst});
            break;
        }
        default:
            return set_error("pt_world_build: unsupported object kind");
        }
        e.n_prims = (uint32_t)prims.size() - e.first_prim;
        Box world = local;
        for (size_t k = chain.size(); k-- > 0;) world = xform_box(world, insts[chain[k]]);   // box of the box, per instance (aabb.rs:50-76)
        if (!chain.empty() && o->kind == OBJ_MESH) {
            // an instanced mesh gets the bounds of its TRANSFORMED VERTICES, not the box of its transformed box
            // (aabb.rs:50-76 does the latter): a mesh turned by ~50 degrees has a third less box to enter, and
            // every ray that enters costs a trip through the mesh pass. Any conservative box gives the same
            // closest hit; store_box pads by 1e-7 relative, far above the rounding of the transform.
            Box tight;
            for (const TriD& t : o->tris)
                for (const double* v : {t.v0, t.v1, t.v2}) tight.grow(to_world(d3(v)));
            world = tight;
        }
        tlas_items.push_back(BuildItem{world, world.centroid(), (uint32_t)entries.size()});
        entries.push_back(e);
    }
    if (prims.size() >= (size_t)HIT_ID_MASK - 4) return set_error("pt_world_build: too many primitives (28-bit ids)");
    for (PrimRef& pr : prims) pr.kind |= (uint32_t)s->mats[pr.mat].kind << PRIM_MAT_KIND_SHIFT;
    std::vector<Box> entry_boxes(tlas_items.size());
    for (const BuildItem& it : tlas_items) entry_boxes[it.ref_payload] = it.box;
    // Build the TLAS over world entries (one entry per leaf).
    Builder tl{nodes, tlas_items, 1, MAX_TLAS_DEPTH, false, nullptr};
    Box wb;
    uint32_t tlas_root = tl.build(0, tlas_items.size(), 0, wb);
    if (tl.depth_reached + 1 + max_blas_depth + 1 > TRAVERSAL_STACK) return set_error("pt_world_build: BVH too deep for the traversal stack");
    s->stack_need = (uint32_t)(tl.depth_reached + 1 + max_blas_depth + 1);
    if (exp_env("PT_VERBOSE")) fprintf(stderr, "[pt] BVH: top-level depth %d, deepest mesh tree %d, %zu nodes, %zu triangles\n", tl.depth_reached, max_blas_depth, nodes.size(), tris.size());

    // texture atlas
    std::vector<TexD> tex(s->tex.size());
    std::vector<uint8_t> atlas;
    std::vector<float> atlas_f;
    for (size_t i = 0; i < s->tex.size(); ++i) {
        tex[i] = s->tex[i].d;
        if (tex[i].kind == TEX_IMAGE) {
            tex[i].ofs = atlas.size();
            atlas.insert(atlas.end(), s->tex[i].image.begin(), s->tex[i].image.end());
        } else if (tex[i].kind == TEX_IMAGE_F32) {
            tex[i].ofs = atlas_f.size();
            atlas_f.insert(atlas_f.end(), s->tex[i].image_f.begin(), s->tex[i].image_f.end());
        }
    }
    for (TexD& t : tex)   // checkers of two solid children carry the children's values
        if (t.kind == TEX_CHECKER) {
            const TexD &a = tex[t.t1], &b = tex[t.t2];
            const bool solid = (a.kind == TEX_SOLID_RGB || a.kind == TEX_SOLID_F) && (b.kind == TEX_SOLID_RGB || b.kind == TEX_SOLID_F);
            t.flat = solid ? 1u : 0u;
            for (int c = 0; c < 3; ++c) { t.c1[c] = solid ? a.v[c] : 0.0; t.c2[c] = solid ? b.v[c] : 0.0; }
        }
    std::vector<MatD> mats = s->mats;   // solid textures' values into the material records (MatD::color_solid)
    for (MatD& m : mats) {
        m.color_solid = m.rough_solid = 0u;
        const bool has_color = m.kind == MAT_DIFFUSE || m.kind == MAT_METAL || m.kind == MAT_PRINCIPLED || m.kind == MAT_LIGHT;
        const bool has_rough = m.kind == MAT_METAL || m.kind == MAT_GLASS;
        if (has_color && m.color_tex >= 0 && (size_t)m.color_tex < tex.size() && tex[m.color_tex].kind == TEX_SOLID_RGB && !exp_env("PT_NO_SOLID_IN_MAT")) {
            m.color_solid = 1u;
            for (int c = 0; c < 3; ++c) m.color_v[c] = tex[m.color_tex].v[c];
        }
        if (has_rough && m.rough_tex >= 0 && (size_t)m.rough_tex < tex.size() && tex[m.rough_tex].kind == TEX_SOLID_F && !exp_env("PT_NO_SOLID_IN_MAT")) {
            m.rough_solid = 1u;
            m.rough_v = tex[m.rough_tex].v[0];
        }
    }
    std::vector<CuboidBox> cuboid_box;
    std::vector<EntryBox> entry_box;   // tlas_items[i] is entry i (built in entry order, before the builder permutes them)
    for (int pass = 0; pass < 2; ++pass)   // spheres / quads / cuboids first: their hits trim the mesh boxes
        for (size_t i = 0; i < entry_boxes.size(); ++i) {
            if ((entries[i].kind == ENTRY_MESH) != (pass == 1)) continue;
            EntryBox eb{};
            const Entry& e = entries[i];
            Builder::store_box(entry_boxes[i], eb.lo, eb.hi);
            eb.entry = (uint32_t)i;
            eb.kind = e.kind;
            eb.first_prim = e.first_prim;
            eb.inst = e.inst;
            eb.blas_root = e.blas_root;
            eb.extent = e.extent;
            eb.n_prims = e.n_prims;
            eb.prim_kind = ENTRYBOX_VIA_PRIMREF;
            if (e.kind != ENTRY_MESH) {   // one kind, consecutive records: the walk addresses them without prims[]
                const PrimRef& p0 = prims[e.first_prim];
                bool direct = true;
                for (uint32_t k = 0; k < e.n_prims; ++k) {
                    const PrimRef& pk = prims[e.first_prim + k];
                    direct = direct && (pk.kind & 0xFFu) == (p0.kind & 0xFFu) && pk.index == p0.index + k && pk.inst == e.inst;
                }
                if (direct && !exp_env("PT_ENTRYBOX_VIA_PRIMREF")) {
                    eb.prim_kind = p0.kind & 0xFFu;
                    eb.prim_index = p0.index;
                }
            }
            CuboidBox cb{};
            if (e.kind == ENTRY_CUBOID && eb.prim_kind == PRIM_QUAD) {   // object-space box of the six faces (they are consecutive QuadD records)
                Box local;
                for (uint32_t k = 0; k < 6; ++k) local.grow(quad_box(quads[eb.prim_index + k]));
                Builder::store_box(local, cb.lo, cb.hi);
                eb.extent = box_extent(local);
            }
            cuboid_box.push_back(cb);
            entry_box.push_back(eb);
        }
    SceneD v{};
    DeviceBuffers& dev = s->dev;
    bool ok = upload(dev, nodes, v.nodes) && upload(dev, entries, v.entries) && upload(dev, prims, v.prims) &&
              upload(dev, spheres, v.spheres) && upload(dev, quads, v.quads) && upload(dev, tris, v.tris) &&
              upload(dev, tri_gid, v.tri_gid) && upload(dev, insts, v.insts) && upload(dev, tex, v.tex) &&
              upload(dev, mats, v.mats) && upload(dev, atlas, v.atlas) && upload(dev, atlas_f, v.atlas_f) && upload(dev, lights, v.lights) && upload(dev, entry_box, v.entry_box) && upload(dev, cuboid_box, v.cuboid_box);
    if (ok && any_attr) ok = upload(dev, tri_attr, v.tri_attr);
    if (!ok) {
        dev.release();
        return -1;
    }
    v.tlas_root = tlas_root;
    v.tlas_extent = box_extent(wb);
    v.n_entries = (uint32_t)entries.size();
    v.n_prims = (uint32_t)prims.size();
    v.n_lights = (uint32_t)lights.size();
    uint32_t flat_max = TLAS_FLAT_MAX;
    if (const char* ev = exp_env("PT_FLAT_MAX")) flat_max = (uint32_t)atoi(ev);
    bool pair_ids_fit = true;   // the flat walk packs (primitive id, lane) into one word: ids of spheres / quads / cuboid faces below 2^26
    for (const Entry& e : entries) pair_ids_fit = pair_ids_fit && (e.kind == ENTRY_MESH || (uint64_t)e.first_prim + e.n_prims <= (1ull << 26));
    v.tlas_flat = entries.size() <= flat_max && pair_ids_fit && !exp_env("PT_NO_FLAT_TLAS") ? 1u : 0u;
    v.flat_pairs = 0;
    for (const Entry& e : entries) v.flat_pairs |= (v.tlas_flat && e.kind == ENTRY_CUBOID && !exp_env("PT_NO_FLAT_PAIRS")) ? 1u : 0u;
    s->stack_need_extend2 = v.tlas_flat ? (uint32_t)(max_blas_depth + 1) : s->stack_need;
    dev.view = v;
    s->n_prims = v.n_prims;
    s->n_mesh_entries = 0;
    s->motionless = true;
    for (const SphereD& sp : spheres)
        for (int i = 0; i < 3; ++i) s->motionless = s->motionless && sp.p1[i] == sp.p2[i];
    for (const Entry& e : entries) s->n_mesh_entries += e.kind == ENTRY_MESH ? 1u : 0u;
    s->built = true;
    return 0;
}
