// Wavefront path-tracing stages for gfx950 (MI355X). One lane = one resident path.
//
//   k_init     K1 raygen       camera.rs:153-168  fills the pool with the first sample of every slot
//   k_extend2  K2 closest hit  world.rs:47-62 -> bvh.rs:123-164 -> sphere/quad/mesh/instance.rs  (two-phase form; k_extend =
//                              batch form for scenes without meshes)
//   k_shade    K3+K4+K1'       camera.rs:177-226 body: miss/env, emission, RR, one-sample MIS,
//                              BSDF sample+pdf+eval, next ray; finished paths are regenerated in place
//   k_resolve  K6 (sum part)   camera.rs:106-109: per-pixel sum of the slot accumulators
//
// All kernels are persistent-thread style: a fixed grid sized to the machine walks the pool
// (grid-stride, or windows drawn from a queue). No MFMA anywhere — this is branchy f64 scalar work bounded by
// VALU issue and L2 latency, with the path pool streaming through HBM once per stage.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pt_dev_geom.h"
#include "pt_kernels.h"

namespace pt {

constexpr int BLOCK = 256;
constexpr int SORT_WINDOW_SLOTS = 2048;   // window size of k_extend2 and k_shade = the granule pt_render.cpp allocates the pool in
// Resident blocks per CU the batch form of K2 is compiled for. Its register count sits right at the 128-register step
// (4 waves per SIMD) and tipped over it with unrelated edits elsewhere in this file: measured on one build pair, scene 3's
// K2 was 10 % faster at four blocks than at three, scene 5's 16 % — so the bound is stated instead of left to chance.
#ifndef PT_EXTEND_BATCH_BLOCKS
#define PT_EXTEND_BATCH_BLOCKS 4
#endif

// K2 writes its result (one primitive id per slot) once and never re-reads it, while the scene tables
// (BVH, primitives: a few MB) are re-read by every wave: the result leaves with non-temporal stores.
// (Non-temporal LOADS of the path records made no measurable difference and are not used.)
template <class T> PT_DEV void stnt(T* p, T v) { __builtin_nontemporal_store(v, p); }
typedef __attribute__((address_space(3))) void* lds_ptr;      // operands of __builtin_amdgcn_global_load_lds (LDS-DMA)
typedef const __attribute__((address_space(1))) void* glb_ptr;

// ---- path records (pt_types.h RayRec / PathRec): one lane moves one whole record, 16 B per access ----
typedef double d2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
// PoolD::compact (scenes in which nothing moves, CamD::motionless: Ray::time reaches no result): the ray record's time slot
// carries (pixel, bounce number) instead, so a path at bounce 0 — throughput (1,1,1) by definition — has no PathRec worth
// writing: a regenerated camera ray costs one 64-byte record, not 96 bytes.
PT_DEV RayD load_ray(const PoolD& pool, uint32_t s) {
    const d2v* p = reinterpret_cast<const d2v*>(&pool.ray[s]);
    const d2v a = p[0], b = p[1], c = p[2], d = p[3];
    return RayD{V3{a.x, a.y, b.x}, V3{b.y, c.x, c.y}, pool.compact ? 0.0 : d.x};
}
// `tail`: the record's words 12, 13 — the bits of Ray::time, or (pixel, bounce) in compact mode
PT_DEV RayD load_ray(const PoolD& pool, uint32_t s, uint32_t& sample, uint32_t& draw, uint32_t (&tail)[2]) {
    const d2v* p = reinterpret_cast<const d2v*>(&pool.ray[s]);
    const d2v a = p[0], b = p[1], c = p[2];
    const u4v d = *reinterpret_cast<const u4v*>(p + 3);
    sample = d.z;
    draw = d.w;
    tail[0] = d.x;
    tail[1] = d.y;
    return RayD{V3{a.x, a.y, b.x}, V3{b.y, c.x, c.y}, pool.compact ? 0.0 : __hiloint2double((int)d.y, (int)d.x)};
}
PT_DEV void store_ray(const PoolD& pool, uint32_t s, const RayD& r, uint32_t sample, uint32_t draw, uint32_t pixel, uint32_t bounce) {
    d2v* p = reinterpret_cast<d2v*>(&pool.ray[s]);
    p[0] = d2v{r.o.x, r.o.y};
    p[1] = d2v{r.o.z, r.d.x};
    p[2] = d2v{r.d.y, r.d.z};
    *reinterpret_cast<u4v*>(p + 3) = pool.compact ? u4v{pixel, bounce, sample, draw}
                                                  : u4v{(uint32_t)__double2loint(r.time), (uint32_t)__double2hiint(r.time), sample, draw};
}
PT_DEV V3 load_path(const PoolD& pool, uint32_t s, uint32_t& pixel, uint32_t& bounce) {
    const d2v* p = reinterpret_cast<const d2v*>(&pool.path[s]);
    const d2v a = p[0];
    const u4v b = *reinterpret_cast<const u4v*>(p + 1);
    pixel = b.z;
    bounce = b.w;
    return V3{a.x, a.y, __hiloint2double((int)b.y, (int)b.x)};
}
PT_DEV void store_path(const PoolD& pool, uint32_t s, V3 thr, uint32_t pixel, uint32_t bounce) {
    d2v* p = reinterpret_cast<d2v*>(&pool.path[s]);
    p[0] = d2v{thr.x, thr.y};
    *reinterpret_cast<u4v*>(p + 1) = u4v{(uint32_t)__double2loint(thr.z), (uint32_t)__double2hiint(thr.z), pixel, bounce};
#if PT_PATHREC_BYTES == 64
    p[2] = d2v{0.0, 0.0};                         // the record is one 64-B sector: write all of it
    p[3] = d2v{0.0, 0.0};
#endif
}

// ---------------------------------------------------------------------------------------
// Closest-hit traversal. Two-level BVH2 walked with one per-lane stack held in LDS
// (stack[level][lane]: a wave touches 64 consecutive dwords per level -> conflict free).
// The result is tree-independent: minimum t; on an exact tie the larger global primitive id
// wins, so any builder/visit order gives the reference's hit.
// ---------------------------------------------------------------------------------------
struct Closest {
    double t;
    uint32_t id;
};
PT_DEV void consider(Closest& best, double t, uint32_t id) {
    if (t < best.t || (t == best.t && id > best.id)) {
        best.t = t;
        best.id = id;
    }
}
// Sum of a per-thread tally over the wave (every lane gets it). The kernels' end-of-launch counters (samples, segments, alive) are added
// once per WAVE instead of once per thread (131 k to 262 k atomics on a single address at the end of every launch). Measured +-0 on
// every pool size — those atomics return nothing and nobody waits for them — unlike k_compact_scan's, whose returns the waves did wait for.
PT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
// K2's result word for a slot (pt_types.h, PoolD::hit_prim): id | class << 28. `id` may be one of the sentinels.
PT_DEV uint32_t hit_word(const SceneD& sc, uint32_t id) {
    if (id >= HIT_SLOT_DEAD) {
        const uint32_t cls = id == HIT_NONE ? CLASS_MISS : id == HIT_SLOT_IDLE ? CLASS_IDLE : CLASS_DEAD;
        return (cls << HIT_CLASS_SHIFT) | HIT_ID_MASK;
    }
    const uint32_t mat_kind = (sc.prims[id].kind >> PRIM_MAT_KIND_SHIFT) & 0xFFu;
    return ((1u + mat_kind) << HIT_CLASS_SHIFT) | id;
}
PT_DEV uint32_t dead_or_idle(uint32_t bounce) { return bounce == SLOT_IDLE ? HIT_SLOT_IDLE : HIT_SLOT_DEAD; }

// ---- conservative f32 slab test -----------------------------------------------------------------
// Per ray and per space (world / instance-local) the f64 ray is reduced to idf = fl32(1/d),
// oif = fl32(o/d) and t' = fma32(b, idf, -oif) for a box bound b. Error analysis (u = 2^-24):
//   t' = (b*id*(1+da) - oi*(1+db))*(1+dc)  =>  |t' - t| <= 2u (|b||id| + |oi|) <= 2u (S|id| + |oi|)
// with S = max |coordinate| of the boxes of the tree being walked (SceneD::tlas_extent /
// Entry::extent). 1/d itself is a fast f32 reciprocal of fl32(d) (<= 3u relative error: the slabs
// of a ray tilted by 3u, another 3u (S|id| + |oi|)). Every axis interval is widened by
// e = 8u (S|id| + |oi|), which also covers the rounding of e itself and of the +-e — so a box the
// exact ray touches inside [t_min, t_best] is never rejected. 1/d is clamped to +-1e30 so that an exactly axis-parallel ray
// (they occur: a direction sampled inside the plane of an axis-aligned light has d.y == 0) yields
// finite products: the axis then behaves as "parallel" — everything when the origin is inside the
// slab, nothing when it is outside. Box tests never influence WHICH hit wins, only how much work
// it takes to find it; the primitive tests keep the reference's f64 arithmetic.
struct RayF {
    float idx, idy, idz, oix, oiy, oiz, ex, ey, ez;
};
PT_DEV void rayf_axis(double o, double d, float S, float& idf, float& oif, float& e) {
    // idf: f32 reciprocal of fl32(d) (relative error <= 3u against 1/d), clamped to +-1e30; the slab
    // parameters are then those of a ray whose direction differs by <= 3u — absorbed by the margin
    float df = (float)d;
    float id = __frcp_rn(df);
    if (!(fabsf(id) <= 1e30f)) id = copysignf(1e30f, df);
    idf = id;
    oif = (float)(o * (double)id);
    e = (S * fabsf(id) + fabsf(oif)) * 4.7683716e-07f;   // 8u >= (2u arithmetic + 3u reciprocal) with slack
}
PT_DEV RayF make_rayf(V3 o, V3 d, float S) {
    RayF f;
    rayf_axis(o.x, d.x, S, f.idx, f.oix, f.ex);
    rayf_axis(o.y, d.y, S, f.idy, f.oiy, f.ey);
    rayf_axis(o.z, d.z, S, f.idz, f.oiz, f.ez);
    return f;
}
PT_DEV bool slab_f32(const float* lo, const float* hi, const RayF& f, float t_min, float t_max, float& t_near) {
    const float t1x = __builtin_fmaf(lo[0], f.idx, -f.oix), t2x = __builtin_fmaf(hi[0], f.idx, -f.oix);
    const float t1y = __builtin_fmaf(lo[1], f.idy, -f.oiy), t2y = __builtin_fmaf(hi[1], f.idy, -f.oiy);
    const float t1z = __builtin_fmaf(lo[2], f.idz, -f.oiz), t2z = __builtin_fmaf(hi[2], f.idz, -f.oiz);
    const float nx = fminf(t1x, t2x) - f.ex, fx = fmaxf(t1x, t2x) + f.ex;
    const float ny = fminf(t1y, t2y) - f.ey, fy = fmaxf(t1y, t2y) + f.ey;
    const float nz = fminf(t1z, t2z) - f.ez, fz = fmaxf(t1z, t2z) + f.ez;
    const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, t_min));
    const float tf = fminf(fminf(fx, fy), fminf(fz, t_max));
    t_near = tn;
    return tn <= tf;
}
// Cuboid face culling (flat top level). `f` = the OBJECT-space ray reduced like any other (make_rayf with the box's extent), lo / hi =
// the cuboid's object-space box. A face's exact f64 quad test (quad.rs:40-59) can only accept a hit with t in [t_min, t_best] at
// a point of the box's surface, and such a point satisfies, for every axis, t_near_axis <= t <= t_far_axis. With the proven
// slab bound |t' - t| <= e per axis (above): the face on plane b of axis a stays a candidate iff t'_b + e_a >= max over the
// axes of (near' - e) and t'_b - e_a <= min over the axes of (far' + e) — the ray's ENTRY and EXIT faces, plus whatever the margin
// cannot tell apart at an edge. Bits follow pt_cuboid's face order (cuboid.rs:18-52): 0 +z front, 1 +x right, 2 -z back,
// 3 -x left, 4 +y top, 5 -y bottom. Conservative only: which hit wins is still decided by the f64 tests of the faces kept.
PT_DEV uint32_t cuboid_face_mask(const float* lo, const float* hi, const RayF& f, float t_min, float t_max) {
    const float lx = __builtin_fmaf(lo[0], f.idx, -f.oix), hx = __builtin_fmaf(hi[0], f.idx, -f.oix);
    const float ly = __builtin_fmaf(lo[1], f.idy, -f.oiy), hy = __builtin_fmaf(hi[1], f.idy, -f.oiy);
    const float lz = __builtin_fmaf(lo[2], f.idz, -f.oiz), hz = __builtin_fmaf(hi[2], f.idz, -f.oiz);
    const float tn = fmaxf(fmaxf(fminf(lx, hx) - f.ex, fminf(ly, hy) - f.ey), fmaxf(fminf(lz, hz) - f.ez, t_min));
    const float tf = fminf(fminf(fmaxf(lx, hx) + f.ex, fmaxf(ly, hy) + f.ey), fminf(fmaxf(lz, hz) + f.ez, t_max));
    if (!(tn <= tf)) return 0u;
    auto cand = [&](float t, float e) -> uint32_t { return (t + e >= tn && t - e <= tf) ? 1u : 0u; };
    return cand(hz, f.ez) | (cand(hx, f.ex) << 1) | (cand(lz, f.ez) << 2) | (cand(lx, f.ex) << 3) | (cand(hy, f.ey) << 4) | (cand(ly, f.ey) << 5);
}
// one BVH2 node: both children tested, near-first order; returns the number of children to visit
PT_DEV int visit_node(const BvhNode* nd, const RayF& f, float t_min, float t_max, uint32_t& first, uint32_t& second) {
    const float4 q0 = ((const float4*)nd)[0], q1 = ((const float4*)nd)[1], q2 = ((const float4*)nd)[2];
    const uint4 q3 = ((const uint4*)nd)[3];
    const float lo0[3] = {q0.x, q0.y, q0.z}, hi0[3] = {q0.w, q1.x, q1.y};
    const float lo1[3] = {q1.z, q1.w, q2.x}, hi1[3] = {q2.y, q2.z, q2.w};
    float tn0, tn1;
    const bool h0 = slab_f32(lo0, hi0, f, t_min, t_max, tn0);
    const bool h1 = slab_f32(lo1, hi1, f, t_min, t_max, tn1);
    if (h0 && h1) {
        const bool swap = tn1 < tn0;
        first = swap ? q3.y : q3.x;
        second = swap ? q3.x : q3.y;
        return 2;
    }
    first = h0 ? q3.x : q3.y;
    return (h0 || h1) ? 1 : 0;
}
PT_DEV float t_max_f32(double t) { return __double2float_ru(t); }   // rounded UP: conservative upper end

// One triangle leaf (<= 8 triangles, BLAS order first .. first+count-1) for the ray of this lane.
// (Tried in round 2 and removed: a two-pass form — a division-free, exactly equivalent screen of every triangle, then
// the full test for the survivors only, so that a wave runs the expensive pass once or twice instead of `count` times.
// Bit-exact, but 1.5 % SLOWER on scene 6: the screen repeats two thirds of the test's arithmetic and the full pass still
// runs once for almost every leaf.)
PT_DEV void test_leaf(const SceneD& sc, uint32_t first, uint32_t count, const RayD& r, double t_min, uint32_t first_prim, Closest& best) {
    for (uint32_t i = first; i < first + count; ++i) {
        double t, u, v;
        if (hit_tri(sc.tris[i], r, t_min, t, u, v)) consider(best, t, first_prim + sc.tri_gid[i]);
    }
}

// U: `gid` is wave-uniform (the flat top-level walk) -> the primitive's record arrives by scalar loads (ldu)
template <bool U = false>
PT_DEV void test_world_prim(const SceneD& sc, const RayD& r, double t_min, uint32_t gid, Closest& best) {
    PrimRef pr;
    if constexpr (U) pr = ldu(&sc.prims[gid]); else pr = sc.prims[gid];
    if ((pr.kind & 0xFFu) == PRIM_SPHERE) {
        double t;
        V3 c;
        bool h;
        if constexpr (U) { const SphereD sp = ldu(&sc.spheres[pr.index]); h = hit_sphere(sp, r, t_min, t, c); }
        else h = hit_sphere(sc.spheres[pr.index], r, t_min, t, c);
        if (h) consider(best, t, gid);
    } else {
        double t, a, b;
        bool h;
        if constexpr (U) { const QuadD qd = ldu(&sc.quads[pr.index]); h = hit_quad(qd, r, t_min, t, a, b); }
        else h = hit_quad(sc.quads[pr.index], r, t_min, t, a, b);
        if (h) consider(best, t, gid);
    }
}

PT_DEV Closest closest_hit(const SceneD& sc, const RayD& wray, double t_min, uint32_t* stk /* &stack[0][lane] */) {
    Closest best{D_INF, HIT_NONE};
    RayD r = wray;
    const RayF fw = make_rayf(wray.o, wray.d, sc.tlas_extent);   // world-space reduction, kept across instances
    RayF f = fw;
    const float t_min_f = __double2float_rd(t_min);
    float t_max_f = t_max_f32(best.t);
    int sp = 0;
    uint32_t cur = sc.tlas_root;
    uint32_t mesh_first_prim = 0;   // Entry::first_prim of the mesh instance being walked
    for (;;) {
        if ((cur & REF_TYPE_MASK) == REF_NODE) {
            const BvhNode* nd = &sc.nodes[cur];
            uint32_t c0, c1;
            const int n = visit_node(nd, f, t_min_f, t_max_f, c0, c1);
            if (n == 2 && sp < TRAVERSAL_STACK) stk[(sp++) * BLOCK] = c1;
            if (n > 0) {
                cur = c0;
                continue;
            }
        } else if ((cur & REF_TYPE_MASK) == REF_TRIS) {
            const uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 7u) + 1u;
            test_leaf(sc, first, count, r, t_min, mesh_first_prim, best);
            t_max_f = t_max_f32(best.t);
        } else if ((cur & REF_TYPE_MASK) == REF_ENTRY) {
            const Entry e = sc.entries[cur & 0x3FFFFFFFu];
            const RayD lr = ray_to_local_chain(sc, e.inst, wray);
            if (e.kind == ENTRY_MESH) {
                r = lr;
                mesh_first_prim = e.first_prim;
                f = make_rayf(r.o, r.d, e.extent);
                if (sp < TRAVERSAL_STACK) stk[(sp++) * BLOCK] = REF_LEAVE_INSTANCE;
                cur = e.blas_root;
                continue;
            }
            const uint32_t n = e.kind == ENTRY_CUBOID ? 6u : 1u;   // cuboid.rs: six quads, linear
            for (uint32_t i = 0; i < n; ++i) test_world_prim(sc, lr, t_min, e.first_prim + i, best);
            t_max_f = t_max_f32(best.t);
        } else if (cur == REF_LEAVE_INSTANCE) {
            r = wray;
            f = fw;
        }
        if (sp == 0) break;
        cur = stk[(--sp) * BLOCK];
    }
    return best;
}

// floor(a / b) and a - b * floor(a / b) for a < 2^53, 0 < b < 2^31, by ONE f64 division and an exact integer correction.
// (The compiler's inline expansion of a 64-bit unsigned division is ~100 instructions, a third of them quarter-rate
// integer multiplies, and it ran once per regenerated camera ray.)
PT_DEV void divmod_u53(unsigned long long a, uint32_t b, unsigned long long& q, uint32_t& r) {
    unsigned long long q0 = (unsigned long long)((double)a / (double)b);     // within 1 of the true quotient
    long long rem = (long long)(a - q0 * (unsigned long long)b);
    if (rem < 0) { --q0; rem += (long long)b; }
    else if (rem >= (long long)b) { ++q0; rem -= (long long)b; }
    q = q0;
    r = (uint32_t)rem;
}
PT_DEV void divmod_u31(uint32_t a, uint32_t b, uint32_t& q, uint32_t& r) {   // a, b < 2^31: the f64 quotient floors exactly
    q = (uint32_t)((double)a / (double)b);
    r = a - q * b;
}
// dynamic mode: work item -> (pixel, sample) and the pixel's row / column; false when the item lies outside a ragged image edge
PT_DEV bool work_to_pixel(const PoolD& pool, unsigned long long w, uint32_t& pixel, uint32_t& sample, uint32_t& row, uint32_t& col) {
    unsigned long long q;
    uint32_t in_frame;
    divmod_u53(w, pool.n_tile_pixels, q, in_frame);
    sample = pool.spp_begin + (uint32_t)q;
    const uint32_t tile = in_frame >> 6, in_tile = in_frame & 63u;
    uint32_t ty, tx;
    divmod_u31(tile, pool.tiles_x, ty, tx);
    const uint32_t x = tx * 8u + (in_tile & 7u), y = ty * 8u + (in_tile >> 3);
    pixel = y * pool.width + x;
    row = y;
    col = x;
    return x < pool.width && y < pool.height;
}
// shard-local counter value -> global work item: 64-item chunks are dealt round-robin to the shards
PT_DEV unsigned long long shard_item(unsigned long long c, uint32_t shard) {
    return (c >> 6) * (unsigned long long)(WORK_SHARDS * 64u) + (unsigned long long)shard * 64ull + (c & 63ull);
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_init(CamD cam, PoolD pool, uint64_t seed) {
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < pool.n_alloc; s += gridDim.x * BLOCK) {
        uint32_t pixel, sample, row = 0, col = 0;
        bool has_work;
        bool idle = false;
        if (pool.dynamic) {   // initial work items 0 .. n_slots-1 (the host starts the shard counters there)
            has_work = s < pool.n_slots && (unsigned long long)s < pool.total_work;
            idle = has_work && !work_to_pixel(pool, s, pixel, sample, row, col);
            if (!has_work || idle) { pixel = 0; sample = 0; }
        } else {
            pixel = s % pool.n_pixels;
            sample = pool.spp_begin + s / pool.n_pixels;
            divmod_u31(pixel, cam.width, row, col);
            has_work = s < pool.n_slots && sample < pool.spp_end;
            pool.ax[s] = 0.0; pool.ay[s] = 0.0; pool.az[s] = 0.0;
            pool.rx[s] = 0.0; pool.ry[s] = 0.0; pool.rz[s] = 0.0;
        }
        pool.hit_prim[s] = (CLASS_DEAD << HIT_CLASS_SHIFT) | HIT_ID_MASK;   // overwritten by the first K2 launch
        if (!pool.compact) store_path(pool, s, V3{1.0, 1.0, 1.0}, pixel, 0u);
        if (!has_work || idle) {
            pool.bounce[s] = idle ? SLOT_IDLE : SLOT_DEAD;
            store_ray(pool, s, RayD{}, sample, 0u, pixel, 0u);
            continue;
        }
        Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), pixel, sample, 0u};
        RayD r = generate_ray(cam, row, col, rng);
        store_ray(pool, s, r, sample, rng.draw, pixel, 0u);
        pool.bounce[s] = 0;
    }
}

template <int STRIDE = BLOCK>
PT_DEV void blas_pass(const SceneD& sc, const RayD& wray, const Entry& e, double t_min, float t_min_f, uint32_t* stk, int cap, Closest& best);

// ---- the FLAT top level (SceneD::tlas_flat: at most TLAS_FLAT_MAX world entries), walked by a whole wave ---------------------
// The wave loops over the entry list together — the entry index is wave-uniform, so boxes and entries arrive by scalar loads
// and there is no top-level stack. Round 2 measured what that loop cost when every entry whose box ANY lane entered was
// tested on the spot by the whole wave: 46 % of k_extend2's time on scene 6 (ten entries: each 64-ray chunk ran five sphere
// tests, one quad test and a cuboid's six, with a handful of lanes active in each). Now the box pass only RECORDS
// (ray, primitive) pairs — ray = lane of the chunk, primitive = global id, one pair per cuboid face — in a small per-wave ring
// in LDS, and whenever 64 pairs are waiting the wave tests them in ONE dense pass: lane i takes pair i, fetches that ray from
// its owner lane (ds_bpermute), transforms it into the primitive's frame and runs the primitive's exact f64 test. Results
// meet in LDS: minimum t per ray (64-bit LDS atomic min on the bits of the positive double), ties -> larger id (atomic max) —
// the same order-independent rule as consider(), so the hit is bit-identical. Mesh entries come second, their boxes trimmed
// by the non-mesh result.
#ifndef PT_PAIR_DENSE_MIN
#define PT_PAIR_DENSE_MIN 16        // lanes of a chunk in one entry's box from which the entry is tested on the spot (break-even of the two forms)
#endif
#ifndef PT_PAIR_PASS_ATTR
#define PT_PAIR_PASS_ATTR PT_DEV
#endif
#ifndef PT_FLAT_DIRECT
#define PT_FLAT_DIRECT 1            // 0: on-the-spot tests look the primitive up in prims[] (A/B)
#endif
#ifndef PT_CUBOID_CULL
#define PT_CUBOID_CULL 1            // 0: all six faces of a cuboid are tested (A/B)
#endif
#ifndef PT_PAIR_SINGLE
#define PT_PAIR_SINGLE 1            // 0: only cuboids (six faces behind one transform) go through the pair passes
#endif
constexpr int PAIR_CAP = 128;       // ring of waiting pairs per wave (a pass runs as soon as 64 wait, an append adds <= 64)
// pair word = id << 6 | lane: the host (pt_scene.cpp) only sets tlas_flat when the ids of spheres / quads / cuboid faces are below 2^26
struct PairLds {                    // per wave
    unsigned long long* bt;         // [64] bits of the closest t so far of every ray of the chunk (+inf: none)
    uint32_t* bid;                  // [64] its primitive id
    uint32_t* pairs;                // [PAIR_CAP]
};
// LDS operations of one wave execute in issue order; this keeps the compiler from moving them across the steps of the protocol
PT_DEV void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
PT_PAIR_PASS_ATTR void pair_pass(const SceneD& sc, const RayD& r, double t_min, int lane, const PairLds& L, uint32_t head, uint32_t n) {
    wave_lds_order();
    const bool mine = (uint32_t)lane < n;
    const uint32_t code = ((volatile uint32_t*)L.pairs)[(head + (uint32_t)lane) % PAIR_CAP];
    const int src = mine ? (int)(code & 63u) : lane;
    const uint32_t gid = code >> 6;
    const RayD pr_ray{V3{__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src)}, V3{__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src)},
                      __shfl(r.time, src)};
    bool hit = false;
    double t = 0.0;
    if (mine) {
        const PrimRef pr = sc.prims[gid];
        const RayD lr = ray_to_local_chain(sc, pr.inst, pr_ray);
        if ((pr.kind & 0xFFu) == PRIM_SPHERE) {
            V3 c;
            hit = hit_sphere(sc.spheres[pr.index], lr, t_min, t, c);
        } else {
            double a, b;
            hit = hit_quad(sc.quads[pr.index], lr, t_min, t, a, b);
        }
    }
    // t > t_min > 0: the bits of t order like t. Every winner of this pass holds the pass's minimum, so all of them see the same
    // `before` and agree on whether the ray's closest t went down (older ids are void) or stayed (ids compete).
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    volatile unsigned long long* bt = L.bt;
    volatile uint32_t* bid = L.bid;
    unsigned long long before = 0ull;
    if (hit) before = bt[src];
    hit = hit && tb <= before;
    wave_lds_order();
    if (hit) atomicMin(&L.bt[src], tb);
    wave_lds_order();
    const bool win = hit && bt[src] == tb;
    if (win && tb < before) bid[src] = 0u;
    wave_lds_order();
    if (win) atomicMax(&L.bid[src], gid);
    wave_lds_order();
}
// Must be called by whole waves (`alive` = false for lanes without a ray). on_mesh(ei, entry, best): a lane's ray entered the
// box of mesh entry `ei` (wave-uniform index).
// PAIRS: cuboids few rays of the chunk enter go through the pair passes (L must be valid); false: everything on the spot.
// CULL: compile the cuboid face culling in (the instantiation for scenes without cuboids leaves it out: its registers spilled there).
template <bool PAIRS, bool CULL, class OnMesh>
PT_DEV Closest flat_top_level(const SceneD& sc, bool alive, const RayD& r, const RayF& f, double t_min, float t_min_f, int lane, const PairLds& L,
                              OnMesh&& on_mesh) {
    if constexpr (PAIRS) {
        ((volatile unsigned long long*)L.bt)[lane] = (unsigned long long)__double_as_longlong(D_INF);
        ((volatile uint32_t*)L.bid)[lane] = HIT_NONE;
    }
    uint32_t head = 0, tail = 0;                                     // wave-uniform
    Closest best{D_INF, HIT_NONE};                                   // hits of the entries tested on the spot
    float t_max_f = t_max_f32(best.t);
    bool pairs_open = PAIRS;                                         // wave-uniform: pair results not yet merged into `best`
    auto close_pairs = [&]() {
        if (tail != head) pair_pass(sc, r, t_min, lane, L, head, tail - head);
        if (tail != 0u) {                                            // some pass ran: its results join the on-the-spot ones (same rule)
            wave_lds_order();
            const double lt = __longlong_as_double((long long)((volatile unsigned long long*)L.bt)[lane]);
            if (lt < D_INF) consider(best, lt, ((volatile uint32_t*)L.bid)[lane]);
            t_max_f = t_max_f32(best.t);
        }
        pairs_open = false;
    };
    for (uint32_t k = 0; k < sc.n_entries; ++k) {                    // SceneD::entry_box: non-mesh entries first
        const EntryBox bx = ldu(&sc.entry_box[k]);
        if (PAIRS && pairs_open && bx.kind == ENTRY_MESH) close_pairs();
        float tn;
        const bool hb = alive && slab_f32(bx.lo, bx.hi, f, t_min_f, t_max_f, tn);
        const unsigned long long m = __ballot(hb);
        if (m == 0ull) continue;
        if (bx.kind == ENTRY_MESH) {
            if (hb) {
                const Entry e{bx.kind, bx.first_prim, bx.inst, bx.blas_root, bx.extent, bx.n_prims, {0u, 0u}};
                on_mesh(bx.entry, e, best);
                t_max_f = t_max_f32(best.t);
            }
            continue;
        }
        const uint32_t n_faces = bx.kind == ENTRY_CUBOID ? 6u : 1u;  // cuboid.rs: six quads, linear
        const uint32_t cnt = (uint32_t)__popcll(m);
        if (!PAIRS || (!PT_PAIR_SINGLE && n_faces == 1u) || cnt >= (uint32_t)PT_PAIR_DENSE_MIN) {
            // a box many rays of the chunk enter: the test on the spot, with the primitive's record in scalar registers, is
            // cheaper than that many pairs — and its hits trim the boxes that follow
            if (CULL && PT_CUBOID_CULL && bx.kind == ENTRY_CUBOID && bx.prim_kind == PRIM_QUAD) {
                // the six faces behind one transform: only those the object-space ray can enter or leave through are tested, and
                // a face no lane of the chunk needs costs neither its 128-byte record nor its test (cuboid_face_mask)
                RayD lr{};
                uint32_t fm = 0u;
                if (hb) {
                    lr = ray_to_local_chain<true>(sc, bx.inst, r);
                    const CuboidBox cb = ldu(&sc.cuboid_box[k]);
                    fm = cuboid_face_mask(cb.lo, cb.hi, make_rayf(lr.o, lr.d, bx.extent), t_min_f, t_max_f);
                }
                for (uint32_t fi = 0; fi < 6u; ++fi) {
                    const bool need = (fm >> fi) & 1u;
                    if (__ballot(need) == 0ull) continue;
                    const QuadD qd = ldu(&sc.quads[bx.prim_index + fi]);
                    double t, a, b;
                    if (need && hit_quad(qd, lr, t_min, t, a, b)) consider(best, t, bx.first_prim + fi);
                }
                t_max_f = t_max_f32(best.t);
                continue;
            }
            if (hb) {
                const RayD lr = ray_to_local_chain<true>(sc, bx.inst, r);
                if (PT_FLAT_DIRECT && bx.prim_kind == PRIM_SPHERE) {
                    const SphereD sp = ldu(&sc.spheres[bx.prim_index]);
                    double t;
                    V3 c;
                    if (hit_sphere(sp, lr, t_min, t, c)) consider(best, t, bx.first_prim);
                } else if (PT_FLAT_DIRECT && bx.prim_kind == PRIM_QUAD) {
                    for (uint32_t fi = 0; fi < n_faces; ++fi) {
                        const QuadD qd = ldu(&sc.quads[bx.prim_index + fi]);
                        double t, a, b;
                        if (hit_quad(qd, lr, t_min, t, a, b)) consider(best, t, bx.first_prim + fi);
                    }
                } else {
                    for (uint32_t fi = 0; fi < n_faces; ++fi) test_world_prim<true>(sc, lr, t_min, bx.first_prim + fi, best);
                }
                t_max_f = t_max_f32(best.t);
            }
            continue;
        }
        if constexpr (PAIRS) {
            uint32_t fm = 0x3Fu;                                         // faces this lane's ray may hit (all, when not culled)
            if (CULL && PT_CUBOID_CULL && bx.kind == ENTRY_CUBOID && bx.prim_kind == PRIM_QUAD) {
                fm = 0u;
                if (hb) {
                    const RayD lr = ray_to_local_chain<true>(sc, bx.inst, r);
                    const CuboidBox cb = ldu(&sc.cuboid_box[k]);
                    fm = cuboid_face_mask(cb.lo, cb.hi, make_rayf(lr.o, lr.d, bx.extent), t_min_f, t_max_f);
                }
            }
            for (uint32_t fi = 0; fi < n_faces; ++fi) {
                const bool need = hb && ((fm >> fi) & 1u);
                const unsigned long long mf = __ballot(need);
                if (mf == 0ull) continue;
                const uint32_t rank = (uint32_t)__popcll(mf & ((1ull << lane) - 1ull));
                if (need) ((volatile uint32_t*)L.pairs)[(tail + rank) % PAIR_CAP] = ((bx.first_prim + fi) << 6) | (uint32_t)lane;
                tail += (uint32_t)__popcll(mf);
                if (tail - head >= 64u) {
                    pair_pass(sc, r, t_min, lane, L, head, 64u);
                    head += 64u;
                }
            }
        }
    }
    if (PAIRS && pairs_open) close_pairs();
    return best;
}
template <bool PAIRS>
PT_DEV Closest closest_hit_flat(const SceneD& sc, bool alive, const RayD& r, double t_min, uint32_t* stk, int lane, const PairLds& L) {
    const float t_min_f = __double2float_rd(t_min);
    const RayF f = make_rayf(r.o, r.d, sc.tlas_extent);
    return flat_top_level<PAIRS, PAIRS>(sc, alive, r, f, t_min, t_min_f, lane, L,   // (the batch kernel's instantiation without pair passes serves scenes without cuboids)
                                [&](uint32_t, const Entry& e, Closest& best) { blas_pass(sc, r, e, t_min, t_min_f, stk, TRAVERSAL_STACK, best); });
}

// K2, batch form: a fixed grid walks the pool with a grid-stride loop; each lane traverses one ray
// at a time, a wave moves on when its slowest lane is done. Lowest overhead; SIMD utilisation
// suffers when traversal lengths inside a wave differ a lot (sky ray next to a mesh ray). Used for
// scenes without meshes (and as the fallback for BVHs deeper than k_extend2's LDS stack).
// FLAT: SceneD::tlas_flat. PAIRS (FLAT only): SceneD::flat_pairs — the scene has cuboids, whose six faces behind one transform
// are what the pair passes of flat_top_level pay for (scene 3: K2 -19 %, scene 7: -11 %); that instantiation runs three blocks
// per CU (its extra state spills at 128 registers and costs more than the fourth block brings), the others four.
template <bool FLAT, bool PAIRS>
__global__ __launch_bounds__(BLOCK, PAIRS ? 3 : PT_EXTEND_BATCH_BLOCKS) void k_extend(SceneD sc, PoolD pool, CountersD* cnt) {
    __shared__ uint32_t stack[TRAVERSAL_STACK * BLOCK];
    __shared__ unsigned long long s_pair_t[PAIRS ? BLOCK : 1];
    __shared__ uint32_t s_pair_id[PAIRS ? BLOCK : 1], s_pairs[PAIRS ? (BLOCK / 64) * PAIR_CAP : 1];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    const PairLds pl{&s_pair_t[PAIRS ? wave * 64 : 0], &s_pair_id[PAIRS ? wave * 64 : 0], &s_pairs[PAIRS ? wave * PAIR_CAP : 0]};
    unsigned long long nseg = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->win_shade = 0;
    if (ldu(&cnt->alive) == 0ull) return;   // the frame is done: the launches the host had already queued behind its last poll cost a few microseconds each
    // n_alloc is a multiple of BLOCK: whole waves run every chunk (closest_hit_flat ballots)
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < pool.n_alloc; s += gridDim.x * BLOCK) {
        const uint32_t state = pool.bounce[s];
        const bool alive = state < SLOT_IDLE;
        RayD r{};
        if (alive) r = load_ray(pool, s);
        Closest c{D_INF, HIT_NONE};
        if (FLAT) c = closest_hit_flat<PAIRS>(sc, alive, r, 1e-3, &stack[threadIdx.x], lane, pl);   // camera.rs:171,179
        else if (alive) c = closest_hit(sc, r, 1e-3, &stack[threadIdx.x]);
        stnt(&pool.hit_prim[s], hit_word(sc, alive ? c.id : dead_or_idle(state)));
        if (alive) ++nseg;
    }
    nseg = wave_sum(nseg);
    if (nseg && (threadIdx.x & 63u) == 0u) atomicAdd(&cnt->segments, nseg);
}

// ---------------------------------------------------------------------------------------
// K2, two-phase form (default). The batch kernel above leaves most lanes idle: a sky ray is done
// after ~5 steps while a lane next to it walks a mesh for 50-150 (17 % VALU lane utilisation in the
// first profile). Here a block takes a WINDOW of 2048 slots and
//   phase A: every ray walks only the TOP-LEVEL tree; spheres / quads / cuboids are intersected on
//            the spot, mesh instances whose box it enters are only RECORDED (<= 4 per ray, in LDS);
//   phase B: the rays that recorded something are compacted into an LDS list and the block's waves
//            draw from it (work stealing): each lane walks the recorded meshes of its ray one after
//            the other, starting from the phase-A best hit, and takes the next ray of the list when
//            it is done (lanes are refilled, a wave does not wait for the longest ray of a group).
// Rays that never touch a mesh finish in the short, uniform phase A; the long mesh traversals run in
// dense waves whose lanes all do the same kind of work. The closest hit is order-independent
// (minimum t, ties -> larger id), so the result is bit-identical to the batch kernel's.
// ---------------------------------------------------------------------------------------
// The phase-A best hit of a ray waits in LDS (t and id), and the window's final primitive ids leave
// with ONE coalesced store per slot: k_shade re-intersects the primitive (reconstruct_hit) and never
// needs t, so 4 B per slot is all this kernel writes.
// LDS per block: STACK x 1 KB (traversal stacks) + 18.5 KB, STACK in {16, 20, 24}: the host picks the
// smallest that covers the scene (pt_scene::stack_need_extend2 — only the deepest mesh tree when the top
// level is walked flat); deeper scenes use the batch kernel. The kernel runs three blocks per CU: a
// fourth would cap it at 128 registers and the spills cost more than the extra waves bring (measured).
// Tried and dropped: the two phases as two kernels with a global candidate list; every
// wave on its own 256-slot window without block barriers; warming the next window's ray lines.
#ifdef PT_STAMPS
// Diagnostic build: where a k_shade wave spends its cycles (s_memtime ticks; MI355X_MICROARCH.md "In-kernel stamps"). The stamp
// after the record loads forces vmcnt(0) so that the first segment is the pure fetch wait. Never part of the product build.
PT_DEV unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__shared__ unsigned long long g_prof[N_CLASSES + 1][PROF_COLS];
#define PT_STAMP(i) const unsigned long long t_##i = stamp()
#define PT_STAMP_VAR(i) unsigned long long t_##i = 0
#define PT_STAMP_SET(i) t_##i = stamp()
#define PT_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define PT_STAMP(i)
#define PT_STAMP_VAR(i)
#define PT_STAMP_SET(i)
#define PT_DRAIN()
#endif

#ifndef PT_TAIL_SPLIT
#define PT_TAIL_SPLIT 1             // 0: whole windows to the end of k_extend2's queue (A/B)
#endif
#ifndef PT_TAIL_PARTS
#define PT_TAIL_PARTS 2             // rounds a tail window is handed out in (2: halves, 4: quarters)
#endif
#ifndef PT_TAIL_SPAN
#define PT_TAIL_SPAN 1              // tail = the last PT_TAIL_SPAN windows per block launched
#endif
#ifndef PT_K2_REVERSE
#define PT_K2_REVERSE 1
#endif
#ifndef PT_K2_PREFETCH
#define PT_K2_PREFETCH 0              // 1: phase A holds the next chunk's ray in registers while it walks the current one — the round-1 form,
#endif                               // which at 128 registers costs 60 B of spills per lane: without it K2 runs 7 % faster and writes 0.46 GB less

constexpr int EXT_WINDOW = 2048;   // slots per block window
#ifndef PT_REFILL_MIN
#define PT_REFILL_MIN 16
#endif
constexpr uint32_t REFILL_MIN = PT_REFILL_MIN;   // idle lanes that trigger a refill of the wave in phase B
#ifndef PT_EXT_CAND
#define PT_EXT_CAND 768
#endif
#ifndef PT_EXT_CAND_SMALL
#define PT_EXT_CAND_SMALL 1024
#endif
// Candidate list per 2048 slots (LDS; scaled with the block size); a fuller window walks the rest in phase A, one mesh after the other
// inside the divergent top-level code — expensive: scene 6, 128-thread blocks: 512 entries K2 +8.6 %, 768 (37.5 % of the window, the
// round-1 choice) the reference, 1024 -2.3 %. The 128-thread form has the LDS for 1024 (19.5 KB per block, eight blocks per CU); the
// 256-thread forms with their deeper stacks stay at 768.
constexpr int EXT_CAND = PT_EXT_CAND, EXT_CAND_SMALL = PT_EXT_CAND_SMALL;
static_assert(EXT_WINDOW % BLOCK == 0 && EXT_WINDOW <= 65536, "k_extend2: s_cand_sl holds 16-bit slot offsets inside the window");
static_assert(EXT_WINDOW == SORT_WINDOW_SLOTS, "the pool is allocated in whole windows of this size (pt_render.cpp rounds n_alloc to 2048)");

template <int STRIDE>   // STRIDE: threads per block = distance of a lane's consecutive stack entries in LDS
PT_DEV void blas_pass(const SceneD& sc, const RayD& wray, const Entry& e, double t_min, float t_min_f, uint32_t* stk, int cap, Closest& best) {
    const RayD r = ray_to_local_chain(sc, e.inst, wray);
    const RayF f = make_rayf(r.o, r.d, e.extent);
    float t_max_f = t_max_f32(best.t);
    int sp = 0;
    uint32_t cur = e.blas_root;
    // "while-while" traversal: every lane first descends until it HOLDS a triangle leaf (cheap f32 box
    // tests; lanes that already found theirs idle), then the wave runs the expensive f64 triangle
    // tests together. Interleaving the two per iteration made almost every iteration pay for a leaf.
    for (;;) {
        while ((cur & REF_TYPE_MASK) == REF_NODE) {
            uint32_t c0, c1;
            const int n = visit_node(&sc.nodes[cur], f, t_min_f, t_max_f, c0, c1);
            if (n == 2 && sp < cap) stk[(sp++) * STRIDE] = c1;
            if (n > 0) cur = c0;
            else if (sp > 0) cur = stk[(--sp) * STRIDE];
            else cur = REF_EMPTY;
        }
        if ((cur & REF_TYPE_MASK) != REF_TRIS) break;             // REF_EMPTY: nothing left
        const uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 7u) + 1u;
        test_leaf(sc, first, count, r, t_min, e.first_prim, best);
        t_max_f = t_max_f32(best.t);
        if (sp == 0) break;
        cur = stk[(--sp) * STRIDE];
    }
}

// KB: threads per block (256; [r3] other sizes for A/B — the window and the candidate list scale with it; MINB = waves per SIMD, which is what hipcc's launch bound means)
template <int EXT_STACK, int MINB, int KB = BLOCK>
__global__ __launch_bounds__(KB, MINB) void k_extend2(SceneD sc, PoolD pool, CountersD* cnt) {
    constexpr int WIN = EXT_WINDOW / BLOCK * KB, CAND = (KB <= 128 ? EXT_CAND_SMALL : EXT_CAND) / BLOCK * KB;
    __shared__ uint32_t stack[EXT_STACK * KB];
    __shared__ uint32_t s_best_id[WIN];                            //  8 KB  closest primitive of every slot of the window
    __shared__ double s_cand_t[CAND];                              //  6 KB  candidates (rays that entered mesh boxes): phase-A best t,
    __shared__ uint32_t s_cand_items[CAND];                        //  3 KB  recorded mesh entries, 8 bit each, 0xFF = none,
    __shared__ uint16_t s_cand_sl[CAND];                           //        slot inside the window
    __shared__ uint32_t s_nrays, s_next, s_win;
    uint32_t* stk = &stack[threadIdx.x];
    const int lane = (int)(threadIdx.x & 63u);
    const double t_min = 1e-3;                                     // camera.rs:171,179
    const float t_min_f = __double2float_rd(t_min);
    unsigned long long nseg = 0;
    const uint32_t n_windows = pool.n_alloc / WIN;
    // [r3] The queue's END is handed out in HALF windows (PT_TAIL_SPLIT): with ~16 windows per block and launch a block idles half a
    // window on average while the last ones finish; the last gridDim.x windows go out as two rounds of four chunks each, so the spread
    // at the launch's end is half as long: K2 -0.8 % on the 33.6 M-slot pool, -2.1 % on a 16.8 M-slot one (quarters: +2 %; the last TWO
    // windows per block in halves: +1.4 %; both: +5 % — PT_TAIL_PARTS / PT_TAIL_SPAN)
    constexpr uint32_t PARTS = PT_TAIL_SPLIT ? PT_TAIL_PARTS : 1, CH = (uint32_t)(WIN / KB) / PARTS;   // rounds per tail window, chunks per round
    static_assert(PARTS * CH == (uint32_t)(WIN / KB), "");
    const uint32_t n_tail = PT_TAIL_SPLIT ? (n_windows < gridDim.x * PT_TAIL_SPAN ? n_windows : gridDim.x * PT_TAIL_SPAN) : 0u;
    const uint32_t n_full = n_windows - n_tail, n_queue = n_full + PARTS * n_tail;
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->win_shade = 0;
    if (ldu(&cnt->alive) == 0ull) return;   // (see k_extend)
#ifdef PT_STAMPS
    if (threadIdx.x < 8) g_prof[CLASS_DEAD][threadIdx.x] = 0ull;
#endif
    // (the next window's index is drawn before the barrier that ends a window and published by it: see k_shade)
    if (threadIdx.x == 0) { s_win = (uint32_t)atomicAdd(&cnt->win_extend, 1ull); s_nrays = 0; s_next = 0; }
    __syncthreads();
    for (;;) {
        const uint32_t q = s_win;
        if (q >= n_queue) break;
        const uint32_t win = q < n_full ? q : n_full + (q - n_full) / PARTS;
        const int j_lo = q < n_full ? 0 : (int)(((q - n_full) % PARTS) * CH), j_hi = q < n_full ? WIN / KB : j_lo + (int)CH;   // this round's chunks
        // [r3] K2 walks the pool from its END, k_shade from its beginning: each kernel starts on the windows the other touched last,
        // i.e. on what the 256 MB memory-side cache still holds of the 3-5 GB the previous launch streamed (PT_K2_REVERSE=0: A/B)
        const uint32_t wbase = (PT_K2_REVERSE ? n_windows - 1u - win : win) * WIN;
        PT_STAMP(e0);
        // ---- phase A: top level only ---------------------------------------------------------------
        // (PT_K2_PREFETCH: the ray of the NEXT chunk requested before this chunk's traversal starts)
#if PT_K2_PREFETCH
        uint32_t state_next = pool.bounce[wbase + (uint32_t)j_lo * KB + threadIdx.x];
        RayD r_next{};
        if (state_next < SLOT_IDLE) r_next = load_ray(pool, wbase + (uint32_t)j_lo * KB + threadIdx.x);
#endif
        for (int j = j_lo; j < j_hi; ++j) {
            const uint32_t sl = (uint32_t)j * KB + threadIdx.x;
#if PT_K2_PREFETCH
            const uint32_t state = state_next;
            const bool alive = state < SLOT_IDLE;
            const RayD r = r_next;
            if (j + 1 < j_hi) {
                state_next = pool.bounce[wbase + sl + KB];
                if (state_next < SLOT_IDLE) r_next = load_ray(pool, wbase + sl + KB);
            }
#else
            const uint32_t state = pool.bounce[wbase + sl];
            const bool alive = state < SLOT_IDLE;
            RayD r{};
            if (alive) r = load_ray(pool, wbase + sl);
#endif
            uint32_t n_my = 0, items = 0xFFFFFFFFu;
            RayF f{};
            Closest best{D_INF, HIT_NONE};
            float t_max_f = t_max_f32(best.t);
            if (alive) {
                ++nseg;
                f = make_rayf(r.o, r.d, sc.tlas_extent);
            }
            // one world entry whose box the ray enters: meshes are recorded, everything else is tested on the spot
            auto visit_entry = [&](auto uniform, uint32_t ei, const Entry& e, int sp) {   // uniform: ei is the same in every lane
                constexpr bool U = decltype(uniform)::value;
                if (e.kind == ENTRY_MESH) {
                    if (n_my < 4u && ei < 0xFFu) {
                        items = (items & ~(0xFFu << (8u * n_my))) | (ei << (8u * n_my));   // defer to phase B
                        ++n_my;
                    } else {
                        blas_pass<KB>(sc, r, e, t_min, t_min_f, stk + (size_t)sp * KB, EXT_STACK - sp, best);   // a fifth mesh / a wide index: walk it now
                        t_max_f = t_max_f32(best.t);
                    }
                } else {
                    const RayD lr = ray_to_local_chain<U>(sc, e.inst, r);
                    const uint32_t n = e.kind == ENTRY_CUBOID ? 6u : 1u;       // cuboid.rs: six quads, linear
                    for (uint32_t i = 0; i < n; ++i) test_world_prim<U>(sc, lr, t_min, e.first_prim + i, best);
                    t_max_f = t_max_f32(best.t);
                }
            };
            if (sc.tlas_flat) {
                // Small top level: the wave walks the ENTRY LIST together instead of each lane walking the tree
                // (flat_top_level; the pair passes are left to the batch kernel: measured slower here, scene 6).
                best = flat_top_level<false, true>(sc, alive, r, f, t_min, t_min_f, lane, PairLds{}, [&](uint32_t ei, const Entry& e, Closest& b) {
                    best = b;                                       // visit_entry works on this frame's `best`
                    visit_entry(std::true_type{}, ei, e, 0);
                    b = best;
                });
            } else if (alive) {
                int sp = 0;
                uint32_t cur = sc.tlas_root;
                for (;;) {
                    if ((cur & REF_TYPE_MASK) == REF_NODE) {
                        uint32_t c0, c1;
                        const int n = visit_node(&sc.nodes[cur], f, t_min_f, t_max_f, c0, c1);
                        if (n == 2 && sp < EXT_STACK) stk[(sp++) * KB] = c1;
                        if (n > 0) {
                            cur = c0;
                            continue;
                        }
                    } else if ((cur & REF_TYPE_MASK) == REF_ENTRY) {
                        const uint32_t ei = cur & 0x3FFFFFFFu;
                        visit_entry(std::false_type{}, ei, sc.entries[ei], sp);
                    }
                    if (sp == 0) break;
                    cur = stk[(--sp) * KB];
                }
            }
            // append the rays that recorded meshes to the window's candidate list (one LDS atomic per wave, slot
            // order kept inside the wave); when the list is full — a window that is nearly all mesh — walk them now
            const unsigned long long m = __ballot(n_my > 0);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&s_nrays, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, leader);
                if (n_my > 0) {
                    const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (pos < (uint32_t)CAND) {
                        s_cand_t[pos] = best.t;
                        s_cand_items[pos] = items;
                        s_cand_sl[pos] = (uint16_t)sl;
                    } else {
                        for (uint32_t k = 0; k < n_my; ++k) blas_pass<KB>(sc, r, sc.entries[(items >> (8u * k)) & 0xFFu], t_min, t_min_f, stk, EXT_STACK, best);
                    }
                }
            }
            s_best_id[sl] = alive ? best.id : dead_or_idle(state);
        }
        PT_STAMP(e1);
        __syncthreads();
        PT_STAMP(e2);
        // ---- phase B: mesh traversals, 64 rays per pull ------------------------------------------------
        const uint32_t n_rays = s_nrays < (uint32_t)CAND ? s_nrays : (uint32_t)CAND;
        {
            // Lanes are refilled: a lane whose ray is done does not wait for the longest ray of a fixed group of 64 —
            // when at least REFILL_MIN lanes of the wave are idle they draw the next candidates from the list (one
            // LDS atomic per wave) and the wave goes on with every lane at its own ray. s_next counts RAYS here.
            bool busy = false, exhausted = false;                   // exhausted: wave-uniform, the list has run out
            uint32_t sl = 0, items = 0, item_k = 0, cur = REF_EMPTY, first_prim = 0;
            int sp = 0;
            RayD wr{}, r{};
            RayF f{};
            Closest best{D_INF, HIT_NONE};
            float t_max_f = 0.0f;
            auto start_item = [&]() -> bool {                       // enters mesh number item_k of this lane's ray, if any
                const uint32_t ei = item_k < 4u ? (items >> (8u * item_k)) & 0xFFu : 0xFFu;
                if (ei == 0xFFu) return false;
                const Entry e = sc.entries[ei];
                r = ray_to_local_chain(sc, e.inst, wr);
                first_prim = e.first_prim;
                f = make_rayf(r.o, r.d, e.extent);
                t_max_f = t_max_f32(best.t);
                cur = e.blas_root;
                sp = 0;
                return true;
            };
            for (;;) {
                const unsigned long long idle = __ballot(!busy);
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                if (!exhausted && (n_idle >= REFILL_MIN || n_idle == 64u)) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&s_next, n_idle);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    exhausted = base + n_idle >= n_rays;
                    if (!busy) {
                        const uint32_t idx = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                        if (idx < n_rays) {
                            sl = s_cand_sl[idx];
                            items = s_cand_items[idx];
                            wr = load_ray(pool, wbase + sl);
                            best = Closest{s_cand_t[idx], s_best_id[sl]};
                            item_k = 0;
                            busy = start_item();                    // a candidate always has at least one item
                        }
                    }
                }
                if (__ballot(busy) == 0ull) {
                    if (exhausted) break;
                    continue;                                       // everybody idle: the refill above was forced, go again
                }
                // descend until every busy lane holds a triangle leaf or has run out of nodes in this mesh
                while (busy && (cur & REF_TYPE_MASK) == REF_NODE) {
                    uint32_t c0, c1;
                    const int n = visit_node(&sc.nodes[cur], f, t_min_f, t_max_f, c0, c1);
                    if (n == 2 && sp < EXT_STACK) stk[(sp++) * KB] = c1;
                    if (n > 0) cur = c0;
                    else if (sp > 0) cur = stk[(--sp) * KB];
                    else cur = REF_EMPTY;
                }
                if (busy) {
                    if ((cur & REF_TYPE_MASK) == REF_TRIS) {
                        const uint32_t first = cur & 0x07FFFFFFu, count = ((cur >> 27) & 7u) + 1u;
                        test_leaf(sc, first, count, r, t_min, first_prim, best);
                        t_max_f = t_max_f32(best.t);
                        cur = sp > 0 ? stk[(--sp) * KB] : REF_EMPTY;
                    }
                    if (cur == REF_EMPTY) {                         // this mesh is done: the ray's next mesh, or the ray is done
                        ++item_k;
                        if (!start_item()) {
                            s_best_id[sl] = best.id;
                            busy = false;
                        }
                    }
                }
            }
        }
        PT_STAMP(e3);
        __syncthreads();
        PT_STAMP(e4);
#ifdef PT_STAMPS
        if (lane == 0) {   // K2's row of the profile: CLASS_DEAD (k_shade never runs a group of that class)
            atomicAdd(&g_prof[CLASS_DEAD][0], 1ull);
            atomicAdd(&g_prof[CLASS_DEAD][1], t_e1 - t_e0);   // phase A
            atomicAdd(&g_prof[CLASS_DEAD][2], t_e2 - t_e1);   // barrier
            atomicAdd(&g_prof[CLASS_DEAD][3], t_e3 - t_e2);   // phase B
            atomicAdd(&g_prof[CLASS_DEAD][4], t_e4 - t_e3);   // barrier
            atomicAdd(&g_prof[CLASS_DEAD][5], (unsigned long long)n_rays);
        }
#endif
        {   // the window's result: one coalesced 4-byte store per slot; the eight PrimRef gathers (material class) go out together
            uint32_t word[WIN / KB];
#pragma unroll
            for (int j = 0; j < WIN / KB; ++j) word[j] = (j >= j_lo && j < j_hi) ? hit_word(sc, s_best_id[(uint32_t)j * KB + threadIdx.x]) : 0u;
#pragma unroll
            for (int j = 0; j < WIN / KB; ++j) if (j >= j_lo && j < j_hi) stnt(&pool.hit_prim[wbase + (uint32_t)j * KB + threadIdx.x], word[j]);
        }
        if (threadIdx.x == 0) { s_win = (uint32_t)atomicAdd(&cnt->win_extend, 1ull); s_nrays = 0; s_next = 0; }   // all of this window's uses are behind the barrier above
        __syncthreads();   // LDS lists are reused by the next window
    }
    nseg = wave_sum(nseg);
    if (nseg && (threadIdx.x & 63u) == 0u) atomicAdd(&cnt->segments, nseg);
#ifdef PT_STAMPS
    __syncthreads();
    if (threadIdx.x < 8 && g_prof[CLASS_DEAD][threadIdx.x]) atomicAdd(&cnt->prof[CLASS_DEAD][threadIdx.x], g_prof[CLASS_DEAD][threadIdx.x]);
#endif
}

// K3: the body of camera.rs:177-226 for the path in slot `s`, executed by all 64 lanes of a wave
// together (it contains wave-level ballots for the work-counter dequeue, K5).
// color += throughput * emitted (camera.rs:182,187). Static mode: into the sample's own sum, which reaches
// the pixel when the sample ends (the reference's order of additions). Dynamic mode: straight into the
// frame accumulator — exact zeros are skipped, NaN/inf are not (they poison the pixel like they do there).
// pixel (row-major) -> its index inside a channel plane of the tiled frame accumulator (PoolD::accum)
PT_DEV uint32_t tiled_index(const PoolD& pool, uint32_t pixel) {
    uint32_t y = (uint32_t)((double)pixel * pool.inv_width);           // within 1 of pixel / width
    int32_t x = (int32_t)(pixel - y * pool.width);
    if (x < 0) { --y; x += (int32_t)pool.width; }
    else if (x >= (int32_t)pool.width) { ++y; x -= (int32_t)pool.width; }
    return ((y >> 3) * pool.tiles_x + ((uint32_t)x >> 3)) * 64u + ((y & 7u) << 3) + ((uint32_t)x & 7u);
}
PT_DEV void add_radiance(const PoolD& pool, uint32_t pixel, V3& rad, V3 c) {
    if (!pool.dynamic) {
        rad = rad + c;
    } else if (!(c.x == 0.0 && c.y == 0.0 && c.z == 0.0)) {
        if (pool.accum_tiled) {
            double* a = pool.accum + tiled_index(pool, pixel);
            unsafeAtomicAdd(a, c.x);
            unsafeAtomicAdd(a + pool.n_tile_pixels, c.y);
            unsafeAtomicAdd(a + 2 * (size_t)pool.n_tile_pixels, c.z);
        } else {
            unsafeAtomicAdd(&pool.accum[3 * (size_t)pixel], c.x);
            unsafeAtomicAdd(&pool.accum[3 * (size_t)pixel + 1], c.y);
            unsafeAtomicAdd(&pool.accum[3 * (size_t)pixel + 2], c.z);
        }
    }
}



// ---- path records of one slot as k_shade consumes them -------------------------------------------------------------
struct SlotIn {
    uint32_t bounce, hw, pixel, sample, draw;   // state, K2's result word, pixel (dynamic mode), sample index, RNG draw counter
    V3 thr;
    RayD ray;
};
// straight from the pool (first group of a window, static mode, unsorted K3). `enable` = false: bystander lane.
// `hw_known`: the caller has the slot's result word already (k_shade's sort keeps the window's words in LDS).
// Whether a slot is alive, idle or dead is in the class of K2's result word (K2 read PoolD::bounce, the slots' STATE array,
// coalesced); the bounce NUMBER of a live path travels in its PathRec. k_shade therefore never reads the state array and writes
// it only when a slot changes state (parked, regenerated from idle, dead) — it used to gather 4 bytes per lane from it and
// scatter 4 bytes per lane back on every bounce of every path.
PT_DEV uint32_t state_of_class(uint32_t hw) {
    const uint32_t cls = hw >> HIT_CLASS_SHIFT;
    return cls == CLASS_IDLE ? SLOT_IDLE : cls == CLASS_DEAD ? SLOT_DEAD : 0u;
}
PT_DEV SlotIn load_slot_global(const PoolD& pool, uint32_t s, bool enable, const uint32_t* hw_known = nullptr) {
    SlotIn in{};
    in.hw = hw_known ? *hw_known : pool.hit_prim[s];
    in.bounce = enable ? state_of_class(in.hw) : SLOT_DEAD;
    if (in.bounce < SLOT_IDLE) {
        uint32_t tail[2];
        in.ray = load_ray(pool, s, in.sample, in.draw, tail);
        if (pool.compact) {
            in.pixel = tail[0];
            in.bounce = tail[1];
            in.thr = V3{1.0, 1.0, 1.0};
            uint32_t p2, b2;
            if (in.bounce != 0u) in.thr = load_path(pool, s, p2, b2);      // a path at bounce 0 has no PathRec
        } else {
            in.thr = load_path(pool, s, in.pixel, in.bounce);
        }
    }
    return in;
}
// Asynchronous fetch of a group's records into the wave's LDS staging area: `global_load_lds` (LDS-DMA) — the data goes
// from HBM to LDS without passing through (or occupying) a single vector register, which is the only way this kernel, at its
// 256-register limit, can have the NEXT group's 6 KB in flight while it computes on the current one.
// [r3] WHOLE SECTORS per instruction. A lane used to fetch the six 16-byte pieces of ITS OWN records, so every wave-instruction
// touched 64 different 64-byte sectors for 16 bytes each and every sector was requested by four instructions (the L2 saw 4x the
// transactions, and a streaming cache policy could not be used: the pieces of a record must find the sector their sibling
// fetched). Now instruction k serves the records of lanes 16k .. 16k+15 with FOUR lanes per RayRec (two per PathRec), each
// fetching a different piece: 16 (32) whole sectors per instruction, every byte requested exactly once. The LDS image —
// wave-uniform base + lane * 16, as the instruction writes — is then simply the records in lane order: RayRec of lane l at
// stage[4 l .. 4 l + 3], PathRec at stage[256 + 2 l ..]. The slots of the other lanes come by ds_bpermute.
// Must be executed by ALL 64 lanes (wave-uniform control flow).
#ifndef PT_STAGE_AUX
#define PT_STAGE_AUX 0             // cache policy of the record stream: 0 default, 2 = nt (MI355X_MICROARCH.md row "nt-weights")
#endif
constexpr int STAGE_CHUNKS = 6;     // 16-byte pieces per lane: the staging area of a wave is uint4[STAGE_CHUNKS * 64]
PT_DEV void stage_fetch(const PoolD& pool, uint32_t s, uint4* stage, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t sk = (uint32_t)__shfl((int)s, (lane >> 2) + 16 * k);
        const char* g = reinterpret_cast<const char*>(&pool.ray[sk]) + 16 * (lane & 3);
        __builtin_amdgcn_global_load_lds((glb_ptr)g, (lds_ptr)(stage + 64 * k), 16, 0, PT_STAGE_AUX);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const uint32_t sm = (uint32_t)__shfl((int)s, (lane >> 1) + 32 * m);
        const char* g = reinterpret_cast<const char*>(&pool.path[sm]) + 16 * (lane & 1);
        __builtin_amdgcn_global_load_lds((glb_ptr)g, (lds_ptr)(stage + 256 + 64 * m), 16, 0, PT_STAGE_AUX);
    }
}
// the staged records of this lane (after the issuing wave's s_waitcnt vmcnt(0): nothing else orders an LDS read behind an LDS-DMA)
PT_DEV SlotIn load_slot_stage(const PoolD& pool, const uint4* stage, int lane, bool enable, uint32_t hw) {
    SlotIn in{};
    in.hw = hw;
    const uint4* rr = stage + 4 * lane;
    const uint4* pr = stage + 256 + 2 * lane;
    const uint4 a = rr[0], b = rr[1], c = rr[2], d = rr[3], e = pr[0], f = pr[1];
    auto f64 = [](uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); };
    in.ray = RayD{V3{f64(a.x, a.y), f64(a.z, a.w), f64(b.x, b.y)}, V3{f64(b.z, b.w), f64(c.x, c.y), f64(c.z, c.w)}, pool.compact ? 0.0 : f64(d.x, d.y)};
    in.sample = d.z;
    in.draw = d.w;
    in.thr = V3{f64(e.x, e.y), f64(e.z, e.w), f64(f.x, f.y)};
    in.pixel = f.z;
    const uint32_t state = state_of_class(hw);
    uint32_t bounce = f.w;                                               // the bounce number rides in the PathRec ...
    if (pool.compact) {                                                  // ... or, with the pixel, in the ray record's time slot
        in.pixel = d.x;
        bounce = d.y;
        if (bounce == 0u) in.thr = V3{1.0, 1.0, 1.0};                    // (the staged PathRec of a fresh path is stale)
    }
    in.bounce = !enable ? SLOT_DEAD : state < SLOT_IDLE ? bounce : state;
    return in;
}
struct NoPrefetch {
    PT_DEV void operator()() const {}
};

// K3: the body of camera.rs:177-226 for the path in slot `s`, executed by all 64 lanes of a wave together (it contains
// wave-level ballots for the work-counter dequeue, K5). `in` = the slot's records; in.bounce == SLOT_DEAD makes the lane
// a bystander that only takes part in the ballots.
// Phases, separated by WAVE-UNIFORM points at which `prefetch()` — the asynchronous fetch of the wave's next group of
// records — may be issued exactly once:
//   A  everything that reads global memory: hit reconstruction, environment lookup, material record, texture values
//   -- P1 (a lane of the wave hit a surface, scene without lights): the arithmetic of B hides the fetch
//   B1 roulette, direction (lights.sample reads the lights' records: scenes with lights prefetch at P1b, after it)
//   B2 pdf, eval, throughput, next ray — pure arithmetic
//   C  work dequeue (a RETURNING atomic: its wait would also wait for a fetch issued before it) -- P2 (nothing hit)
//   D  regeneration (arithmetic), stores
// vmcnt counts loads, stores, atomics and LDS-DMA in issue order, so a fetch can only hide behind a stretch in which no
// younger load is waited for — hence the phase discipline (tex values fetched up front, pt_dev_bsdf.h fetch_tex).
// LIGHTS: the scene has a lights list (World::lights non-empty). The instantiation without compiles lights.sample / lights.pdf, the
// selector draw and the later prefetch point out: p_light = 0 there (camera.rs:199-200), so no result changes.
template <bool LIGHTS, class Prefetch>
// pre_mask / pre_base: the work items of this group's certain-to-end lanes were requested one group AHEAD (k_shade's prefetch point,
// [r3]): pre_mask = those lanes, pre_base = the returning atomic's value in the mask's first lane. 0 = not requested: ask here.
PT_DEV void shade_slot(const SceneD& sc, const CamD& cam, const PoolD& pool, CountersD* cnt, uint64_t seed, uint32_t s, int lane, const SlotIn& in,
                       uint32_t& shard, uint32_t& n_done, uint32_t& n_died, Prefetch&& prefetch, unsigned long long pre_mask = 0ull,
                       unsigned long long pre_base = 0ull, uint32_t pre_shard = 0u) {
    PT_STAMP(1);
    uint32_t bounce = in.bounce;
    const bool alive = bounce != SLOT_DEAD;
    const bool was_idle = bounce == SLOT_IDLE;
    const bool live = alive && !was_idle;
    bool finished = was_idle;
    // A path that ends on a surface (roulette, sampler returned None, depth bound) would make its whole wave run the
    // regeneration code — dequeue, camera ray: ~400 instructions — for one or two lanes: with 64 lanes and a few per cent
    // of such endings per bounce, most surface groups paid for it. Instead the slot is parked as SLOT_IDLE and refilled
    // next iteration together with the other idle slots, where every lane regenerates (class sort: CLASS_IDLE).
    bool parked = false;
    uint32_t pixel = in.pixel, sample = in.sample;
    RayD ray = in.ray;
    V3 thr = in.thr, rad{};
    Rng rng{};
#ifdef PT_STAMPS
    uint32_t prof_class = was_idle ? CLASS_IDLE : CLASS_DEAD;
    if (live) prof_class = in.hw >> HIT_CLASS_SHIFT;
    prof_class = (uint32_t)__builtin_amdgcn_readfirstlane((int)prof_class);
#endif
    // Lanes that are certain to end here — the ray left the scene (class in K2's result word) or the slot is idle — are known
    // before anything is computed: their work items are requested NOW, so that the returning atomic's round trip to the work
    // counter (~3 k cycles, which only the SIMD's other wave could cover) runs under the environment lookup instead of in
    // front of the regeneration. Phase C consumes the answer; paths that end on a surface ask there, as before.
    // [r3] One step further: the class of a wave's NEXT group is known a whole group ahead (the window's sorted result words), so
    // k_shade requests these items at the previous group's prefetch point and hands the pending answer in (pre_mask, pre_base):
    // the round trip — 3-4 us with every CU dequeuing — then runs under a whole group's work instead of under one environment lookup.
    unsigned long long early = 0ull, early_base = 0ull;
    uint32_t early_shard = shard;
    if (pool.dynamic) {
        early = __ballot(alive && (was_idle || (in.hw >> HIT_CLASS_SHIFT) == CLASS_MISS));
        early_shard = shard;
        if (pre_mask != 0ull) { early_base = pre_base; early_shard = pre_shard; }   // (the same lanes by construction: both come from the slots' result words;
                                                                                    //  the wave may have moved on to another shard since it asked)
        else if (early && lane == __ffsll((long long)early) - 1) early_base = atomicAdd(&cnt->work[shard].next, (unsigned long long)__popcll(early));
    }
    // ---- phase A: all global-memory reads of the bounce -------------------------------------------------------------
    bool is_hit = false;
    HitD hit{};
    const MatD* mp = nullptr;
    TexVals tv{};
    LocalFrame lf{};
    PT_STAMP_VAR(a1);
    if (live) {
        if (!pool.dynamic) {
            pixel = s % pool.n_pixels;
            rad = V3{pool.rx[s], pool.ry[s], pool.rz[s]};
        }
        rng = Rng{(uint32_t)seed, (uint32_t)(seed >> 32), pixel, sample, in.draw};
        const uint32_t gid = in.hw & HIT_ID_MASK;
        const bool surface = (in.hw >> HIT_CLASS_SHIFT) != CLASS_MISS && reconstruct_hit(sc, ray, gid, 1e-3, hit);
        PT_STAMP_SET(a1);
        if (!surface) {
            add_radiance(pool, pixel, rad, thr * sample_environment(sc, cam, ray.d));   // camera.rs:180-183
            finished = true;
        } else {
            is_hit = true;
            mp = &sc.mats[hit.mat];
            tv = fetch_tex(sc, *mp, hit);
            lf = make_local_frame(*mp, hit, -ray.d);
            // camera.rs:186-187 — added for every material (zero unless emissive) so that a
            // non-finite throughput poisons the sample exactly as it does in the reference
            V3 emission = mp->kind == MAT_LIGHT ? tv.color : V3{0.0, 0.0, 0.0};
            add_radiance(pool, pixel, rad, thr * emission);
        }
    }
    PT_STAMP(a2);
    const bool any_hit = __ballot(is_hit) != 0ull;
    bool fetched = false;                                              // wave-uniform
    if (any_hit && !LIGHTS) { prefetch(); fetched = true; }            // P1
    // ---- phase B1: roulette and the next direction ------------------------------------------------------------------------
    const double p_light = LIGHTS ? 0.5 : 0.0;                         // :199-200 (the host picks the instantiation by World::lights)
    const double p_bsdf = 1.0 - p_light;
    const V3 wo = -ray.d;
    V3 dir{};
    bool have_dir = false;
    if (is_hit) {
        if (bounce > 5) {                                              // russian roulette :190-196
            double p = clampd(luminance(thr), 0.01, 1.0);
            if (rng_f64(rng) > p) finished = parked = true;
            else thr = thr / p;
        }
        if (!finished) {
            // :201 draws the selector even when there are no lights (p_light = 0: never below it) — then only the counter moves
            double rsel = 1.0;
            bool ok = true;
            if constexpr (LIGHTS) {
                rsel = rng_f64(rng);
                if (rsel < p_light) dir = lights_sample(sc, hit.point, ray.time, rng);
                else ok = mat_sample(sc, *mp, hit, wo, rng, cam.two_pi_scale, tv, lf, dir);
            } else {
                ++rng.draw;
                ok = mat_sample(sc, *mp, hit, wo, rng, cam.two_pi_scale, tv, lf, dir);
            }
            if (!ok) finished = parked = true;                         // :209-211
            else have_dir = true;
        }
    }
    PT_STAMP(b1);
    if (any_hit && !fetched) { prefetch(); fetched = true; }           // P1b
    // ---- phase B2: pdf, eval, throughput, next ray (arithmetic only; lights.pdf reads through the scalar cache) -----------------
    if (have_dir) {
        double bsdf_pdf;
        V3 brdf;
        mat_pdf_eval(sc, *mp, hit, wo, dir, tv, lf, bsdf_pdf, brdf);
        double light_pdf = 0.0;
        if constexpr (LIGHTS) light_pdf = lights_pdf(sc, hit.point, dir, ray.time);
        double pdf = p_bsdf * bsdf_pdf + p_light * light_pdf;
        V3 attenuation = brdf / pdf;
        double e = 1e-3 * signum(dot(dir, hit.gn));                    // :217-222
        ray = make_ray(hit.point + e * hit.gn, dir, ray.time);
        thr = thr * attenuation;
        ++bounce;
        if (bounce >= cam.max_depth) finished = parked = true;         // loop bound :177
    }
#ifdef PT_STAMPS
    const unsigned long long prof_live = __ballot(live), prof_hit = __ballot(is_hit), prof_dir = __ballot(have_dir);
#endif
    PT_STAMP(2);
    // ---- phase C: finished paths accumulate (camera.rs:107) and draw their next work item ------------------------------------
    uint32_t next_pixel = pixel, next_sample = 0, next_row = 0, next_col = 0;
    bool more = false, next_idle = false;
    if (pool.dynamic) {
        // K5: wave ballot + prefix popcount, ONE atomic per wave on the wave's shard of the work counter. When the
        // shard has run dry the wave looks at all shards at once (lane i reads shard i) and moves on to the next one
        // that still has items — without this, slots died while other shards still held work and the frame ended
        // on a long, thin tail.
        parked = parked && pool.defer_regen != 0u;
        // first round: the early request's answer (its lanes are a subset of the finished ones); then whoever is still without
        unsigned long long need = early ? early : __ballot(alive && finished && !parked);
        bool have_base = early != 0ull;
        while (need) {
            const int leader = __ffsll((long long)need) - 1;
            const bool asking = (need >> lane) & 1ull;
            unsigned long long base = early_base;
            uint32_t from = early_shard;                                              // the shard the answer in hand came from
            if (!have_base) {
                base = 0;
                from = shard;
                if (lane == leader) base = atomicAdd(&cnt->work[shard].next, (unsigned long long)__popcll(need));
            }
            have_base = false;
            base = __shfl(base, leader);
            if (asking) {
                const unsigned long long w = shard_item(base + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull)), from);
                if (w < pool.total_work) {
                    more = true;
                    next_idle = !work_to_pixel(pool, w, next_pixel, next_sample, next_row, next_col);
                }
            }
            if (__ballot(asking && !more)) {
                static_assert(WORK_SHARDS == 64, "one lane per shard");
                const unsigned long long live_shards = __ballot(shard_item(cnt->work[lane].next, (uint32_t)lane) < pool.total_work);
                if (live_shards == 0ull) break;                                       // the frame's sample budget is handed out
                const unsigned long long above = live_shards & ~((2ull << shard) - 1ull);   // next live shard after this one, cyclically
                shard = (uint32_t)(__ffsll((long long)(above ? above : live_shards)) - 1);
            }
            need = __ballot(alive && finished && !parked && !more);
        }
    } else if (alive && finished) {
        pool.ax[s] += rad.x; pool.ay[s] += rad.y; pool.az[s] += rad.z;
        next_sample = sample + pool.k;
        more = next_sample < pool.spp_end;
    }
    if (!fetched) prefetch();                                          // P2: behind the dequeue, in front of the regeneration arithmetic
#ifdef PT_STAMPS
    const unsigned long long prof_regen = __ballot(alive && finished && more && !next_idle && !(parked && pool.dynamic));
#endif
    PT_STAMP(3);
    // ---- phase D: regeneration in place, stores -----------------------------------------------------------------------------
    if (alive && finished) {
        if (!was_idle) ++n_done;
        if (parked && pool.dynamic) {
            bounce = SLOT_IDLE;
        } else if (more && next_idle) {
            bounce = SLOT_IDLE;
        } else if (more) {
            rng = Rng{(uint32_t)seed, (uint32_t)(seed >> 32), next_pixel, next_sample, 0u};
            if (!pool.dynamic) divmod_u31(next_pixel, cam.width, next_row, next_col);
            ray = generate_ray(cam, next_row, next_col, rng);
            thr = V3{1.0, 1.0, 1.0};
            rad = V3{0.0, 0.0, 0.0};
            bounce = 0;
            sample = next_sample;
            pixel = next_pixel;
        } else {
            bounce = SLOT_DEAD;
            ++n_died;
        }
    }
    if (alive) {
        const uint32_t state_new = bounce < SLOT_IDLE ? 0u : bounce, state_old = was_idle ? SLOT_IDLE : 0u;
        if (state_new != state_old) pool.bounce[s] = state_new;       // the state array changes only with the slot's state
        if (bounce < SLOT_IDLE) {
            store_ray(pool, s, ray, sample, rng.draw, pixel, bounce);
            if (!pool.compact || bounce != 0u) store_path(pool, s, thr, pixel, bounce);
            if (!pool.dynamic) { pool.rx[s] = rad.x; pool.ry[s] = rad.y; pool.rz[s] = rad.z; }
        }
    }
#ifdef PT_STAMPS
    PT_STAMP(4);
    if (lane == 0) {   // block-local sums in LDS (global atomics here would themselves be what the next group waits for)
        atomicAdd(&g_prof[prof_class][0], 1ull);
        atomicAdd(&g_prof[prof_class][8], (unsigned long long)__popcll(prof_live));
        atomicAdd(&g_prof[prof_class][9], (unsigned long long)__popcll(prof_hit));
        atomicAdd(&g_prof[prof_class][10], (unsigned long long)__popcll(prof_dir));
        atomicAdd(&g_prof[prof_class][11], (unsigned long long)__popcll(prof_regen));
        atomicAdd(&g_prof[prof_class][1], (t_a1 ? t_a1 : t_a2) - t_1);   // records unpacked, hit reconstructed
        atomicAdd(&g_prof[prof_class][6], t_a2 - (t_a1 ? t_a1 : t_a2));  // environment / textures
        atomicAdd(&g_prof[prof_class][7], t_b1 - t_a2);                  // roulette + direction
        atomicAdd(&g_prof[prof_class][2], t_2 - t_1);
        atomicAdd(&g_prof[prof_class][3], t_3 - t_2);
        atomicAdd(&g_prof[prof_class][4], t_4 - t_3);
        atomicAdd(&g_prof[prof_class][5], t_4 - t_1);
    }
#endif
}

constexpr int SORT_WINDOW = SORT_WINDOW_SLOTS;   // slots sorted together by k_shade<true, *>
// Every thread packs the class keys of its SORT_WINDOW / BLOCK slots into ONE 32-bit word, 4 bits each. A 4096-slot window
// (16 keys) overflowed that word in round 2 and the kernel hung: the bound is a compile error now, not a comment.
static_assert((SORT_WINDOW / BLOCK) * 4 <= 32, "k_shade: the per-thread `keys` word holds at most eight 4-bit class keys — widen it before enlarging SORT_WINDOW");
static_assert(N_CLASSES <= 16, "k_shade: a class key is 4 bits wide (and the class field of K2's result word is bits 28..31)");
static_assert(SORT_WINDOW % BLOCK == 0 && SORT_WINDOW / 64 == 32, "k_shade: one half-wave scans the 32 group counts of a class");
static_assert(SORT_WINDOW <= 65536, "k_shade: s_perm holds 16-bit slot offsets");
#ifndef PT_DEQUEUE_AHEAD
#define PT_DEQUEUE_AHEAD 1          // 0: certain-to-end lanes request their work items at the start of their own group (the round-2 form)
#endif
#ifndef PT_K3_PREFETCH
#define PT_K3_PREFETCH 1            // 0: every group's records straight from the pool (the round-1 form), for A/B
#endif

// K3 launcher kernel. SORT = false: blocks walk the pool in 256-slot chunks, lane i shades slot i.
// SORT = true (default): a block draws a WINDOW of 2048 slots from a queue, counting-sorts their indices
// by class in LDS (miss, one class per material kind, idle, dead), then its four waves pull groups of 64
// same-class slots from an LDS cursor until the window is done — waves execute one material's code
// instead of serialising through all of them, the expensive classes go first and are spread over all
// waves of the block (work stealing), and every slot's records are moved whole by its own lane.
// KB: threads per block (256, or [r3] 512 with a 4096-slot window: the sort's barriers and the window's end are paid once per twice
// as many slots and eight waves level a window's end better than four; one block per CU then).
template <bool SORT, int MINW, bool LIGHTS, int KB = BLOCK, int PER = SORT_WINDOW / BLOCK>
__global__ __launch_bounds__(KB, KB == BLOCK ? MINW : 1) void k_shade(SceneD sc, CamD cam, PoolD pool, CountersD* cnt, uint64_t seed) {
    uint32_t n_done = 0, n_died = 0;   // per thread and launch: far below 2^32 (64-bit counters here were the kernel's only spills)
    const int lane = (int)(threadIdx.x & 63u);
#ifdef PT_STAMPS
    for (uint32_t i = threadIdx.x; i < (N_CLASSES + 1) * PROF_COLS; i += KB) (&g_prof[0][0])[i] = 0ull;
    __syncthreads();
#endif
    uint32_t shard = blockIdx.x % WORK_SHARDS;   // work-counter shard this wave draws from (wave-uniform; moves on when it runs dry)
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->win_extend = 0;
    if (ldu(&cnt->alive) == 0ull) return;   // (see k_extend; a block subtracts its dead slots when it has run out of windows: zero means every window of the pool has been shaded)
    if (!SORT) {
        // n_alloc is a multiple of 256: whole waves run every chunk (wave ballots inside shade_slot)
        for (uint32_t base = blockIdx.x * KB; base < pool.n_alloc; base += gridDim.x * KB) {
            const uint32_t s = base + threadIdx.x;
            const SlotIn in = load_slot_global(pool, s, true);
            shade_slot<LIGHTS>(sc, cam, pool, cnt, seed, s, lane, in, shard, n_done, n_died, NoPrefetch{});
        }
    } else {
        constexpr int WIN = KB * PER;                           // slots per window: eight (or sixteen) per thread
        static_assert(PER * 4 <= 64 && WIN <= 65536 && WIN % 64 == 0, "sixteen 4-bit keys per thread at most, 16-bit slot offsets");
        __shared__ uint16_t s_perm[WIN];
        __shared__ uint32_t s_hw[WIN];                          //  8 KB: K2's result words of the window
        constexpr uint32_t NCLASS = N_CLASSES, K_DEAD = CLASS_DEAD;   // miss, one per material kind, idle, dead
        __shared__ uint32_t s_cnt[NCLASS][WIN / 64];   // [class][64-slot group of the window, in slot order]
        __shared__ uint32_t s_hist[NCLASS], s_next;
        __shared__ uint4 s_stage[KB / 64][STAGE_CHUNKS * 64];   // 24 KB: one staging area per wave (stage_fetch)
        constexpr int NGRP = WIN / 64;
        const int wave = (int)(threadIdx.x >> 6);
        __shared__ uint32_t s_win;
        const uint32_t n_windows = pool.n_alloc / WIN;
        // (Handing the queue's END out in half windows, as k_extend2 does, was measured here too — the other half's slots counted as
        // dead in the sort —: K3 +-0 on the 33.6 M-slot pool, +1.7 % on a 16.8 M-slot one (+2.4 % with the window's loads predicated): a half
        // window pays the whole window's sort and barriers and levels its eight waves' end worse.)
        // The window index of the NEXT round is drawn by thread 0 when its wave has run out of groups and published by the
        // barrier that ends the window anyway: no barrier of its own, and the atomic's round trip (2-3 k cycles the whole
        // block used to sit out at the top of every window) runs while the other waves finish their groups.
        if (threadIdx.x == 0) s_win = (uint32_t)atomicAdd(&cnt->win_shade, 1ull);
        __syncthreads();
        for (;;) {
            PT_STAMP(w0);
            const uint32_t win = s_win;
            if (win >= n_windows) break;
            if (threadIdx.x == 0) s_next = 0;      // (every wave is past the previous window's last grab; the first one of this window comes three barriers later)
            const uint32_t wbase = win * WIN;
            // classify; STABLE counting sort (slot order is kept inside a class, so the work items a
            // wave dequeues — consecutive pixels of one tile — stay together in a group).
            typename std::conditional<(PER > 8), uint64_t, uint32_t>::type keys = 0;   // PER x 4-bit class keys: K2 left the class in the top bits of its result word
            uint32_t rank[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                // the window's result words stay in LDS: the groups take theirs from here instead of gathering 4 bytes per lane
                // from the pool a second time (a 32-byte sector each)
                const uint32_t hw = pool.hit_prim[wbase + (uint32_t)j * KB + threadIdx.x];
                s_hw[(uint32_t)j * KB + threadIdx.x] = hw;
                keys |= (decltype(keys))(hw >> HIT_CLASS_SHIFT) << (4 * j);
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t key = (uint32_t)(keys >> (4 * j)) & 15u;
                rank[j] = 0;
                // only the classes present among the wave's 64 slots cost a ballot (typically two to four); lane k keeps class k's
                // count and stores it — one LDS store per wave and chunk ([r3]; lane 0 used to zero eleven words and write the rest)
                uint32_t mine = 0;
                unsigned long long todo = ~0ull;
                while (todo) {
                    const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)key, __ffsll((long long)todo) - 1));
                    const unsigned long long m = __ballot(key == k);
                    if (key == k) rank[j] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if ((uint32_t)lane == k) mine = (uint32_t)__popcll(m);
                    todo &= ~m;
                }
                if ((uint32_t)lane < NCLASS) s_cnt[lane][j * (KB / 64) + wave] = mine;
            }
            __syncthreads();
            // exclusive prefix over the groups in slot order, per class: NGRP = 32 lanes scan one class with five shuffles (the
            // round-1 form — one thread per class walking its 32 counts through LDS, a chain of 32 dependent reads the other
            // 245 threads waited for at the barrier — was a fifth of the sort's time); the block's waves share the classes
            static_assert(NGRP == 32 || NGRP == 64 || NGRP == 128, "one half-wave or one wave per class (two counts per lane for 128)");
            if constexpr (NGRP <= 64) {
            constexpr uint32_t PER_PASS = 64u / (uint32_t)NGRP;          // classes a wave scans at once
            for (uint32_t k = (uint32_t)wave * PER_PASS + (uint32_t)lane / (uint32_t)NGRP; k < NCLASS; k += (KB / 64) * PER_PASS) {
                const int g = lane % NGRP;
                const uint32_t c = s_cnt[k][g];
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < NGRP; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, NGRP);
                    if (g >= d) incl += up;
                }
                s_cnt[k][g] = incl - c;
                if (g == NGRP - 1) s_hist[k] = incl;
            }
            } else {
            for (uint32_t k = (uint32_t)wave; k < NCLASS; k += KB / 64) {
                const uint32_t c0 = s_cnt[k][2 * lane], c1 = s_cnt[k][2 * lane + 1];
                uint32_t incl = c0 + c1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                    if (lane >= d) incl += up;
                }
                s_cnt[k][2 * lane] = incl - c0 - c1;
                s_cnt[k][2 * lane + 1] = incl - c1;
                if (lane == 63) s_hist[k] = incl;
            }
            }
            __syncthreads();
            // first position of every class: lane k of each wave sums the histogram below k (eleven LDS reads by eleven lanes)
            // and the slots fetch theirs by a lane shuffle ([r3]; every thread used to build the table and select from it with
            // eleven compares per slot)
            uint32_t my_base = 0;
            if ((uint32_t)lane < NCLASS)
                for (uint32_t k = 0; k < (uint32_t)lane; ++k) my_base += s_hist[k];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t key = (uint32_t)(keys >> (4 * j)) & 15u;
                const uint32_t cb = (uint32_t)__shfl((int)my_base, (int)key);
                const uint32_t pos = cb + s_cnt[key][j * (KB / 64) + wave] + rank[j];
                s_perm[pos] = (uint16_t)(j * KB + threadIdx.x);
            }
            __syncthreads();
            const uint32_t n_live = (uint32_t)WIN - s_hist[K_DEAD];
            PT_STAMP(w1);
            // groups are taken from the END of the sorted order: the expensive classes (principled, glass) sort
            // last, and starting with them keeps the four waves level when the window runs out (the cheap
            // misses fill the gaps). Lanes past n_live in the top group are bystanders.
            const uint32_t n_groups = (n_live + 63u) / 64u;
            auto grab = [&]() -> uint32_t {                          // this wave's next group of the window (wave-uniform)
                uint32_t g = 0;
                if (lane == 0) g = atomicAdd(&s_next, 1u);
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
            };
            auto slot_of = [&](uint32_t g, bool& enable) -> uint32_t {
                const uint32_t q = (n_groups - 1u - g) * 64u + (uint32_t)lane;
                enable = q < n_live;
                return wbase + s_perm[enable ? q : 0u];
            };
            // The wave's next group is reserved and its records requested (LDS-DMA, stage_fetch) from inside shade_slot, at the
            // point where the current group's arithmetic can hide the fetch; the first group of a window comes straight from
            // the pool. Dynamic mode only (the static mode's extra per-slot arrays are not staged).
            const bool use_stage = PT_K3_PREFETCH && pool.dynamic != 0u;
            uint4* stage = s_stage[wave];
            uint32_t g = grab();
            bool staged = false;
            unsigned long long pre_mask = 0ull, pre_base = 0ull;      // work items requested a group ahead (shade_slot)
            uint32_t pre_shard = 0u;
            while (g < n_groups) {
                PT_STAMP(0);
                bool enable;
                const uint32_t s = slot_of(g, enable);
                SlotIn in;
                const uint32_t hw = s_hw[s - wbase];
                if (staged) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA has landed (and this wave's older stores with it)
                    in = load_slot_stage(pool, stage, lane, enable, hw);
                } else {
                    in = load_slot_global(pool, s, enable, &hw);
                }
                PT_DRAIN();
#ifdef PT_STAMPS
                PT_STAMP(ld);
                if (lane == 0) atomicAdd(&g_prof[N_CLASSES][5], t_ld - t_0);   // record wait (load or staged), all classes
#endif
                uint32_t g_next = n_groups;
                bool staged_next = false;
                unsigned long long pre_mask_next = 0ull, pre_base_next = 0ull;
                uint32_t pre_shard_next = 0u;
                auto prefetch = [&]() {
                    g_next = grab();
                    if (use_stage && g_next < n_groups) {
                        bool en;
                        const uint32_t sn = slot_of(g_next, en);
                        __builtin_amdgcn_sched_barrier(0);            // nothing of the current group's loads may sink below the DMA
                        stage_fetch(pool, sn, stage, lane);
                        __builtin_amdgcn_sched_barrier(0);
                        staged_next = true;
#if PT_DEQUEUE_AHEAD
                        // the next group's lanes that are certain to end there (ray left the scene / idle slot): their work items now.
                        // Scenes without a lights list only: K3 -1.2 % (scene 6), -0.7 % (scene 5); the lights instantiation, three
                        // registers from the limit, got 0.9 % SLOWER with it (closed scenes have next to no leaving rays anyway).
                        if constexpr (!LIGHTS) {
                        const uint32_t cn = s_hw[sn - wbase] >> HIT_CLASS_SHIFT;
                        pre_mask_next = __ballot(en && (cn == CLASS_MISS || cn == CLASS_IDLE));
                        pre_shard_next = shard;
                        if (pre_mask_next && lane == __ffsll((long long)pre_mask_next) - 1)
                            pre_base_next = atomicAdd(&cnt->work[shard].next, (unsigned long long)__popcll(pre_mask_next));
                        }
#endif
                    }
                };
                shade_slot<LIGHTS>(sc, cam, pool, cnt, seed, s, lane, in, shard, n_done, n_died, prefetch, pre_mask, pre_base, pre_shard);
                pre_mask = pre_mask_next;
                pre_base = pre_base_next;
                pre_shard = pre_shard_next;
#ifdef PT_STAMPS
                if (lane == 0) atomicAdd(&g_prof[N_CLASSES][4], 1ull);
#endif
                g = g_next;
                staged = staged_next;
            }
            PT_STAMP(w2);
            if (threadIdx.x == 0) s_win = (uint32_t)atomicAdd(&cnt->win_shade, 1ull);   // everybody read s_win before this window's first barrier
            __syncthreads();   // LDS is reused by the next window
#ifdef PT_STAMPS
            PT_STAMP(w3);
            if (lane == 0) {
                atomicAdd(&g_prof[N_CLASSES][0], 1ull);
                atomicAdd(&g_prof[N_CLASSES][1], t_w1 - t_w0);     // window draw + classification + sort
                atomicAdd(&g_prof[N_CLASSES][2], t_w2 - t_w1);     // shading groups
                atomicAdd(&g_prof[N_CLASSES][3], t_w3 - t_w2);     // waiting for the block's other waves
            }
#endif
        }
    }
    const unsigned long long w_done = wave_sum(n_done), w_died = wave_sum(n_died);
    if (lane == 0) {
        if (w_done) atomicAdd(&cnt->samples, w_done);
        if (w_died) atomicSub(&cnt->alive, w_died);
    }
#ifdef PT_STAMPS
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (N_CLASSES + 1) * PROF_COLS; i += KB)
        if ((&g_prof[0][0])[i]) atomicAdd(&cnt->prof[0][0] + i, (&g_prof[0][0])[i]);
#endif
}

// accum[p*3+c] += sum over the k slots of pixel p, in slot order (k == 1: the exact
// sample-order sum the reference computes at camera.rs:106-108)
__global__ __launch_bounds__(BLOCK) void k_resolve(PoolD pool, double* accum) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < pool.n_pixels; p += gridDim.x * BLOCK) {
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (uint32_t j = 0; j < pool.k; ++j) {
            const uint32_t s = j * pool.n_pixels + p;
            if (j == 0) { sx = pool.ax[s]; sy = pool.ay[s]; sz = pool.az[s]; }
            else { sx += pool.ax[s]; sy += pool.ay[s]; sz += pool.az[s]; }
        }
        accum[3 * (size_t)p] += sx;
        accum[3 * (size_t)p + 1] += sy;
        accum[3 * (size_t)p + 2] += sz;
    }
}

// dynamic mode with the tiled frame accumulator (PoolD::accum_tiled): accum[p*3+c] += plane c's sum of pixel p — once per render
__global__ __launch_bounds__(BLOCK) void k_detile(PoolD pool, double* accum) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < pool.n_pixels; p += gridDim.x * BLOCK) {
        const double* a = pool.accum + tiled_index(pool, p);
        accum[3 * (size_t)p] += a[0];
        accum[3 * (size_t)p + 1] += a[pool.n_tile_pixels];
        accum[3 * (size_t)p + 2] += a[2 * (size_t)pool.n_tile_pixels];
    }
}

// ---- the frame's END: compaction of the thinning pool (dynamic mode) -----------------------------------------------------------
// When the sample budget is handed out the slots die one by one, and for the last ~60 iterations both kernels sweep a pool
// that is mostly dead — every window pays its sort and its barriers for a handful of live paths (3-6 % of a frame on the pool
// sizes one rank of a multi-GPU frame uses). The host, which polls the live count anyway, then moves the survivors to the
// FRONT: the live slots beyond the new end L ("movers") go into the dead slots below it ("holes"), and every later launch covers
// [0, L) only. k_compact_scan lists both kinds (one atomic per wave and list, any order); k_compact_move copies mover i's two
// records and its state into hole i and marks the old slot dead. Which slot a path sits in decides nothing (the RNG is keyed by
// pixel and sample, the frame accumulator by pixel): no result changes.
__global__ __launch_bounds__(BLOCK) void k_compact_scan(PoolD pool, uint32_t new_end, uint32_t* holes, uint32_t* movers, uint32_t* counts /* [0] holes, [1] movers */,
                                                        uint32_t cap) {
    // [r3] A block takes 4096 slots at a time (n_alloc is a multiple of 8192), keeps their sixteen states per thread in registers,
    // ranks its holes and movers inside the block and reserves the block's stretch of each list with ONE atomic: one atomic per wave and
    // list on two addresses (the first form) serialised — 8 M of them, 47 ms, on a 268 M-slot pool (rocprofv3, config 2).
    constexpr int PER = 16, SUPER = PER * BLOCK;
    __shared__ uint32_t s_wave[2][BLOCK / 64], s_base[2];
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    for (uint32_t base = blockIdx.x * SUPER; base < pool.n_alloc; base += gridDim.x * SUPER) {
        uint32_t hole_bits = 0, mover_bits = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const uint32_t s = base + (uint32_t)j * BLOCK + threadIdx.x;
            const bool dead = pool.bounce[s] == SLOT_DEAD;
            hole_bits |= (uint32_t)(s < new_end && dead) << j;
            mover_bits |= (uint32_t)(s >= new_end && !dead) << j;
        }
        uint32_t mine[2] = {(uint32_t)__popc(hole_bits), (uint32_t)__popc(mover_bits)}, before[2];
        for (int which = 0; which < 2; ++which) {              // exclusive prefix of the threads' counts inside the wave, wave totals to LDS
            uint32_t incl = mine[which];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
                if (lane >= d) incl += up;
            }
            before[which] = incl - mine[which];
            if (lane == 63) s_wave[which][wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            uint32_t tot = 0;
            for (int w = 0; w < BLOCK / 64; ++w) tot += s_wave[threadIdx.x][w];
            s_base[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x], tot) : 0u;
        }
        __syncthreads();
        for (int which = 0; which < 2; ++which) {
            uint32_t at = s_base[which] + before[which];
            for (int w = 0; w < wave; ++w) at += s_wave[which][w];
            uint32_t bits = which == 0 ? hole_bits : mover_bits;
            uint32_t* list = which == 0 ? holes : movers;
            while (bits) {
                const int j = __ffs((int)bits) - 1;
                bits &= bits - 1u;
                if (at < cap) list[at] = base + (uint32_t)j * BLOCK + threadIdx.x;
                ++at;
            }
        }
        __syncthreads();                                        // s_wave / s_base are reused by the next 4096 slots
    }
}
__global__ __launch_bounds__(BLOCK) void k_compact_move(PoolD pool, const uint32_t* holes, const uint32_t* movers, const uint32_t* counts, uint32_t cap) {
    const uint32_t n = counts[1] < counts[0] ? counts[1] : counts[0];          // movers <= holes by construction (live slots <= new end)
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n && i < cap; i += gridDim.x * BLOCK) {
        const uint32_t src = movers[i], dst = holes[i];
        const u4v* r = reinterpret_cast<const u4v*>(&pool.ray[src]);
        u4v* w = reinterpret_cast<u4v*>(&pool.ray[dst]);
        const u4v a = r[0], b = r[1], c = r[2], d = r[3];
        w[0] = a; w[1] = b; w[2] = c; w[3] = d;
        const u4v* pr = reinterpret_cast<const u4v*>(&pool.path[src]);
        u4v* pw = reinterpret_cast<u4v*>(&pool.path[dst]);
        for (unsigned k = 0; k < sizeof(PathRec) / 16; ++k) pw[k] = pr[k];
        pool.bounce[dst] = pool.bounce[src];
        pool.bounce[src] = SLOT_DEAD;
    }
}

// camera.rs:109-114,128-130: mean, sqrt gamma, clamp, truncate to u8
__global__ __launch_bounds__(BLOCK) void k_quantise(const double* accum, uint32_t n, double scale, uint8_t* rgb8) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        double c = accum[i] * scale;
        double g = sqrt(fmax(c, 0.0));
        double q = clampd(g, 0.0, 0.999) * 256.0;
        rgb8[i] = (q != q) ? (uint8_t)0 : (uint8_t)q;
    }
}

// Debug/parity probe: closest hit + reconstructed HitInfo for a batch of arbitrary rays.
// out[15*i..] = {hit, t, prim_id, u, v, front, p.xyz, gn.xyz, sn.xyz}
__global__ __launch_bounds__(BLOCK) void k_probe(SceneD sc, const double* rays /* o.xyz d.xyz time */, uint32_t n, double* out) {
    __shared__ uint32_t stack[TRAVERSAL_STACK * BLOCK];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const double* q = rays + 7 * (size_t)i;
        RayD r = make_ray(V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}, q[6]);
        Closest c = closest_hit(sc, r, 1e-3, &stack[threadIdx.x]);
        double* o = out + 15 * (size_t)i;
        for (int j = 0; j < 15; ++j) o[j] = 0.0;
        HitD h;
        if (c.id != HIT_NONE && reconstruct_hit(sc, r, c.id, 1e-3, h)) {
            o[0] = 1.0; o[1] = c.t; o[2] = (double)c.id; o[3] = h.u; o[4] = h.v; o[5] = h.front ? 1.0 : 0.0;
            o[6] = h.point.x; o[7] = h.point.y; o[8] = h.point.z;
            o[9] = h.gn.x; o[10] = h.gn.y; o[11] = h.gn.z;
            o[12] = h.sn.x; o[13] = h.sn.y; o[14] = h.sn.z;
        }
    }
}

// Elementwise probes of the device arithmetic (sqrt/div/fma-free mul-add, libm calls, RNG)
// so that tests can compare them with the host bit for bit / ulp for ulp.
__global__ void k_math_probe(int which, const double* in, uint32_t n, double* out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a = in[2 * (size_t)i], b = in[2 * (size_t)i + 1], r = 0.0;
        switch (which) {
        case 0: r = sqrt(a); break;
        case 1: r = a / b; break;
        case 2: r = a * b + a; break;      // must NOT be fused
        case 3: r = detmath::sin(a); break;
        case 4: r = detmath::cos(a); break;
        case 5: r = detmath::acos(a); break;
        case 6: r = detmath::atan2(a, b); break;
        case 7: r = detmath::pow(a, b); break;
        case 8: r = detmath::log2(a); break;
        case 10: r = detmath::log(a); break;
        case 11: r = detmath::exp(a); break;
        case 9: {
            Rng g{(uint32_t)(long long)a, 0u, (uint32_t)(long long)b, 7u, (uint32_t)i};
            r = rng_f64(g);
            break;
        }
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------- launchers
static inline dim3 grid_for(uint32_t n, int max_blocks) {
    uint32_t b = (n + BLOCK - 1) / BLOCK;
    if (b > (uint32_t)max_blocks) b = (uint32_t)max_blocks;
    if (b == 0) b = 1;
    return dim3(b);
}
void launch_init(const CamD& cam, const PoolD& pool, uint64_t seed, int max_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_init, grid_for(pool.n_alloc, max_blocks), dim3(BLOCK), 0, st, cam, pool, seed);
}
typedef void (*extend2_fn)(SceneD, PoolD, CountersD*);
static extend2_fn pick_extend2(int code) {   // code = stack entries * 10 + min blocks per CU
    switch (code) {
    case 163: return k_extend2<16, 3>;
    case 164: return k_extend2<16, 4>;
    case 204: return k_extend2<20, 4>;
    case 283: return k_extend2<28, 3>;   // deep trees (the GPU builder's LBVHs of million-triangle meshes): 47.5 / 51.5 KB of LDS per block,
    case 323: return k_extend2<32, 3>;   // still three blocks per CU (160 KB)
    case 1164: return k_extend2<16, 4, 64>;    // [r3] block-size A/B (PT_EXT2): one wave per block and 512-slot windows ...
    case 2164: return k_extend2<16, 4, 128>;
    case 8164: return k_extend2<16, 4, 512>;   // ... to eight waves and 4096-slot windows
    default: return k_extend2<24, 3>;   // 24 stack entries: 43.5 KB of LDS per block, three blocks per CU
    }
}
static int extend2_threads(int code) { return code >= 1000 ? (code / 1000) * 64 : BLOCK; }
typedef void (*extend_fn)(SceneD, PoolD, CountersD*);
static extend_fn pick_extend_batch(uint32_t flat, uint32_t pairs) {
    return !flat ? k_extend<false, false> : pairs ? k_extend<true, true> : k_extend<true, false>;
}
void launch_extend(const SceneD& sc, const PoolD& pool, CountersD* cnt, int max_blocks, int code, hipStream_t st) {   // code: pt_render.cpp extend_code
    if (code <= -100) {
        const int kb = extend2_threads(-code);
        uint32_t blocks = pool.n_alloc / (uint32_t)(EXT_WINDOW / BLOCK * kb);   // one per window at most
        if (blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
        if (blocks == 0) blocks = 1;
        hipLaunchKernelGGL(pick_extend2(-code), dim3(blocks), dim3((uint32_t)kb), 0, st, sc, pool, cnt);
    }
    else hipLaunchKernelGGL(pick_extend_batch(sc.tlas_flat, sc.flat_pairs), grid_for(pool.n_alloc, max_blocks), dim3(BLOCK), 0, st, sc, pool, cnt);
}
typedef void (*shade_fn)(SceneD, CamD, PoolD, CountersD*, uint64_t);
static shade_fn pick_shade(int variant, bool lights) {   // variant = sort*10 + min waves per SIMD; 22 = sorted, 512 threads / 4096-slot windows
    switch (variant) {
    case 2: return lights ? k_shade<false, 2, true> : k_shade<false, 2, false>;
    case 3: return lights ? k_shade<false, 3, true> : k_shade<false, 3, false>;
    case 12: return lights ? k_shade<true, 2, true> : k_shade<true, 2, false>;
    case 13: return lights ? k_shade<true, 3, true> : k_shade<true, 3, false>;
    case 22: return lights ? k_shade<true, 2, true, 512> : k_shade<true, 2, false, 512>;
    case 32: case 42: return lights ? k_shade<true, 2, true, 512, 16> : k_shade<true, 2, false, 512, 16>;
    case 52: return lights ? k_shade<true, 2, true, 256, 16> : k_shade<true, 2, false, 256, 16>;   // A/B: 256 threads over 4096-slot windows (window size vs block size)
    default: return lights ? k_shade<false, 2, true> : k_shade<false, 2, false>;
    }
}
static int shade_threads(int variant) { return variant == 22 || variant == 32 || variant == 42 ? 512 : BLOCK; }
static int shade_window(int variant) { return variant == 32 ? 8192 : variant == 22 || variant == 52 ? 4096 : SORT_WINDOW; }
void launch_shade(const SceneD& sc, const CamD& cam, const PoolD& pool, CountersD* cnt, uint64_t seed, int max_blocks, int variant,
                  hipStream_t st, uint32_t wide_window_min) {
    // 42: 8192-slot windows while the pool holds at least PT_WIDE_WINDOW_MIN of them per block launched, 4096-slot windows below
    // (a thinner pool — smaller frames, one rank's share, the frame's end after compaction — levels its end better with more, smaller windows)
    if (variant == 42) variant = pool.n_alloc / 8192u >= (uint32_t)max_blocks * (wide_window_min ? wide_window_min : 1u) ? 32 : 22;
    const int kb = shade_threads(variant);
    uint32_t blocks = variant >= 10 ? pool.n_alloc / (uint32_t)shade_window(variant) : (pool.n_alloc + (uint32_t)kb - 1u) / (uint32_t)kb;   // one block per window / chunk
    if (blocks > (uint32_t)max_blocks) blocks = (uint32_t)max_blocks;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(pick_shade(variant, sc.n_lights != 0u), dim3(blocks), dim3((uint32_t)kb), 0, st, sc, cam, pool, cnt, seed);
}
void launch_resolve(const PoolD& pool, double* accum, int max_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_resolve, grid_for(pool.n_pixels, max_blocks), dim3(BLOCK), 0, st, pool, accum);
}
void launch_compact(const PoolD& pool, uint32_t new_end, uint32_t* holes, uint32_t* movers, uint32_t* counts, uint32_t cap, int max_blocks, hipStream_t st) {
    (void)hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_compact_scan, grid_for(pool.n_alloc / 16u, max_blocks), dim3(BLOCK), 0, st, pool, new_end, holes, movers, counts, cap);
    hipLaunchKernelGGL(k_compact_move, grid_for(cap, max_blocks), dim3(BLOCK), 0, st, pool, holes, movers, counts, cap);
}
void launch_detile(const PoolD& pool, double* accum, int max_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_detile, grid_for(pool.n_pixels, max_blocks), dim3(BLOCK), 0, st, pool, accum);
}
void launch_quantise(const double* accum, uint32_t n, double scale, uint8_t* rgb8, hipStream_t st) {
    hipLaunchKernelGGL(k_quantise, grid_for(n, 4096), dim3(BLOCK), 0, st, accum, n, scale, rgb8);
}
void launch_probe(const SceneD& sc, const double* rays, uint32_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_probe, grid_for(n, 2048), dim3(BLOCK), 0, st, sc, rays, n, out);
}
void launch_math_probe(int which, const double* in, uint32_t n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_math_probe, grid_for(n, 2048), dim3(BLOCK), 0, st, which, in, n, out);
}
int kernel_occupancy_blocks(int which, int variant, bool lights) {
    int nb = 0;
    const void* f = which == 0 ? (variant <= -100 ? (const void*)pick_extend2(-variant) : (const void*)pick_extend_batch(variant <= -2, variant == -3)) : (const void*)pick_shade(variant, lights);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, which == 1 ? shade_threads(variant) : variant <= -100 ? extend2_threads(-variant) : BLOCK, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}

}  // namespace pt
