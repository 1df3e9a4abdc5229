// Device-side f64 vector maths, quaternion frames and the counter-based RNG.
// Arithmetic contract: plain IEEE binary64, one rounding per
// written operation — the translation unit is built with -ffp-contract=off — in the
// operation order of the reference's expressions (glam 0.29 semantics, vec3.rs,
// bsdf/sampling.rs:8-16), so results are reproducible to the last bit except for the
// elementary functions, which come from pt_detmath.h (deterministic, shared with the oracle's
// "det" mode) — so a whole render is reproducible bit for bit on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_detmath.h"

#define PT_DEV __device__ __forceinline__

namespace pt {

constexpr double D_PI = 3.14159265358979323846264338327950288;
constexpr double D_INF = __builtin_huge_val();

struct V3 {
    double x, y, z;
};
PT_DEV V3 mk(double x, double y, double z) { return V3{x, y, z}; }
PT_DEV V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
PT_DEV V3 splat(double s) { return V3{s, s, s}; }
PT_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
PT_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
PT_DEV V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_DEV V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
PT_DEV V3 operator-(double s, V3 a) { return {s - a.x, s - a.y, s - a.z}; }
PT_DEV double dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
PT_DEV V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
PT_DEV double length_squared(V3 a) { return dot(a, a); }
PT_DEV double length(V3 a) { return sqrt(dot(a, a)); }
PT_DEV V3 normalize(V3 a) { return a * (1.0 / length(a)); }   // glam: v * length_recip
PT_DEV bool is_zero(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
PT_DEV double clampd(double x, double lo, double hi) {         // Rust f64::clamp
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
PT_DEV double signum(double x) {                                // Rust f64::signum
    if (x != x) return x;
    return __builtin_signbit(x) ? -1.0 : 1.0;
}
PT_DEV double powi2(double x) { return x * x; }
PT_DEV double powi5(double x) {
    double x2 = x * x;
    double x4 = x2 * x2;
    return x * x4;
}
PT_DEV V3 vlerp(V3 a, V3 b, double s) { return a * (1.0 - s) + b * s; }
PT_DEV double flerp(double a, double b, double s) { return a + (b - a) * s; }
PT_DEV V3 reflect(V3 i, V3 n) { return i - n * (2.0 * dot(i, n)); }
PT_DEV V3 refract(V3 i, V3 n, double eta) {
    double n_dot_i = dot(n, i);
    double k = 1.0 - eta * eta * (1.0 - n_dot_i * n_dot_i);
    if (k >= 0.0) return eta * i - (eta * n_dot_i + sqrt(k)) * n;
    return V3{0.0, 0.0, 0.0};
}
PT_DEV double luminance(V3 c) { return 0.2126 * c.x + 0.7152 * c.y + 0.0722 * c.z; }

// Elementary functions as real (non-inlined) device functions: k_shade calls sincos at ten sites and acos/atan2 at four;
// inlined, each copy brings its ~40 registers of temporaries to a spot that is already at the kernel's 256-register limit
// (the round-2 sin/cos kernels carry double-double terms) and the compiler spilled 300 B per lane. One shared body keeps
// the callers' allocation where it was and shrinks the kernel's code by a third.
struct SinCos {
    double s, c;
};
#ifndef PT_DETMATH_INLINE
#define PT_DM_CALL __device__ __noinline__
#else
#define PT_DM_CALL __device__ __forceinline__
#endif
PT_DM_CALL SinCos dev_sincos(double x) {
    SinCos r;
    detmath::sincos(x, r.s, r.c);
    return r;
}
PT_DM_CALL double dev_acos(double x) { return detmath::acos(x); }
PT_DM_CALL double dev_atan2(double y, double x) { return detmath::atan2(y, x); }
PT_DM_CALL double dev_pow(double x, double y) { return detmath::pow(x, y); }       // GTR1 sampling only (sampling.rs:132): inlined, its
PT_DM_CALL double dev_log2(double x) { return detmath::log2(x); }                  // constant tables were k_shade's last spills

// Wave-uniform read of read-only scene data. The kernels also STORE to global memory (the path pool), so the compiler
// cannot prove that a plain load with a uniform address is never clobbered and issues a vector load for it: every lane
// fetches the same bytes into VGPRs. Read through the constant address space the same access becomes a scalar load
// (s_load_dwordxN): the operand stays in SGPRs, costs no vector-memory instruction and no vector registers. The scene
// tables are written by the host before the launch and never by a kernel. `p` MUST be wave-uniform.
template <class T> PT_DEV T ldu(const T* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized records only");
    typedef const __attribute__((address_space(4))) uint32_t* cptr;
    cptr q = (cptr)p;
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = q[i];
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}

// Shading frame: the shortest-arc quaternion taking n onto +z (vec3.rs:23-29). It is built
// once per (normal) and reused for every to_local/to_world of a bounce — the reference
// rebuilds it 4-7x per bounce with identical inputs, so reuse changes no bit.
struct Frame {
    double x, y, z, w;   // unit quaternion
};
PT_DEV Frame frame_to_z(V3 n) {
    if (n.z < -0.99999) return Frame{1.0, 0.0, 0.0, 0.0};
    double qx = n.y, qy = -n.x, qz = 0.0, qw = 1.0 + n.z;
    double len = sqrt((qx * qx) + (qy * qy) + (qz * qz) + (qw * qw));
    double r = 1.0 / len;
    return Frame{qx * r, qy * r, qz * r, qw * r};
}
PT_DEV V3 quat_mul(double qx, double qy, double qz, double qw, V3 rhs) {   // glam DQuat * DVec3
    V3 b{qx, qy, qz};
    double b2 = dot(b, b);
    return rhs * (qw * qw - b2) + b * (dot(rhs, b) * 2.0) + cross(b, rhs) * (qw * 2.0);
}
PT_DEV V3 to_local(const Frame& f, V3 w) { return quat_mul(f.x, f.y, f.z, f.w, w); }
PT_DEV V3 to_world(const Frame& f, V3 w) { return quat_mul(-f.x, -f.y, -f.z, f.w, w); }

// Rigid instance transform applied column-wise like glam's transform_point3/vector3.
PT_DEV V3 xform_vector(const double* c0, const double* c1, const double* c2, V3 v) {
    V3 r = ld3(c0) * v.x;
    r = ld3(c1) * v.y + r;
    r = ld3(c2) * v.z + r;
    return r;
}
PT_DEV V3 xform_point(const double* c0, const double* c1, const double* c2, const double* t, V3 p) {
    return ld3(t) + xform_vector(c0, c1, c2, p);
}

// ---- RNG: Philox4x32-10, key = (seed_lo, pixel), counter = (draw>>1, sample, seed_hi, 0) ----
// Stands in for rand 0.8.5's unseedable thread_rng; draw ORDER is the
// reference's. Two u64 per block: even draw = out[1]:out[0], odd draw = out[3]:out[2].
struct Rng {
    uint32_t seed_lo, seed_hi, pixel, sample, draw;
};
// One Philox4x32-10 block. Two things about its cost on this chip: (1) a 32-bit integer multiply is a quarter-rate
// instruction and the textbook round needs four of them (mulhi + mullo, twice) — written as two 64-bit products the
// compiler emits ONE v_mad_u64_u32 per product, halving the multiplies (the round-1 form was 22 % of k_shade's issue
// time: ~8 blocks per sample x 40 quarter-rate multiplies); (2) inlined at every draw site the block appeared 41 times
// in k_shade — as a real function (PT_PHILOX_CALL) it is there once.
struct PhiloxOut {
    uint32_t x, y, z, w;
};
#ifndef PT_PHILOX_INLINE
#define PT_PHILOX_CALL __device__ __noinline__
#else
#define PT_PHILOX_CALL __device__ __forceinline__
#endif
PT_PHILOX_CALL PhiloxOut philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += W0; k1 += W1;
    }
    return PhiloxOut{c0, c1, c2, c3};
}
PT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const PhiloxOut o = philox_block(c0, c1, c2, c3, k0, k1);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
}
PT_DEV uint64_t rng_u64(Rng& r) {
    uint32_t o[4];
    philox4x32_10(r.draw >> 1, r.sample, r.seed_hi, 0u, r.seed_lo, r.pixel, o);
    uint64_t v = (r.draw & 1u) ? (((uint64_t)o[3] << 32) | o[2]) : (((uint64_t)o[1] << 32) | o[0]);
    ++r.draw;
    return v;
}
// two consecutive draws; one Philox block when the first draw index is even
PT_DEV void rng_u64x2(Rng& r, uint64_t& a, uint64_t& b) {
    if ((r.draw & 1u) == 0u) {
        uint32_t o[4];
        philox4x32_10(r.draw >> 1, r.sample, r.seed_hi, 0u, r.seed_lo, r.pixel, o);
        a = ((uint64_t)o[1] << 32) | o[0];
        b = ((uint64_t)o[3] << 32) | o[2];
        r.draw += 2;
    } else {
        a = rng_u64(r);
        b = rng_u64(r);
    }
}
PT_DEV double u64_to_unit(uint64_t v) { return (double)(v >> 11) * (1.0 / 9007199254740992.0); }
PT_DEV double rng_f64(Rng& r) { return u64_to_unit(rng_u64(r)); }                  // rand Standard f64
PT_DEV double rng_range_inclusive(Rng& r, double scale) {                          // gen_range(0.0..=hi)
    return ((double)(rng_u64(r) >> 12) * (1.0 / 4503599627370496.0)) * scale;
}
PT_DEV uint32_t rng_index(Rng& r, uint32_t n) {                                    // gen_range(0..n) usize
    uint64_t range = n;
    uint64_t zone = (range << __clzll((long long)range)) - 1;
    for (int it = 0; it < 64; ++it) {
        uint64_t v = rng_u64(r);
        uint64_t hi = __umul64hi(v, range), lo = v * range;
        if (lo <= zone) return (uint32_t)hi;
    }
    return 0;
}

}  // namespace pt
