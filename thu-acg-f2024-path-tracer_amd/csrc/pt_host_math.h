// Host-side f64 helpers for the scene builder: the derived constants the reference computes
// in its constructors (Quad::new quad.rs:17-36, Instance::new instance.rs:20-30,
// Camera::init camera.rs:51-77, PrincipledBSDF lobe weights principled.rs:75-100) are
// computed here once, in the reference's operation order, and shipped to the GPU as data.
// Built with -ffp-contract=off like the kernels.
#pragma once
#include <cmath>
#include <cstdint>

namespace pt {
namespace host {

constexpr double PI = 3.14159265358979323846264338327950288;

struct D3 {
    double x, y, z;
};
inline D3 d3(const double* p) { return D3{p[0], p[1], p[2]}; }
inline void st3(double* p, D3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator-(D3 a) { return {-a.x, -a.y, -a.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline D3 operator/(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(D3 a, D3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline double length(D3 a) { return std::sqrt(dot(a, a)); }
inline D3 normalize(D3 a) { return a * (1.0 / length(a)); }
inline D3 vmin(D3 a, D3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline D3 vmax(D3 a, D3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

inline D3 xform_vector(D3 c0, D3 c1, D3 c2, D3 v) {
    D3 r = c0 * v.x;
    r = c1 * v.y + r;
    r = c2 * v.z + r;
    return r;
}
inline D3 xform_point(D3 c0, D3 c1, D3 c2, D3 t, D3 p) { return t + xform_vector(c0, c1, c2, p); }

struct Box {
    D3 lo{INFINITY, INFINITY, INFINITY}, hi{-INFINITY, -INFINITY, -INFINITY};
    void grow(D3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Box& b) { lo = vmin(lo, b.lo); hi = vmax(hi, b.hi); }
    D3 centroid() const { return (lo + hi) * 0.5; }
    double half_area() const {
        D3 e = hi - lo;
        return e.x * e.y + e.x * e.z + e.y * e.z;
    }
    bool valid() const { return lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z; }
};

}  // namespace host
}  // namespace pt
