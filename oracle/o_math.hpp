// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product path).
// CPU restatement of the reference's math layer (math/src/*.rs) on top of the
// arithmetic of its glam 0.30.3 dependency (not vendored under /root/reference:
// restated from glam's published scalar algorithms; parity unpinned — the
// reference holds no known-answer vectors for this layer).
//
// Compile with -ffp-contract=off: rustc never contracts a*b+c into an FMA.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <limits>

namespace oracle {
constexpr float PI_F = 3.14159265358979323846f;

struct V2 { float x, y; };

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glam Vec3::dot: (x*x' + y*y') + z*z'
static inline float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam Vec3::cross
static inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
static inline float length_squared(V3 a) { return dot(a, a); }
static inline float length(V3 a) { return std::sqrt(dot(a, a)); }
// glam Vec3::normalize = self * length_recip()  (math/src/vector.rs:48, normal.rs:93-100)
static inline V3 normalize(V3 a) { return a * (1.0f / length(a)); }
static inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
static inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
static inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }
static inline float max_element(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
// glam Vec3::max_position: index of the first maximum (strict '>' updates)
static inline int max_position(V3 a) {
    float m = a.x; int idx = 0;
    if (a.y > m) { m = a.y; idx = 1; }
    if (a.z > m) { idx = 2; }
    return idx;
}
static inline bool is_nan(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
// Rust f32::signum: +1 for +0/positive, -1 for -0/negative, NaN for NaN
static inline float signum(float x) { return std::isnan(x) ? x : std::copysign(1.0f, x); }
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

struct V4 { float x, y, z, w; };
static inline V4 operator*(V4 a, V4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline V4 operator-(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline V4 operator+(V4 a, V4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline V4 operator*(V4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// Column-major 4x4, glam::Mat4 semantics.
struct M4 {
    V4 c[4];
    static M4 identity() { return M4{{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
    static M4 from_cols16(const float* m) {
        M4 r;
        for (int i = 0; i < 4; ++i) r.c[i] = {m[4 * i + 0], m[4 * i + 1], m[4 * i + 2], m[4 * i + 3]};
        return r;
    }
    static M4 from_translation(V3 t) {
        M4 r = identity(); r.c[3] = {t.x, t.y, t.z, 1.0f}; return r;
    }
    static M4 from_axes(V3 x, V3 y, V3 z) {
        return M4{{{x.x, x.y, x.z, 0}, {y.x, y.y, y.z, 0}, {z.x, z.y, z.z, 0}, {0, 0, 0, 1}}};
    }
};
static inline V4 mul(const M4& m, V4 v) {
    // glam Mat4::mul_vec4: x_axis*v.x + y_axis*v.y + z_axis*v.z + w_axis*v.w
    V4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    r = r + m.c[3] * v.w;
    return r;
}
static inline M4 operator*(const M4& a, const M4& b) {
    M4 r;
    for (int i = 0; i < 4; ++i) r.c[i] = mul(a, b.c[i]);
    return r;
}
// glam Mat4::transform_point3 (no perspective divide)
static inline V3 transform_point3(const M4& m, V3 p) {
    V4 r = m.c[0] * p.x;
    r = r + m.c[1] * p.y;
    r = r + m.c[2] * p.z;
    r = r + m.c[3];
    return {r.x, r.y, r.z};
}
// glam Mat4::transform_vector3
static inline V3 transform_vector3(const M4& m, V3 v) {
    V4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    return {r.x, r.y, r.z};
}
static inline M4 transpose(const M4& m) {
    return M4{{{m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x},
               {m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y},
               {m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z},
               {m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w}}};
}
// glam Mat4::inverse (cofactor form inherited from GLM).
static inline M4 inverse(const M4& s) {
    float m00 = s.c[0].x, m01 = s.c[0].y, m02 = s.c[0].z, m03 = s.c[0].w;
    float m10 = s.c[1].x, m11 = s.c[1].y, m12 = s.c[1].z, m13 = s.c[1].w;
    float m20 = s.c[2].x, m21 = s.c[2].y, m22 = s.c[2].z, m23 = s.c[2].w;
    float m30 = s.c[3].x, m31 = s.c[3].y, m32 = s.c[3].z, m33 = s.c[3].w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    V4 fac0{coef00, coef00, coef02, coef03}, fac1{coef04, coef04, coef06, coef07};
    V4 fac2{coef08, coef08, coef10, coef11}, fac3{coef12, coef12, coef14, coef15};
    V4 fac4{coef16, coef16, coef18, coef19}, fac5{coef20, coef20, coef22, coef23};
    V4 vec0{m10, m00, m00, m00}, vec1{m11, m01, m01, m01}, vec2{m12, m02, m02, m02}, vec3_{m13, m03, m03, m03};
    V4 inv0 = (vec1 * fac0 - vec2 * fac1) + vec3_ * fac2;
    V4 inv1 = (vec0 * fac0 - vec2 * fac3) + vec3_ * fac4;
    V4 inv2 = (vec0 * fac1 - vec1 * fac3) + vec3_ * fac5;
    V4 inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5;
    V4 sa{1, -1, 1, -1}, sb{-1, 1, -1, 1};
    M4 inv{{inv0 * sa, inv1 * sb, inv2 * sa, inv3 * sb}};
    V4 col0{inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x};
    V4 d0 = s.c[0] * col0;
    float det = ((d0.x + d0.y) + d0.z) + d0.w;
    float rcp = 1.0f / det;
    return M4{{inv.c[0] * rcp, inv.c[1] * rcp, inv.c[2] * rcp, inv.c[3] * rcp}};
}

// Transform * Normal: (M^-1)^T * n, then Normal::from renormalises
// (math/src/transform.rs:45-51, math/src/normal.rs:93-100).
static inline V3 transform_normal(const M4& m, V3 n) {
    M4 it = transpose(inverse(m));
    return normalize(transform_vector3(it, n));
}

struct Ray { V3 o, d; };
// Ray::move_forward (math/src/ray.rs:20-23)
static inline Ray move_forward(const Ray& r, float dist) { return Ray{r.o + r.d * dist, r.d}; }
// Transform * Ray (math/src/transform.rs:53-60): direction is NOT renormalised
static inline Ray transform_ray(const M4& m, const Ray& r) {
    return Ray{transform_point3(m, r.o), transform_vector3(m, r.d)};
}

struct Bounds {
    V3 mn, mx;
    V3 center() const { return (mn + mx) * 0.5f; }                       // bounds.rs:58-61
    float area() const {                                                  // bounds.rs:64-67
        V3 d = mx - mn; return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    Bounds merge(const Bounds& o) const { return {vmin(mn, o.mn), vmax(mx, o.mx)}; }
};
// Bounds::intersect slab test (math/src/bounds.rs:27-55). Returns hit flag.
static inline bool bounds_intersect(const Bounds& b, const Ray& ray, float t_max, V3 inv_dir) {
    float t0 = 0.0f, t1 = t_max;
    for (int i = 0; i < 3; ++i) {
        float t_near = (b.mn[i] - ray.o[i]) * inv_dir[i];
        float t_far = (b.mx[i] - ray.o[i]) * inv_dir[i];
        if (t_near > t_far) std::swap(t_near, t_far);
        t0 = t_near > t0 ? t_near : t0;
        t1 = t_far < t1 ? t_far : t1;
        if (t0 > t1) return false;
    }
    return true;
}
// Transform * Bounds via the 8 corners (math/src/transform.rs:61-74)
static inline Bounds transform_bounds(const M4& m, const Bounds& b) {
    float inf = std::numeric_limits<float>::infinity();
    V3 mn{inf, inf, inf}, mx{-inf, -inf, -inf};
    for (int k = 0; k < 8; ++k) {
        V3 p{(k & 1) ? b.mx.x : b.mn.x, (k & 2) ? b.mx.y : b.mn.y, (k & 4) ? b.mx.z : b.mn.z};
        V3 q = transform_point3(m, p);
        mn = vmin(mn, q); mx = vmax(mx, q);
    }
    return {mn, mx};
}

struct TriHit { float t; V3 p; V3 n; float b[3]; };

// math::intersect_triangle (math/src/ray.rs:44-182): PBRT-style watertight test.
static inline bool intersect_triangle(const Ray& ray, float t_max, const V3 ps[3], TriHit* out) {
    if (length_squared(cross(ps[1] - ps[0], ps[2] - ps[0])) == 0.0f) return false;
    V3 p0o = ps[0] - ray.o, p1o = ps[1] - ray.o, p2o = ps[2] - ray.o;
    V3 d = ray.d;
    int kz = max_position(vabs(d));
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    V3 dd{d[kx], d[ky], d[kz]};
    V3 p0{p0o[kx], p0o[ky], p0o[kz]}, p1{p1o[kx], p1o[ky], p1o[kz]}, p2{p2o[kx], p2o[ky], p2o[kz]};
    float sx = -dd.x / dd.z, sy = -dd.y / dd.z, sz = 1.0f / dd.z;
    p0.x += sx * p0.z; p0.y += sy * p0.z;
    p1.x += sx * p1.z; p1.y += sy * p1.z;
    p2.x += sx * p2.z; p2.y += sy * p2.z;
    float e0 = p2.x * p1.y - p2.y * p1.x;
    float e1 = p0.x * p2.y - p0.y * p2.x;
    float e2 = p1.x * p0.y - p1.y * p0.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        e0 = (float)((double)p2.x * (double)p1.y - (double)p2.y * (double)p1.x);
        e1 = (float)((double)p0.x * (double)p2.y - (double)p0.y * (double)p2.x);
        e2 = (float)((double)p1.x * (double)p0.y - (double)p1.y * (double)p0.x);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0.z *= sz; p1.z *= sz; p2.z *= sz;
    float t_scaled = e0 * p0.z + e1 * p1.z + e2 * p2.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) return false;
    else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det)) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t_hit = t_scaled * inv_det;
    // conservative t > 0 check (ray.rs:137-158)
    const float EPS = 1.1920929e-7f * 0.5f;
    auto gamma = [&](int n) { return ((float)n * EPS) / (1.0f - (float)n * EPS); };
    float max_zt = max_element(vabs(V3{p0.z, p1.z, p2.z}));
    float delta_z = gamma(3) * max_zt;
    float max_xt = max_element(vabs(V3{p0.x, p1.x, p2.x}));
    float max_yt = max_element(vabs(V3{p0.y, p1.y, p2.y}));
    float delta_x = gamma(5) * max_xt, delta_y = gamma(5) * max_yt;
    float delta_e = 2.0f * (gamma(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max_element(vabs(V3{e0, e1, e2}));
    float delta_t = 3.0f * (gamma(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::fabs(inv_det);
    if (t_hit < delta_t) return false;
    if (out) {
        out->t = t_hit;
        out->p = ps[0] * b0 + ps[1] * b1 + ps[2] * b2;
        // Normal::from(cross.normalize()) renormalises once more (ray.rs:167-174, normal.rs:93-100)
        out->n = normalize(normalize(cross(ps[1] - ps[0], ps[2] - ps[0])));
        out->b[0] = b0; out->b[1] = b1; out->b[2] = b2;
    }
    return true;
}

// Normal::orthogonalize_vector (math/src/normal.rs:45-51)
static inline V3 orthogonalize_vector(V3 n, V3 v) {
    float pm = dot(n, v);
    return normalize(v - n * pm);
}
// Normal::generate_tangent (math/src/normal.rs:55-65)
static inline V3 generate_tangent(V3 n) {
    V3 cand = std::fabs(n.x) > 0.999f ? V3{0, 1, 0} : V3{1, 0, 0};
    return orthogonalize_vector(n, cand);
}

// Transform::from_shading_normal_tangent (math/src/transform.rs:186-203):
// Render -> VertexNormalTangent = inverse([T B N]).
static inline M4 from_shading_normal_tangent(V3 shading_normal, V3 tangent) {
    V3 n = normalize(shading_normal);
    V3 b = normalize(cross(normalize(n), tangent));
    V3 t = normalize(cross(b, n));
    return inverse(M4::from_axes(t, b, n));
}
// Transform::from_normal_map (math/src/transform.rs:216-244)
static inline M4 from_normal_map(V3 nm) {
    V3 z = normalize(nm);
    V3 cx = std::fabs(dot(z, V3{1, 0, 0})) < 0.9f ? V3{1, 0, 0} : V3{0, 1, 0};
    V3 x = normalize(cx - dot(z, cx) * z);
    V3 y = normalize(cross(z, x));
    return inverse(M4::from_axes(x, y, z));
}

}  // namespace oracle
