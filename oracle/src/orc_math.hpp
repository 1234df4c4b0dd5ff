// ORACLE (test infrastructure only — never linked into the product library).
//
// Minimal f32 vector/matrix arithmetic restating the operation ORDER of the
// third-party `glam 0.30.10` crate (engine/Cargo.lock:1072-1075) as wrapped by
// impact_math (engine/crates/impact_math/src/vector.rs:36-72,477-515,1139-1141;
// point.rs:285,426-460; matrix.rs:643-682). glam is not vendored under
// /root/reference, so the conventions below come from its published SSE2
// implementation and are "parity unpinned" at the bit level:
//   dot(a,b)      = (ax*bx + ay*by) + az*bz
//   length(v)     = sqrt(dot(v,v))
//   normalize(v)  = v / length(v)           (SSE2 Vec3A/Vec4/Quat: _mm_div_ps)
//   v / s         = v * (1/s)               (impact_math Div impls: a.mul(b.recip()))
//   M4 * point    = ((c0*x + c1*y) + c2*z) + c3
//   M3 * v        = (c0*x + c1*y) + c2*z
// Everything is compiled with -ffp-contract=off so no FMA is ever formed.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct V3 {
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3s(float s) { return V3{s, s, s}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 cmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// impact_math `Div<f32>`: multiply by the reciprocal (vector.rs:679, point.rs:444)
inline V3 div_recip(V3 a, float s) {
    float r = 1.0f / s;
    return {a.x * r, a.y * r, a.z * r};
}
inline V3 div_elem(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// glam SSE2 Vec3A::normalize: self / sqrt(dot)
inline V3 normalize(V3 a) { return div_elem(a, length(a)); }
inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline float fmin_rs(float a, float b) { return (b < a) ? b : a; }  // f32::min (no NaN here)
inline float fmax_rs(float a, float b) { return (b > a) ? b : a; }
inline V3 vmin(V3 a, V3 b) { return {fmin_rs(a.x, b.x), fmin_rs(a.y, b.y), fmin_rs(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {fmax_rs(a.x, b.x), fmax_rs(a.y, b.y), fmax_rs(a.z, b.z)}; }
inline float max_component(V3 a) { return fmax_rs(fmax_rs(a.x, a.y), a.z); }
inline bool sign_neg(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return (u >> 31) != 0;
}
inline uint32_t neg_mask(V3 a) {
    return (sign_neg(a.x) ? 1u : 0u) | (sign_neg(a.y) ? 2u : 0u) | (sign_neg(a.z) ? 4u : 0u);
}

struct Quat {
    float x, y, z, w;
};
inline Quat conj(Quat q) { return {-q.x, -q.y, -q.z, q.w}; }
inline Quat qmul(Quat a, Quat b) {
    // Hamilton product (scalar form)
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
            a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat qnormalize(Quat q) {
    float d = ((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w;  // dot4 order: ((x+y)+z)+w? see note
    float l = std::sqrt(d);
    return {q.x / l, q.y / l, q.z / l, q.w / l};
}

// Column-major 3x3
struct M3 {
    V3 c0, c1, c2;
};
inline V3 mul(const M3& m, V3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
inline M3 transpose(const M3& m) {
    return {{m.c0.x, m.c1.x, m.c2.x}, {m.c0.y, m.c1.y, m.c2.y}, {m.c0.z, m.c1.z, m.c2.z}};
}
inline M3 mul(const M3& a, const M3& b) { return {mul(a, b.c0), mul(a, b.c1), mul(a, b.c2)}; }
inline M3 operator*(const M3& a, float s) { return {a.c0 * s, a.c1 * s, a.c2 * s}; }
inline M3 operator+(const M3& a, const M3& b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
inline M3 m3_identity() { return {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; }
inline M3 m3_from_diag(V3 d) { return {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}}; }
// glam Mat3::from_quat
inline M3 m3_from_quat(Quat q) {
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2;
    float yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
    float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
    return {{1.0f - (yy + zz), xy + wz, xz - wy},
            {xy - wz, 1.0f - (xx + zz), yz + wx},
            {xz + wy, yz - wx, 1.0f - (xx + yy)}};
}
// glam Mat3A::inverse (cofactor / determinant form)
inline M3 inverse(const M3& m) {
    V3 tmp0 = cross(m.c1, m.c2);
    V3 tmp1 = cross(m.c2, m.c0);
    V3 tmp2 = cross(m.c0, m.c1);
    float det = dot(m.c2, tmp2);
    V3 inv_det = v3s(1.0f / det);
    M3 t{cmul(tmp0, inv_det), cmul(tmp1, inv_det), cmul(tmp2, inv_det)};
    return transpose(t);
}

// Column-major 4x4 homogeneous transform; only affine use.
struct M4 {
    float c[4][4];  // c[col][row]
};
inline M4 m4_identity() {
    M4 m{};
    for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f;
    return m;
}
inline V3 col3(const M4& m, int col) { return {m.c[col][0], m.c[col][1], m.c[col][2]}; }
// glam Mat4::transform_point3a: x_axis*x, + y_axis*y, + z_axis*z, + w_axis
inline V3 transform_point(const M4& m, V3 p) {
    V3 r = col3(m, 0) * p.x;
    r = col3(m, 1) * p.y + r;
    r = col3(m, 2) * p.z + r;
    r = col3(m, 3) + r;
    return r;
}
inline M3 linear_part(const M4& m) { return {col3(m, 0), col3(m, 1), col3(m, 2)}; }
inline M4 m4_from_quat(Quat q) {
    M3 r = m3_from_quat(q);
    M4 m = m4_identity();
    V3 cs[3] = {r.c0, r.c1, r.c2};
    for (int j = 0; j < 3; ++j) {
        m.c[j][0] = cs[j].x;
        m.c[j][1] = cs[j].y;
        m.c[j][2] = cs[j].z;
    }
    return m;
}
// glam Mat4 * Mat4: each result column = a.mul_vec4(b.col) = ((a0*x + a1*y) + a2*z) + a3*w
inline M4 mul(const M4& a, const M4& b) {
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) {
            float s = a.c[0][i] * b.c[j][0];
            s = s + a.c[1][i] * b.c[j][1];
            s = s + a.c[2][i] * b.c[j][2];
            s = s + a.c[3][i] * b.c[j][3];
            r.c[j][i] = s;
        }
    return r;
}

struct AABB {
    V3 lo, hi;
};
// impact_geometry/src/axis_aligned_box.rs:126-139,253-268,332-366
inline V3 center(const AABB& b) { return 0.5f * (b.lo + b.hi); }
inline V3 extents(const AABB& b) { return b.hi - b.lo; }
inline V3 half_extents(const AABB& b) { return 0.5f * extents(b); }
inline AABB expanded(const AABB& b, float margin) { return {b.lo - v3s(margin), b.hi + v3s(margin)}; }
inline bool contains_box(const AABB& self, const AABB& other) {
    return (neg_mask(other.lo - self.lo) | neg_mask(self.hi - other.hi)) == 0;
}
inline bool box_lies_outside(const AABB& self, const AABB& other) {
    return (neg_mask(other.hi - self.lo) | neg_mask(self.hi - other.lo)) != 0;
}
inline AABB aabb_of_transformed(const AABB& b, const M4& m) {
    V3 c = transform_point(m, center(b));
    M3 l = linear_part(m);
    M3 a{vabs(l.c0), vabs(l.c1), vabs(l.c2)};
    V3 h = mul(a, half_extents(b));
    return {c - h, c + h};
}

}  // namespace orc
