// hmath.h — the handful of glm operations the reference's host code uses,
// re-implemented with glm's own evaluation order so that matrices, inverses and
// camera vectors carry the same fp32 bits the reference would hand its kernel.
// glm is an un-vendored submodule of the reference (.gitmodules:7-9); the
// algorithms are the published ones of glm 1.0.x (detail/func_matrix.inl,
// ext/matrix_transform.inl, gtc/quaternion.inl).
#ifndef RTA_HMATH_H
#define RTA_HMATH_H

#include <math.h>
#include <string.h>

namespace rta {

struct vec2 {
  float x = 0, y = 0;
};

struct vec3 {
  float x = 0, y = 0, z = 0;
  vec3() = default;
  vec3(float s) : x(s), y(s), z(s) {}
  vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
  explicit vec3(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  void store(float *p) const { p[0] = x, p[1] = y, p[2] = z; }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// glm::min(x, y) = (y < x) ? y : x ; glm::max(x, y) = (x < y) ? y : x
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline vec3 min(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline vec3 max(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 x, vec3 y) {
  return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); } // v * inversesqrt(dot)

struct vec4 {
  float x = 0, y = 0, z = 0, w = 0;
  vec4() = default;
  vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
  vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
  float operator[](int i) const { return (&x)[i]; }
  float &operator[](int i) { return (&x)[i]; }
};
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

struct quat { // glm::quat(w, x, y, z)
  float w = 1, x = 0, y = 0, z = 0;
};

struct mat4 { // column-major: c[col][row]
  vec4 c[4];
  mat4() : mat4(1.f) {}
  explicit mat4(float d) {
    c[0] = {d, 0, 0, 0}, c[1] = {0, d, 0, 0}, c[2] = {0, 0, d, 0}, c[3] = {0, 0, 0, d};
  }
  explicit mat4(const float *p) { memcpy(static_cast<void *>(c), p, 64); }
  vec4 &operator[](int i) { return c[i]; }
  const vec4 &operator[](int i) const { return c[i]; }
  void store(float *p) const { memcpy(p, c, 64); }
};
static_assert(sizeof(mat4) == 64, "mat4 must be 16 packed floats");

// operator*(mat4, vec4): (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)
inline vec4 operator*(const mat4 &m, vec4 v) {
  vec4 add0 = m[0] * v.x + m[1] * v.y;
  vec4 add1 = m[2] * v.z + m[3] * v.w;
  return add0 + add1;
}
// operator*(mat4, mat4): column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline mat4 operator*(const mat4 &a, const mat4 &b) {
  mat4 r;
  for (int j = 0; j < 4; ++j) {
    vec4 t = a[0] * b[j].x;
    t = t + a[1] * b[j].y;
    t = t + a[2] * b[j].z;
    t = t + a[3] * b[j].w;
    r[j] = t;
  }
  return r;
}
inline mat4 operator*(const mat4 &m, float s) {
  mat4 r;
  for (int j = 0; j < 4; ++j)
    r[j] = m[j] * s;
  return r;
}

// glm::inverse(mat4): detail::compute_inverse<4,4,float> (cofactors, 1/det)
inline mat4 inverse(const mat4 &m) {
  float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
  float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
  float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
  float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
  float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
  float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
  float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
  float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
  float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
  float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  vec4 Fac0(Coef00, Coef00, Coef02, Coef03), Fac1(Coef04, Coef04, Coef06, Coef07);
  vec4 Fac2(Coef08, Coef08, Coef10, Coef11), Fac3(Coef12, Coef12, Coef14, Coef15);
  vec4 Fac4(Coef16, Coef16, Coef18, Coef19), Fac5(Coef20, Coef20, Coef22, Coef23);
  vec4 Vec0(m[1][0], m[0][0], m[0][0], m[0][0]), Vec1(m[1][1], m[0][1], m[0][1], m[0][1]);
  vec4 Vec2(m[1][2], m[0][2], m[0][2], m[0][2]), Vec3(m[1][3], m[0][3], m[0][3], m[0][3]);
  vec4 Inv0 = Vec1 * Fac0 - Vec2 * Fac1 + Vec3 * Fac2;
  vec4 Inv1 = Vec0 * Fac0 - Vec2 * Fac3 + Vec3 * Fac4;
  vec4 Inv2 = Vec0 * Fac1 - Vec1 * Fac3 + Vec3 * Fac5;
  vec4 Inv3 = Vec0 * Fac2 - Vec1 * Fac4 + Vec2 * Fac5;
  vec4 SignA(+1, -1, +1, -1), SignB(-1, +1, -1, +1);
  mat4 Inverse;
  Inverse[0] = Inv0 * SignA, Inverse[1] = Inv1 * SignB, Inverse[2] = Inv2 * SignA,
  Inverse[3] = Inv3 * SignB;
  vec4 Row0(Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]);
  vec4 Dot0 = m[0] * Row0;
  float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
  float OneOverDeterminant = 1.0f / Dot1;
  return Inverse * OneOverDeterminant;
}

// glm::translate(m, v)
inline mat4 translate(const mat4 &m, vec3 v) {
  mat4 r = m;
  r[3] = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
  return r;
}
// glm::scale(m, v)
inline mat4 scale(const mat4 &m, vec3 v) {
  mat4 r;
  r[0] = m[0] * v.x, r[1] = m[1] * v.y, r[2] = m[2] * v.z, r[3] = m[3];
  return r;
}
// glm::angleAxis(angle, axis)
inline quat angle_axis(float angle, vec3 axis) {
  float s = sinf(angle * 0.5f);
  quat q;
  q.w = cosf(angle * 0.5f);
  q.x = axis.x * s, q.y = axis.y * s, q.z = axis.z * s;
  return q;
}
// glm::toMat4(q) = mat4(mat3_cast(q))
inline mat4 to_mat4(quat q) {
  float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
  float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
  float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
  mat4 r(1.f);
  r[0][0] = 1.0f - 2.0f * (qyy + qzz);
  r[0][1] = 2.0f * (qxy + qwz);
  r[0][2] = 2.0f * (qxz - qwy);
  r[1][0] = 2.0f * (qxy - qwz);
  r[1][1] = 1.0f - 2.0f * (qxx + qzz);
  r[1][2] = 2.0f * (qyz + qwx);
  r[2][0] = 2.0f * (qxz + qwy);
  r[2][1] = 2.0f * (qyz - qwx);
  r[2][2] = 1.0f - 2.0f * (qxx + qyy);
  return r;
}
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// Transform::get_mat4, src/Transform.hpp:15-18
inline mat4 trs(vec3 position, quat rotation, vec3 scl) {
  return translate(mat4(1.f), position) * to_mat4(rotation) * scale(mat4(1.f), scl);
}

constexpr float kInfinity = __builtin_huge_valf();

} // namespace rta
#endif
