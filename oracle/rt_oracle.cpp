// rt_oracle.cpp — CPU oracle: a plain restatement of the reference's render
// path.  TEST INFRASTRUCTURE ONLY (see rt_oracle.h): the product never links it.
//
// PARITY UNPINNED: the reference holds no golden vectors for this path and
// cannot be built here; every function cites the reference
// lines it restates, and tests/test_oracle_*.py pin it with hand-derived
// known answers.
//
// Arithmetic rules (they are what makes CPU<->GPU bit-exactness possible):
//   * fp32 only, no FMA contraction (-ffp-contract=off), no fast-math;
//   * dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z, left to right;
//   * min(x,y) = y < x ? y : x and max(x,y) = x < y ? y : x (GLSL.std.450
//     FMin/FMax wording; matters only when a NaN is involved);
//   * normalize(v) = v * (1 / sqrt(dot(v,v)))  (glm::normalize);
//   * mat4 * vec4 = (c0*x + c1*y) + (c2*z + c3*w)  (glm operator*);
//   * sin/cos through orc_sincos (Cody-Waite + minimax polynomials) because the
//     reference's are driver-defined; pow(x,5) = ((x*x)*(x*x))*x;
//   * lerp(a,b,t) = a*(1-t) + b*t (GLSL.std.450 FMix);
//   * reflect(i,n) = i - (2*dot(n,i))*n.
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <thread>
#include <vector>

namespace {

struct V3 {
  float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z,
            a.x * b.y - a.y * b.x);
}
inline float fmin_(float x, float y) { return y < x ? y : x; }
inline float fmax_(float x, float y) { return x < y ? y : x; }
inline V3 vmin(V3 a, V3 b) {
  return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z));
}
inline V3 vmax(V3 a, V3 b) {
  return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z));
}
inline V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
inline V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }
inline void st3(float *p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}
inline float comp(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// glm operator*(mat4, vec4): (c0*x + c1*y) + (c2*z + c3*w); xyz only.
inline V3 mul_m4_v(const float *m, V3 v, float w) {
  V3 c0 = ld3(m + 0), c1 = ld3(m + 4), c2 = ld3(m + 8), c3 = ld3(m + 12);
  return (c0 * v.x + c1 * v.y) + (c2 * v.z + c3 * w);
}
// mul(transpose(M), (v,0)).xyz : column c of transpose(M) is row c of M.
inline V3 mul_m4t_v0(const float *m, V3 v) {
  V3 c0 = v3(m[0], m[4], m[8]), c1 = v3(m[1], m[5], m[9]),
     c2 = v3(m[2], m[6], m[10]), c3 = v3(m[3], m[7], m[11]);
  return (c0 * v.x + c1 * v.y) + (c2 * v.z + c3 * 0.0f);
}

const float kInfinity = INFINITY;
const float kEpsilon = 1.192092896e-07f; // Ray.slang:4
const float kMiss = 1e30f;               // BVHNode.slang:25

} // namespace

// ---------------------------------------------------------------------------
// glm restatements
// ---------------------------------------------------------------------------
extern "C" void orc_mat4_identity(float *m) {
  for (int i = 0; i < 16; ++i)
    m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
}

// glm operator*(mat4, mat4): Result[j] = ((A0*B[j][0] + A1*B[j][1]) + A2*B[j][2]) + A3*B[j][3]
extern "C" void orc_mat4_mul(const float *a, const float *b, float *out) {
  float r[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float t = a[0 + i] * b[j * 4 + 0];
      t = t + a[4 + i] * b[j * 4 + 1];
      t = t + a[8 + i] * b[j * 4 + 2];
      t = t + a[12 + i] * b[j * 4 + 3];
      r[j * 4 + i] = t;
    }
  memcpy(out, r, sizeof r);
}

extern "C" void orc_mat4_mul_vec4(const float *m, const float *v, float *out) {
  float r[4];
  for (int i = 0; i < 4; ++i)
    r[i] = (m[0 + i] * v[0] + m[4 + i] * v[1]) +
           (m[8 + i] * v[2] + m[12 + i] * v[3]);
  memcpy(out, r, sizeof r);
}

// glm::inverse(mat4) — detail::compute_inverse<4,4> (cofactor form).
// m(c,r) below is glm's m[c][r].
extern "C" void orc_mat4_inverse(const float *mm, float *out) {
#define M(c, r) mm[(c) * 4 + (r)]
  float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
  float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
  float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
  float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
  float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
  float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
  float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
  float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
  float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
  float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
  float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
  float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
  float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

  float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
  float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
  float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
  float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
  float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
  float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
  float Vec0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
  float Vec1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
  float Vec2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
  float Vec3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  static const float SignA[4] = {+1, -1, +1, -1};
  static const float SignB[4] = {-1, +1, -1, +1};
  float Inv[16];
  for (int i = 0; i < 4; ++i) {
    float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
    float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
    float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
    float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
    Inv[0 * 4 + i] = Inv0 * SignA[i];
    Inv[1 * 4 + i] = Inv1 * SignB[i];
    Inv[2 * 4 + i] = Inv2 * SignA[i];
    Inv[3 * 4 + i] = Inv3 * SignB[i];
  }
  // Row0 = (Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0])
  float Dot0[4] = {M(0, 0) * Inv[0], M(0, 1) * Inv[4], M(0, 2) * Inv[8],
                   M(0, 3) * Inv[12]};
  float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
  float OneOverDeterminant = 1.0f / Dot1;
  for (int i = 0; i < 16; ++i)
    out[i] = Inv[i] * OneOverDeterminant;
#undef M
}

// glm::translate(m, v): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
extern "C" void orc_mat4_translate(const float *m, const float *v, float *out) {
  float r[16];
  memcpy(r, m, sizeof r);
  for (int i = 0; i < 4; ++i)
    r[12 + i] = ((m[0 + i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) +
                m[12 + i];
  memcpy(out, r, sizeof r);
}

// glm::scale(m, v): Result[i] = m[i] * v[i], Result[3] = m[3]
extern "C" void orc_mat4_scale(const float *m, const float *v, float *out) {
  float r[16];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < 4; ++i)
      r[c * 4 + i] = m[c * 4 + i] * v[c];
  for (int i = 0; i < 4; ++i)
    r[12 + i] = m[12 + i];
  memcpy(out, r, sizeof r);
}

// glm::angleAxis(angle, v): quat(cos(angle/2), v * sin(angle/2)); w first.
extern "C" void orc_quat_angle_axis(float angle, const float *axis, float *q) {
  float a = angle;
  float s = sinf(a * 0.5f);
  q[0] = cosf(a * 0.5f);
  q[1] = axis[0] * s;
  q[2] = axis[1] * s;
  q[3] = axis[2] * s;
}

// glm::toMat4(quat) = mat4(mat3_cast(q))
extern "C" void orc_quat_to_mat4(const float *q, float *out) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float qxx = x * x, qyy = y * y, qzz = z * z;
  float qxz = x * z, qxy = x * y, qyz = y * z;
  float qwx = w * x, qwy = w * y, qwz = w * z;
  orc_mat4_identity(out);
  out[0] = 1.0f - 2.0f * (qyy + qzz);
  out[1] = 2.0f * (qxy + qwz);
  out[2] = 2.0f * (qxz - qwy);
  out[4] = 2.0f * (qxy - qwz);
  out[5] = 1.0f - 2.0f * (qxx + qzz);
  out[6] = 2.0f * (qyz + qwx);
  out[8] = 2.0f * (qxz + qwy);
  out[9] = 2.0f * (qyz - qwx);
  out[10] = 1.0f - 2.0f * (qxx + qyy);
}

// Transform::get_mat4, src/Transform.hpp:15-18:
//   translate(I, position) * toMat4(rotation) * scale(I, scale)
extern "C" void orc_trs(const float *pos, const float *q, const float *scale,
                        float *out) {
  float I[16], T[16], R[16], S[16], TR[16];
  orc_mat4_identity(I);
  orc_mat4_translate(I, pos, T);
  orc_quat_to_mat4(q, R);
  orc_mat4_scale(I, scale, S);
  orc_mat4_mul(T, R, TR);
  orc_mat4_mul(TR, S, out);
}

// ---------------------------------------------------------------------------
// Host: camera uniforms — Renderer::render, src/Renderer.cpp:506-537
// (degrees_to_radians: src/Core/Defines.hpp:86-89)
// ---------------------------------------------------------------------------
extern "C" void orc_camera_uniforms(const float *position, const float *look_at,
                                    const float *v_up, float fov_deg,
                                    uint32_t width, uint32_t height,
                                    rt_camera_uniforms *out) {
  V3 pos = ld3(position), at = ld3(look_at), up = ld3(v_up);
  V3 d = pos - at;
  float focal_length = sqrtf(dot(d, d)); // glm::length
  // degrees_to_radians: degrees * pi / 180 (Core helper, RTIOW form)
  float theta = fov_deg * 3.1415926535897932385f / 180.0f;
  float h = tanf(theta / 2.f);
  float viewport_height = 2.f * h * focal_length;
  float viewport_width =
      viewport_height * (static_cast<float>(width) / height);
  V3 w = normalize(pos - at);
  V3 u = normalize(cross(up, w));
  V3 v = cross(w, u);
  V3 viewport_u = viewport_width * u;
  V3 viewport_v = viewport_height * neg(v);
  V3 pixel_delta_u = viewport_u / static_cast<float>(width);
  V3 pixel_delta_v = viewport_v / static_cast<float>(height);
  V3 viewport_upper_left =
      pos - (focal_length * w) - 0.5f * (viewport_u + viewport_v);
  V3 pixel00 = viewport_upper_left + 0.5f * (pixel_delta_u + pixel_delta_v);
  memset(out, 0, sizeof *out);
  st3(out->pixel00_loc, pixel00);
  out->pixel00_loc[3] = 1.f;
  st3(out->pixel_delta_u, pixel_delta_u);
  out->pixel_delta_u[3] = 1.f;
  st3(out->pixel_delta_v, pixel_delta_v);
  out->pixel_delta_v[3] = 1.f;
  st3(out->camera_center, pos);
  out->camera_center[3] = 1.f;
}

// ---------------------------------------------------------------------------
// Host: Renderer::add_blas de-indexing — src/Renderer.cpp:773-787
// ---------------------------------------------------------------------------
extern "C" void orc_deindex(const float *positions, const float *normals,
                            const float *uvs, const uint32_t *indices,
                            uint32_t index_count, uint32_t tri_id_index,
                            rt_triangle_geom *geom, rt_triangle_shading *shading,
                            float *centroids, uint32_t *tri_ids) {
  uint32_t tri_index = tri_id_index;
  uint32_t index = 0;
  for (uint32_t i = 0; i + 2 < index_count; i += 3) {
    uint32_t ia = indices[i], ib = indices[i + 1], ic = indices[i + 2];
    rt_triangle_geom &g = geom[tri_index];
    st3(g.v0, ld3(positions + 3 * ia));
    g.v0[3] = 1.f;
    st3(g.v1, ld3(positions + 3 * ib));
    g.v1[3] = 1.f;
    st3(g.v2, ld3(positions + 3 * ic));
    g.v2[3] = 1.f;
    rt_triangle_shading &s = shading[tri_index];
    st3(s.n0, ld3(normals + 3 * ia));
    s.n0[3] = 1.f;
    st3(s.n1, ld3(normals + 3 * ib));
    s.n1[3] = 1.f;
    st3(s.n2, ld3(normals + 3 * ic));
    s.n2[3] = 1.f;
    s.uv0[0] = uvs[2 * ia], s.uv0[1] = uvs[2 * ia + 1];
    s.uv1[0] = uvs[2 * ib], s.uv1[1] = uvs[2 * ib + 1];
    s.uv2[0] = uvs[2 * ic], s.uv2[1] = uvs[2 * ic + 1];
    s.padding[0] = s.padding[1] = 0.f;
    V3 c = ((ld3(positions + 3 * ia) + ld3(positions + 3 * ib)) +
            ld3(positions + 3 * ic)) *
           0.3333f; // sic, Renderer.cpp:783
    st3(centroids + 3 * tri_index, c);
    tri_ids[tri_id_index + index] = tri_id_index + index;
    ++tri_index;
    ++index;
  }
}

// ---------------------------------------------------------------------------
// Host: BLAS builder — src/BVHNode.cpp:5-161, src/AABB.hpp:7-27
// ---------------------------------------------------------------------------
namespace {
const uint32_t BIN_COUNT = 200; // BVHNode.cpp:5

struct AABB { // AABB.hpp:7-27
  V3 mn, mx;
  AABB() : mn(v3(kInfinity, kInfinity, kInfinity)),
           mx(v3(-kInfinity, -kInfinity, -kInfinity)) {}
  void grow(V3 p) {
    mn = vmin(mn, p);
    mx = vmax(mx, p);
  }
  void grow(const AABB &b) {
    mn = vmin(mn, b.mn);
    mx = vmax(mx, b.mx);
  }
  float half_area() const {
    V3 e = mx - mn;
    return e.x * e.y + e.y * e.z + e.x * e.z;
  }
};
struct Bin { // BVHNode.cpp:9-12
  AABB bounds;
  uint32_t tri_count = 0;
};

struct BlasBuild {
  rt_bvh_node *nodes;
  const rt_triangle_geom *tris;
  const float *centroids;
  uint32_t *tri_ids;
  uint32_t nodes_count;
};

// BVHNode.cpp:14-74
float find_best_split_plane(BlasBuild &b, rt_bvh_node &node, int &axis,
                            float &split_pos) {
  float best_cost = kInfinity;
  for (int a = 0; a < 3; ++a) {
    float bounds_min = kInfinity, bounds_max = -kInfinity;
    for (uint32_t i = 0; i < node.tri_count; ++i) {
      float c = b.centroids[3 * b.tri_ids[node.local_left_first + i] + a];
      bounds_min = c < bounds_min ? c : bounds_min; // std::min(bounds_min, c)
      bounds_max = bounds_max < c ? c : bounds_max; // std::max(bounds_max, c)
    }
    if (bounds_min == bounds_max)
      continue;
    Bin bins[BIN_COUNT];
    float scale = BIN_COUNT / (bounds_max - bounds_min);
    for (uint32_t i = 0; i < node.tri_count; ++i) {
      uint32_t id = b.tri_ids[node.local_left_first + i];
      const rt_triangle_geom &t = b.tris[id];
      float c = b.centroids[3 * id + a];
      uint32_t raw = (uint32_t)((c - bounds_min) * scale);
      uint32_t bin_idx = raw < BIN_COUNT - 1 ? raw : BIN_COUNT - 1;
      Bin &bin = bins[bin_idx];
      bin.tri_count++;
      bin.bounds.grow(ld3(t.v0));
      bin.bounds.grow(ld3(t.v1));
      bin.bounds.grow(ld3(t.v2));
    }
    float left_half_area[BIN_COUNT - 1], right_half_area[BIN_COUNT - 1];
    int left_count[BIN_COUNT - 1], right_count[BIN_COUNT - 1];
    AABB left_box, right_box;
    int left_sum = 0, right_sum = 0;
    for (uint32_t i = 0; i < BIN_COUNT - 1; ++i) {
      left_sum += bins[i].tri_count;
      left_count[i] = left_sum;
      left_box.grow(bins[i].bounds);
      left_half_area[i] = left_box.half_area();
      right_sum += bins[BIN_COUNT - 1 - i].tri_count;
      right_count[BIN_COUNT - 2 - i] = right_sum;
      right_box.grow(bins[BIN_COUNT - 1 - i].bounds);
      right_half_area[BIN_COUNT - 2 - i] = right_box.half_area();
    }
    scale = (bounds_max - bounds_min) / BIN_COUNT;
    for (uint32_t i = 0; i < BIN_COUNT - 1; ++i) {
      float plane_cost = left_count[i] * left_half_area[i] +
                         right_count[i] * right_half_area[i];
      if (plane_cost < best_cost) {
        best_cost = plane_cost;
        split_pos = bounds_min + scale * (i + 1);
        axis = a;
      }
    }
  }
  return best_cost;
}

// BVHNode.cpp:89-105 (glm::min(a,b) = b<a?b:a, glm::max(a,b) = a<b?b:a)
void update_node_bounds(BlasBuild &b, uint32_t node_idx) {
  rt_bvh_node &node = b.nodes[node_idx];
  V3 mn = v3(kInfinity, kInfinity, kInfinity);
  V3 mx = v3(-kInfinity, -kInfinity, -kInfinity);
  for (uint32_t i = 0; i < node.tri_count; ++i) {
    const rt_triangle_geom &t = b.tris[b.tri_ids[node.local_left_first + i]];
    mn = vmin(mn, ld3(t.v0));
    mn = vmin(mn, ld3(t.v1));
    mn = vmin(mn, ld3(t.v2));
    mx = vmax(mx, ld3(t.v0));
    mx = vmax(mx, ld3(t.v1));
    mx = vmax(mx, ld3(t.v2));
  }
  st3(node.aabb_min, mn);
  st3(node.aabb_max, mx);
}

// BVHNode.cpp:107-161
void subdivide(BlasBuild &b, uint32_t node_idx) {
  rt_bvh_node &node = b.nodes[node_idx];
  int best_axis = -1;
  float best_pos = 0.f;
  float best_cost = find_best_split_plane(b, node, best_axis, best_pos);

  V3 e = ld3(node.aabb_max) - ld3(node.aabb_min);
  float parent_area = e.x * e.y + e.y * e.z + e.z * e.x;
  float parent_cost = node.tri_count * parent_area;
  if (best_cost >= parent_cost)
    return;

  int axis = best_axis;
  float split_pos = best_pos;
  // The reference indexes with u32 and never sees first == 0 (its pool offsets
  // start behind the TLSF header); 64-bit signed indices give the same
  // partition for every first, including 0.
  int64_t l = node.local_left_first;
  int64_t r = l + node.tri_count - 1;
  while (l <= r) {
    if (b.centroids[3 * b.tri_ids[l] + axis] < split_pos) {
      ++l;
    } else {
      uint32_t tmp = b.tri_ids[l];
      b.tri_ids[l] = b.tri_ids[r];
      b.tri_ids[r] = tmp;
      --r;
    }
  }
  uint32_t left_count = (uint32_t)(l - node.local_left_first);
  if (left_count == 0 || left_count == node.tri_count)
    return;

  uint32_t left_idx = b.nodes_count++;
  uint32_t right_idx = b.nodes_count++;
  rt_bvh_node &left = b.nodes[left_idx];
  left.local_left_first = node.local_left_first;
  left.tri_count = left_count;
  rt_bvh_node &right = b.nodes[right_idx];
  right.local_left_first = (uint32_t)l;
  right.tri_count = node.tri_count - left_count;
  node.local_left_first = left_idx;
  node.tri_count = 0;
  update_node_bounds(b, left_idx);
  update_node_bounds(b, right_idx);
  subdivide(b, left_idx);
  subdivide(b, right_idx);
}
} // namespace

// BLAS::build, BVHNode.cpp:76-87
extern "C" uint32_t orc_build_blas(rt_bvh_node *nodes,
                                   const rt_triangle_geom *geom,
                                   const float *centroids, uint32_t *tri_ids,
                                   uint32_t tri_count, uint32_t tri_id_offset) {
  BlasBuild b{nodes, geom, centroids, tri_ids, 1};
  nodes[0].local_left_first = tri_id_offset;
  nodes[0].tri_count = tri_count;
  update_node_bounds(b, 0);
  subdivide(b, 0);
  return b.nodes_count;
}

// BLAS::refit, BVHNode.cpp:163-177
extern "C" void orc_refit_blas(rt_bvh_node *nodes, uint32_t nodes_count,
                               const rt_triangle_geom *geom,
                               const uint32_t *tri_ids) {
  BlasBuild b{nodes, geom, nullptr, const_cast<uint32_t *>(tri_ids),
              nodes_count};
  for (int i = int(nodes_count) - 1; i >= 0; --i) {
    rt_bvh_node &node = nodes[i];
    if (node.tri_count) {
      update_node_bounds(b, i);
    } else {
      rt_bvh_node &lc = nodes[node.local_left_first];
      rt_bvh_node &rc = nodes[node.local_left_first + 1];
      st3(node.aabb_min, vmin(ld3(lc.aabb_min), ld3(rc.aabb_min)));
      st3(node.aabb_max, vmax(ld3(lc.aabb_max), ld3(rc.aabb_max)));
    }
  }
}

// ---------------------------------------------------------------------------
// Host: TLAS builder — src/TLAS.cpp:6-80
// ---------------------------------------------------------------------------
namespace {
// TLAS.cpp:61-80
int find_best_match(const rt_tlas_node *nodes, const int *list, int N, int a) {
  float smallest = kInfinity;
  int best_b = -1;
  for (int b = 0; b < N; ++b) {
    if (b != a) {
      V3 bmax = vmax(ld3(nodes[list[a]].aabb_max), ld3(nodes[list[b]].aabb_max));
      V3 bmin = vmin(ld3(nodes[list[a]].aabb_min), ld3(nodes[list[b]].aabb_min));
      V3 e = bmax - bmin;
      float h_surface_area = e.x * e.y + e.y * e.z + e.z * e.x;
      if (h_surface_area < smallest) {
        smallest = h_surface_area;
        best_b = b;
      }
    }
  }
  return best_b;
}
} // namespace

extern "C" uint32_t orc_build_tlas(rt_tlas_node *tlas_nodes,
                                   const rt_blas_instance *instances,
                                   const uint32_t *instance_ids, uint32_t n,
                                   const rt_blas *blas,
                                   const rt_bvh_node *bvh_nodes) {
  if (n == 0)
    return 0;
  uint32_t node_count = 1;
  std::vector<int> node_ids(n);
  int node_indices = (int)n;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t inst_id = instance_ids[i];
    uint32_t blas_idx = instances[inst_id].blas_id;
    const rt_bvh_node &root = bvh_nodes[blas[blas_idx].bvh_nodes_offset];
    V3 bmin = ld3(root.aabb_min), bmax = ld3(root.aabb_max);
    AABB bounds;
    float transform[16];
    orc_mat4_inverse(instances[inst_id].inv_transform, transform); // sic: double inverse
    for (int j = 0; j < 8; ++j) {
      float corner[4] = {(j & 1) ? bmax.x : bmin.x, (j & 2) ? bmax.y : bmin.y,
                         (j & 4) ? bmax.z : bmin.z, 1.0f};
      float wp[4];
      orc_mat4_mul_vec4(transform, corner, wp);
      bounds.grow(ld3(wp));
    }
    node_ids[i] = (int)node_count;
    st3(tlas_nodes[node_count].aabb_min, bounds.mn);
    st3(tlas_nodes[node_count].aabb_max, bounds.mx);
    tlas_nodes[node_count].blas_instance_idx = inst_id;
    tlas_nodes[node_count++].left_right = 0;
  }
  int a = 0, b = find_best_match(tlas_nodes, node_ids.data(), node_indices, a);
  while (node_indices > 1) {
    int c = find_best_match(tlas_nodes, node_ids.data(), node_indices, b);
    if (a == c) {
      int node_id_a = node_ids[a], node_id_b = node_ids[b];
      rt_tlas_node &node_a = tlas_nodes[node_id_a];
      rt_tlas_node &node_b = tlas_nodes[node_id_b];
      rt_tlas_node &new_node = tlas_nodes[node_count];
      new_node.left_right = (uint32_t)node_id_a + ((uint32_t)node_id_b << 16);
      new_node.blas_instance_idx = 0;
      st3(new_node.aabb_min, vmin(ld3(node_a.aabb_min), ld3(node_b.aabb_min)));
      st3(new_node.aabb_max, vmax(ld3(node_a.aabb_max), ld3(node_b.aabb_max)));
      node_ids[a] = (int)node_count++;
      node_ids[b] = node_ids[node_indices - 1];
      b = find_best_match(tlas_nodes, node_ids.data(), --node_indices, a);
    } else {
      a = b, b = c;
    }
  }
  tlas_nodes[0] = tlas_nodes[node_ids[a]];
  node_count--;
  return node_count;
}

// ---------------------------------------------------------------------------
// Shader: RNG — Random.slang
// ---------------------------------------------------------------------------
extern "C" uint32_t orc_wang_hash(uint32_t *seed) { // Random.slang:7-14
  uint32_t s = *seed;
  s = (s ^ 61u) ^ (s >> 16);
  s *= 9u;
  s = s ^ (s >> 4);
  s *= 0x27d4eb2du;
  s = s ^ (s >> 15);
  *seed = s;
  return s;
}

extern "C" float orc_rand(uint32_t *seed) { // Random.slang:17-19
  return (float)orc_wang_hash(seed) / 4294967296.f;
}

// Deterministic sin/cos: nearest-quadrant Cody-Waite reduction by pi/2 in three
// parts, then the single-precision minimax polynomials on [-pi/4, pi/4].  Every
// operation is a separately rounded fp32 mul/add, so any IEEE machine gives
// the same bits.  (Driver-defined in the reference, Random.slang:38-39.)
extern "C" void orc_sincos(float a, float *s_out, float *c_out) {
  float kf = a * 0.636619772f;
  int k = (int)(kf + (kf < 0.0f ? -0.5f : 0.5f));
  float fk = (float)k;
  float r = a - fk * 1.5703125f;
  r = r - fk * 4.837512969970703125e-4f;
  r = r - fk * 7.54978995489188216e-8f;
  float z = r * r;
  float ps = -1.9515295891e-4f * z;
  ps = ps + 8.3321608736e-3f;
  ps = ps * z;
  ps = ps + -1.6666654611e-1f;
  ps = ps * z;
  ps = ps * r;
  float s = ps + r;
  float pc = 2.443315711809948e-5f * z;
  pc = pc + -1.388731625493765e-3f;
  pc = pc * z;
  pc = pc + 4.166664568298827e-2f;
  pc = pc * z;
  pc = pc * z;
  float c = pc - 0.5f * z;
  c = c + 1.0f;
  switch (k & 3) {
  case 0:
    *s_out = s, *c_out = c;
    break;
  case 1:
    *s_out = c, *c_out = -s;
    break;
  case 2:
    *s_out = -s, *c_out = -c;
    break;
  default:
    *s_out = -c, *c_out = s;
    break;
  }
}

namespace {

// PCG hash (Jarzynski & Olano, "Hash Functions for GPU Rendering", 2020).
inline uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

// RNG state of one pixel.  WANG: the reference's sequential state.  PCG: the
// north-star's counter-based generator keyed on (pixel, sample, bounce, dim).
struct Rng {
  uint32_t mode;
  uint32_t seed;   // WANG state
  uint32_t pixel;  // PCG key: y * width + x
  uint32_t sample; // PCG key: frame_index * spp + sample-in-frame
  uint32_t slot;   // PCG key: 0 = camera, d + 1 = bounce d
  uint32_t dim;    // PCG key: draw counter inside the slot
  float next() {
    if (mode == RT_RNG_WANG)
      return orc_rand(&seed);
    uint32_t h =
        pcg_hash(pixel + pcg_hash(sample + pcg_hash(slot * 64u + dim)));
    ++dim;
    return (float)h / 4294967296.f;
  }
  void set_slot(uint32_t s) {
    slot = s;
    dim = 0;
  }
};

const float kPi = 3.14159265359f;  // Random.slang:3
const float kTwoPi = 2.0f * kPi;   // Random.slang:4

// Random.slang:34-41 (rand_range(min,max) = min + (max - min) * rand)
V3 rand_unit_vector(Rng &rng) {
  float z = -1.f + (1.f - -1.f) * rng.next();
  float a = rng.next() * kTwoPi;
  float r = sqrtf(1.f - z * z);
  float s, c;
  orc_sincos(a, &s, &c);
  float x = r * c;
  float y = r * s;
  return v3(x, y, z);
}

// Random.slang:51-55
bool near_zero(V3 v) {
  float s = 1e-8f;
  return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

// Random.slang:57-63
V3 rand_in_unit_disk(Rng &rng) {
  while (true) {
    float px = -1.f + (1.f - -1.f) * rng.next();
    float py = -1.f + (1.f - -1.f) * rng.next();
    V3 p = v3(px, py, 0.f);
    if (dot(p, p) < 1.f)
      return p;
  }
}

struct Ray {
  V3 origin, direction;
};

struct HitRecord { // HitRecord.slang:5-22
  V3 p, normal;
  uint32_t tri_geom_id, tri_surface_id, blas_instance_id;
  float t;
  bool front_face;
  float u, v;
  void set_face_normal(const Ray &r, V3 outward_normal) {
    front_face = dot(r.direction, outward_normal) < 0.f;
    normal = front_face ? outward_normal : neg(outward_normal);
  }
};

// BVHNode.slang:8-26 (RT_AABB_DIV) / AABB.hpp:29-48 (RT_AABB_RCP)
template <int MODE>
inline float intersect_aabb(const Ray &ray, V3 inv_dir, V3 bmin, V3 bmax,
                            float t) {
  float tx1, tx2, ty1, ty2, tz1, tz2;
  if (MODE == RT_AABB_DIV) {
    tx1 = (bmin.x - ray.origin.x) / ray.direction.x;
    tx2 = (bmax.x - ray.origin.x) / ray.direction.x;
  } else {
    tx1 = (bmin.x - ray.origin.x) * inv_dir.x;
    tx2 = (bmax.x - ray.origin.x) * inv_dir.x;
  }
  float tmin = fmin_(tx1, tx2);
  float tmax = fmax_(tx1, tx2);
  if (MODE == RT_AABB_DIV) {
    ty1 = (bmin.y - ray.origin.y) / ray.direction.y;
    ty2 = (bmax.y - ray.origin.y) / ray.direction.y;
  } else {
    ty1 = (bmin.y - ray.origin.y) * inv_dir.y;
    ty2 = (bmax.y - ray.origin.y) * inv_dir.y;
  }
  tmin = fmax_(tmin, fmin_(ty1, ty2));
  tmax = fmin_(tmax, fmax_(ty1, ty2));
  if (MODE == RT_AABB_DIV) {
    tz1 = (bmin.z - ray.origin.z) / ray.direction.z;
    tz2 = (bmax.z - ray.origin.z) / ray.direction.z;
  } else {
    tz1 = (bmin.z - ray.origin.z) * inv_dir.z;
    tz2 = (bmax.z - ray.origin.z) * inv_dir.z;
  }
  tmin = fmax_(tmin, fmin_(tz1, tz2));
  tmax = fmin_(tmax, fmax_(tz1, tz2));
  if (tmax >= tmin && tmin < t && tmax > 0)
    return tmin;
  else
    return kMiss;
}

inline V3 inv_direction(V3 d) { return v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// Triangle.slang:6-42
inline bool triangle_hit(const rt_triangle_geom &tri, const Ray &r, float tmin,
                         float tmax, HitRecord &rec) {
  V3 v0 = ld3(tri.v0);
  V3 edge_1 = ld3(tri.v1) - v0;
  V3 edge_2 = ld3(tri.v2) - v0;
  V3 h = cross(r.direction, edge_2);
  float a = dot(edge_1, h);
  if (a > -kEpsilon && a < kEpsilon)
    return false;
  float f = 1.f / a;
  V3 s = r.origin - v0;
  float u = f * dot(s, h);
  if (u < 0 || u > 1)
    return false;
  V3 q = cross(s, edge_1);
  float v = f * dot(r.direction, q);
  if (v < 0 || u + v > 1)
    return false;
  float t = f * dot(edge_2, q);
  if (t < kEpsilon || !(tmin <= t && t <= tmax))
    return false;
  rec.t = t;
  rec.p = r.origin + t * r.direction;
  rec.u = u;
  rec.v = v;
  // The face normal written here (Triangle.slang:38-39) is always overwritten
  // by the interpolated one before use (RayTracing.slang:111-114).
  return true;
}

// BLAS::intersect, BVHNode.slang:43-109
template <int MODE>
bool blas_intersect(const rt_scene_desc *sc, const rt_blas &blas, const Ray &ray,
                    float tmin, float tmax, HitRecord &rec, orc_counters &cn) {
  uint32_t node_id_stack[128];
  node_id_stack[0] = blas.bvh_nodes_offset;
  uint32_t stack_ptr = 0;
  float closest_so_far = tmax;
  bool hit = false;
  V3 inv_dir = MODE == RT_AABB_RCP ? inv_direction(ray.direction) : v3(0, 0, 0);
  const rt_bvh_node *nodes = sc->bvh_nodes;
  while (true) {
    const rt_bvh_node &node = nodes[node_id_stack[stack_ptr]];
    if (node.tri_count > 0) {
      ++cn.blas_leaf;
      for (uint32_t i = 0; i < node.tri_count; ++i) {
        uint32_t tri_index = sc->tri_ids[node.local_left_first + i];
        ++cn.tris_tested;
        if (triangle_hit(sc->tri_geom[tri_index], ray, tmin, closest_so_far,
                         rec)) {
          hit = true;
          rec.tri_geom_id = tri_index;
          rec.tri_surface_id = tri_index;
          closest_so_far = rec.t;
        }
      }
      if (stack_ptr == 0)
        break;
      else
        --stack_ptr;
    } else {
      ++cn.blas_interior;
      uint32_t child1_idx = blas.bvh_nodes_offset + node.local_left_first;
      uint32_t child2_idx = child1_idx + 1;
      const rt_bvh_node &child1 = nodes[child1_idx];
      const rt_bvh_node &child2 = nodes[child2_idx];
      float dist1 = intersect_aabb<MODE>(ray, inv_dir, ld3(child1.aabb_min),
                                         ld3(child1.aabb_max), closest_so_far);
      float dist2 = intersect_aabb<MODE>(ray, inv_dir, ld3(child2.aabb_min),
                                         ld3(child2.aabb_max), closest_so_far);
      if (dist1 > dist2) {
        float temp = dist1;
        dist1 = dist2;
        dist2 = temp;
        uint32_t temp_c = child1_idx;
        child1_idx = child2_idx;
        child2_idx = temp_c;
      }
      if (dist1 == kMiss) {
        if (stack_ptr == 0)
          break;
        else
          --stack_ptr;
      } else {
        if (dist2 != kMiss) {
          node_id_stack[stack_ptr++] = child2_idx;
        }
        node_id_stack[stack_ptr] = child1_idx;
        if (stack_ptr > cn.max_stack)
          cn.max_stack = stack_ptr;
      }
    }
  }
  return hit;
}

// intersect_tlas, TLAS.slang:14-67 (+ BLASInstance::intersect, BVHNode.slang:120-128)
template <int MODE>
bool intersect_tlas(const rt_scene_desc *sc, const Ray &ray, float tmin,
                    float tmax, HitRecord &rec, orc_counters &cn) {
  ++cn.rays;
  if (sc->tlas_node_count == 0)
    return false; // empty scene: the reference never dispatches over one
  uint32_t node_id_stack[128];
  node_id_stack[0] = 0;
  uint32_t stack_ptr = 0;
  float closest_so_far = tmax;
  bool hit = false;
  V3 inv_dir = MODE == RT_AABB_RCP ? inv_direction(ray.direction) : v3(0, 0, 0);
  const rt_tlas_node *tlas = sc->tlas_nodes;
  while (true) {
    const rt_tlas_node &node = tlas[node_id_stack[stack_ptr]];
    if (node.left_right == 0) {
      ++cn.tlas_leaf;
      const rt_blas_instance &inst = sc->instances[node.blas_instance_idx];
      Ray oray;
      oray.origin = mul_m4_v(inst.inv_transform, ray.origin, 1.f);
      oray.direction = mul_m4_v(inst.inv_transform, ray.direction, 0.f);
      if (blas_intersect<MODE>(sc, sc->blas[inst.blas_id], oray, tmin,
                               closest_so_far, rec, cn)) {
        hit = true;
        closest_so_far = rec.t;
        rec.blas_instance_id = node.blas_instance_idx;
      }
      if (stack_ptr == 0)
        break;
      else
        --stack_ptr;
    } else {
      ++cn.tlas_interior;
      uint32_t child1_idx = node.left_right & 0x0000FFFF;
      uint32_t child2_idx = (node.left_right & 0xFFFF0000) >> 16;
      const rt_tlas_node &child1 = tlas[child1_idx];
      const rt_tlas_node &child2 = tlas[child2_idx];
      float dist1 = intersect_aabb<MODE>(ray, inv_dir, ld3(child1.aabb_min),
                                         ld3(child1.aabb_max), closest_so_far);
      float dist2 = intersect_aabb<MODE>(ray, inv_dir, ld3(child2.aabb_min),
                                         ld3(child2.aabb_max), closest_so_far);
      if (dist1 > dist2) {
        float temp = dist1;
        dist1 = dist2;
        dist2 = temp;
        uint32_t temp_c = child1_idx;
        child1_idx = child2_idx;
        child2_idx = temp_c;
      }
      if (dist1 == kMiss) {
        if (stack_ptr == 0)
          break;
        else
          --stack_ptr;
      } else {
        if (dist2 != kMiss) {
          node_id_stack[stack_ptr++] = child2_idx;
        }
        node_id_stack[stack_ptr] = child1_idx;
        if (stack_ptr > cn.max_stack)
          cn.max_stack = stack_ptr;
      }
    }
  }
  return hit;
}

inline bool trace(const rt_scene_desc *sc, uint32_t mode, const Ray &ray,
                  float tmin, float tmax, HitRecord &rec, orc_counters &cn) {
  return mode == RT_AABB_RCP
             ? intersect_tlas<RT_AABB_RCP>(sc, ray, tmin, tmax, rec, cn)
             : intersect_tlas<RT_AABB_DIV>(sc, ray, tmin, tmax, rec, cn);
}

// Nearest / repeat RGBA8-UNORM fetch (sampler: Renderer.cpp:347-361; format:
// Material.cpp:170).  Texel i = floor(u * w) wrapped into [0, w).
V3 sample_albedo(const rt_scene_desc *sc, uint32_t tex_index, float u, float v) {
  const rt_texture_desc &td = sc->textures[tex_index];
  float fu = floorf(u * (float)td.width);
  float fv = floorf(v * (float)td.height);
  int w = (int)td.width, h = (int)td.height;
  // wrap in float first so the int conversion is always in range
  fu = fu - floorf(fu / (float)w) * (float)w;
  fv = fv - floorf(fv / (float)h) * (float)h;
  int ix = (fu == fu) ? (int)fu : 0;
  int iy = (fv == fv) ? (int)fv : 0;
  if (ix < 0) ix = 0;
  if (ix > w - 1) ix = w - 1;
  if (iy < 0) iy = 0;
  if (iy > h - 1) iy = h - 1;
  const uint8_t *px = sc->texels + td.offset + 4u * ((uint32_t)iy * td.width + (uint32_t)ix);
  return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

inline V3 reflect(V3 i, V3 n) { return i - (2.0f * dot(n, i)) * n; }

// Material.slang:63-68
float reflectance(float cosine, float refraction_index) {
  float r0 = (1 - refraction_index) / (1 + refraction_index);
  r0 = r0 * r0;
  float x = 1 - cosine;
  float x2 = x * x;
  float x5 = (x2 * x2) * x;
  return r0 + (1 - r0) * x5;
}

// Material.slang:70-75
V3 rtiow_refract(V3 uv, V3 n, float etai_over_etat) {
  float cos_theta = fmin_(dot(neg(uv), n), 1.0f);
  V3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
  V3 r_out_parallel = (-sqrtf(fabsf(1.0f - dot(r_out_perp, r_out_perp)))) * n;
  return r_out_perp + r_out_parallel;
}

struct Capture {
  rt_ray *rays;
  uint64_t cap;
  uint64_t count;
  uint32_t bounce;
};

// One pixel of compute_main, RayTracing.slang:66-180.
void shade_pixel(const rt_scene_desc *sc, const rt_camera_uniforms *cam,
                 const rt_render_params *pc, uint32_t x, uint32_t y,
                 float *accum, orc_counters &cn, Capture *capture) {
  Rng rng;
  rng.mode = pc->rng_mode;
  rng.seed = (x * 1973u) ^ (y * 9277u) ^ (pc->frame_index * 26699u); // :71-72
  rng.pixel = y * pc->image_width + x;
  rng.sample = 0;
  rng.slot = 0;
  rng.dim = 0;
  const uint32_t sqrt_spp = pc->sqrt_spp;
  const float recip_sqrt_spp = 1.f / sqrt_spp;                 // Renderer.cpp:628
  const float pixel_sample_scale = 1.f / (sqrt_spp * sqrt_spp); // Renderer.cpp:626
  const V3 pixel00 = ld3(cam->pixel00_loc), du = ld3(cam->pixel_delta_u),
           dv = ld3(cam->pixel_delta_v), center = ld3(cam->camera_center);
  const bool defocus = cam->defocus_disk_u[3] != 0.f;

  V3 radiance = v3(0.f, 0.f, 0.f);
  uint32_t sample_in_frame = 0;
  for (uint32_t s_j = 0; s_j < sqrt_spp; ++s_j)
    for (uint32_t s_i = 0; s_i < sqrt_spp; ++s_i, ++sample_in_frame) {
      rng.sample = pc->frame_index * (sqrt_spp * sqrt_spp) + sample_in_frame;
      rng.set_slot(0);
      // sample_square_stratified, :47-57
      float jx = (((float)s_i + rng.next()) * recip_sqrt_spp) - 0.5f;
      float jy = (((float)s_j + rng.next()) * recip_sqrt_spp) - 0.5f;
      V3 pixel_sample = (pixel00 + (((float)(int)x + jx) * du)) +
                        (((float)(int)y + jy) * dv); // :81-84
      V3 origin = center;
      if (defocus) { // Random.slang:65-70 (unused by the reference's kernel)
        V3 p = rand_in_unit_disk(rng);
        origin = (center + (p.x * ld3(cam->defocus_disk_u))) +
                 (p.y * ld3(cam->defocus_disk_v));
      }
      Ray r;
      r.origin = origin;
      r.direction = normalize(pixel_sample - origin); // :86
      V3 attenuation = v3(1.f, 1.f, 1.f);
      V3 sample_radiance = v3(0.f, 0.f, 0.f);

      for (uint32_t d = 0; d < pc->max_depth; ++d) { // :92
        rng.set_slot(d + 1);
        if (capture && sample_in_frame == 0 && d == capture->bounce &&
            capture->count < capture->cap) {
          rt_ray &o = capture->rays[capture->count++];
          st3(o.origin, r.origin);
          st3(o.direction, r.direction);
          o.t_min = 0.0001f;
          o.t_max = 1000.f;
        }
        HitRecord rec;
        rec.t = 1000.f;
        V3 emission = v3(0.f, 0.f, 0.f);
        bool hit_anything = trace(sc, pc->aabb_mode, r, 0.0001f, 1000.f, rec, cn);
        if (hit_anything) {
          ++cn.shaded_hits;
          const rt_blas_instance &inst = sc->instances[rec.blas_instance_id];
          const rt_triangle_shading &sh = sc->tri_shading[rec.tri_surface_id];
          float alpha = 1.f - rec.u - rec.v; // Triangle.slang:50-58
          V3 local_normal = normalize((alpha * ld3(sh.n0) + rec.u * ld3(sh.n1)) +
                                      rec.v * ld3(sh.n2));
          rec.set_face_normal(
              r, normalize(mul_m4t_v0(inst.inv_transform, local_normal)));
          float uvx = (alpha * sh.uv0[0] + rec.u * sh.uv1[0]) + rec.v * sh.uv2[0];
          float uvy = (alpha * sh.uv0[1] + rec.u * sh.uv1[1]) + rec.v * sh.uv2[1];
          rec.p = mul_m4_v(inst.transform, rec.p, 1.f); // :117

          bool ray_scattered = false;
          Ray r_out = r;
          V3 material_attenuation = v3(0.f, 0.f, 0.f);
          switch (inst.material.type) {
          case RT_MATERIAL_LAMBERT: { // Material.slang:23-37
            V3 scattered_direction = rec.normal + rand_unit_vector(rng);
            if (near_zero(scattered_direction))
              scattered_direction = rec.normal;
            r_out.origin = rec.p;
            r_out.direction = scattered_direction;
            material_attenuation =
                sample_albedo(sc, sc->lamberts[inst.material.index].index, uvx, uvy);
            ray_scattered = true;
            break;
          }
          case RT_MATERIAL_METAL: { // Material.slang:39-57, NORMALIZE_REFLECTION branch
            const rt_metal &m = sc->metals[inst.material.index];
            V3 reflected = reflect(r.direction, rec.normal);
            reflected = reflected + (m.albedo_fuzz[3] * rand_unit_vector(rng));
            r_out.origin = rec.p;
            r_out.direction = reflected;
            material_attenuation = ld3(m.albedo_fuzz);
            ray_scattered = true;
            break;
          }
          case RT_MATERIAL_DIELECTRIC: { // Material.slang:77-100
            float refraction_index =
                sc->dielectrics[inst.material.index].refraction_index;
            material_attenuation = v3(1.f, 1.f, 1.f);
            float ri = rec.front_face ? (1.f / refraction_index) : refraction_index;
            V3 unit_direction = normalize(r.direction);
            float cos_theta = fmin_(dot(neg(unit_direction), rec.normal), 1.f);
            float sin_theta = sqrtf(fmax_(0.f, 1.f - cos_theta * cos_theta));
            bool cannot_refract = ri * sin_theta > 1.f;
            V3 direction;
            // `||` short-circuits: no draw when the ray cannot refract — unless the caller selects the other
            // reading of Material.slang:91 (RT_RENDER_DIELECTRIC_ALWAYS_DRAWS: both operands evaluated)
            bool reflects;
            if (pc->flags & RT_RENDER_DIELECTRIC_ALWAYS_DRAWS) {
              const float xi = rng.next();
              reflects = cannot_refract || reflectance(cos_theta, ri) > xi;
            } else {
              reflects = cannot_refract || reflectance(cos_theta, ri) > rng.next();
            }
            if (reflects)
              direction = reflect(unit_direction, rec.normal);
            else
              direction = rtiow_refract(unit_direction, rec.normal, ri);
            r_out.origin = rec.p;
            r_out.direction = direction;
            ray_scattered = true;
            break;
          }
          case RT_MATERIAL_EMISSIVE: // Material.slang:102-108
            emission = ld3(sc->emissives[inst.material.index].intensity);
            ray_scattered = false;
            break;
          default:
            break;
          }
          sample_radiance = sample_radiance + attenuation * emission; // :151
          if (!ray_scattered)
            break;
          attenuation = attenuation * material_attenuation;
          r = r_out;
        } else { // :159-167
          V3 unit_direction = normalize(r.direction);
          float a = 0.5f * (unit_direction.y + 1.f);
          // (RT_RENDER_SKY_WHITE_HORIZON: float3(1.0) for float3(0.7) — the revision behind Screenshot3.png, rt_types.h)
          const float hz = (pc->flags & RT_RENDER_SKY_WHITE_HORIZON) ? 1.0f : 0.7f;
          V3 sky = v3(hz, hz, hz) * (1.f - a) + v3(0.5f, 0.7f, 1.f) * a;
          sample_radiance = sample_radiance + attenuation * sky;
          break;
        }
      }
      radiance = radiance + sample_radiance;
    }
  if (capture)
    return;
  radiance = radiance * pixel_sample_scale; // :174
  float *px = accum + 4ull * ((uint64_t)y * pc->image_width + x);
  V3 prev = ld3(px);
  float fi = (float)pc->accum_count;
  float fi1 = (float)(pc->accum_count + 1u);
  V3 accumulated = (prev * fi + radiance) / fi1; // :176-179
  st3(px, accumulated);
  px[3] = 1.f;
}

void add_counters(orc_counters &dst, const orc_counters &src) {
  dst.rays += src.rays;
  dst.tlas_interior += src.tlas_interior;
  dst.tlas_leaf += src.tlas_leaf;
  dst.blas_interior += src.blas_interior;
  dst.blas_leaf += src.blas_leaf;
  dst.tris_tested += src.tris_tested;
  dst.shaded_hits += src.shaded_hits;
  if (src.max_stack > dst.max_stack)
    dst.max_stack = src.max_stack;
}

} // namespace

extern "C" void orc_trace_closest(const rt_scene_desc *scene, const rt_ray *rays,
                                  uint64_t n, rt_hit *hits, uint32_t aabb_mode,
                                  orc_counters *counters) {
  orc_counters cn;
  memset(&cn, 0, sizeof cn);
  for (uint64_t i = 0; i < n; ++i) {
    Ray r;
    r.origin = ld3(rays[i].origin);
    r.direction = ld3(rays[i].direction);
    HitRecord rec;
    rec.t = rays[i].t_max;
    rec.u = rec.v = 0.f;
    rec.tri_geom_id = rec.blas_instance_id = RT_INVALID_ID;
    bool hit = trace(scene, aabb_mode, r, rays[i].t_min, rays[i].t_max, rec, cn);
    rt_hit &h = hits[i];
    memset(&h, 0, sizeof h);
    h.t = hit ? rec.t : rays[i].t_max;
    h.tri_id = hit ? rec.tri_geom_id : RT_INVALID_ID;
    h.inst_id = hit ? rec.blas_instance_id : RT_INVALID_ID;
    h.u = hit ? rec.u : 0.f;
    h.v = hit ? rec.v : 0.f;
  }
  if (counters)
    add_counters(*counters, cn);
}

extern "C" void orc_trace_brute(const rt_scene_desc *sc, const rt_ray *rays,
                                uint64_t n, rt_hit *hits) {
  for (uint64_t i = 0; i < n; ++i) {
    Ray r;
    r.origin = ld3(rays[i].origin);
    r.direction = ld3(rays[i].direction);
    rt_hit &h = hits[i];
    memset(&h, 0, sizeof h);
    h.t = rays[i].t_max;
    h.tri_id = h.inst_id = RT_INVALID_ID;
    for (uint32_t k = 0; k < sc->tlas_node_count; ++k) {
      const rt_tlas_node &leaf = sc->tlas_nodes[k];
      if (leaf.left_right != 0)
        continue;
      const rt_blas_instance &inst = sc->instances[leaf.blas_instance_idx];
      const rt_blas &blas = sc->blas[inst.blas_id];
      Ray o;
      o.origin = mul_m4_v(inst.inv_transform, r.origin, 1.f);
      o.direction = mul_m4_v(inst.inv_transform, r.direction, 0.f);
      // the BLAS's triangles are the tri_ids slots of its root range; walk the
      // node array's leaves to find them
      for (uint32_t nidx = 0; nidx < blas.nodes_count; ++nidx) {
        const rt_bvh_node &node = sc->bvh_nodes[blas.bvh_nodes_offset + nidx];
        for (uint32_t j = 0; j < node.tri_count; ++j) {
          uint32_t tri = sc->tri_ids[node.local_left_first + j];
          HitRecord rec;
          if (triangle_hit(sc->tri_geom[tri], o, rays[i].t_min, h.t, rec) &&
              rec.t < h.t) {
            h.t = rec.t;
            h.tri_id = tri;
            h.inst_id = leaf.blas_instance_idx;
            h.u = rec.u;
            h.v = rec.v;
          }
        }
      }
    }
  }
}

extern "C" void orc_render(const rt_scene_desc *scene,
                           const rt_camera_uniforms *cam,
                           const rt_render_params *params, float *accum,
                           orc_counters *counters, uint32_t nthreads) {
  uint32_t y0 = params->row_begin, y1 = params->row_end;
  if (y0 == 0 && y1 == 0)
    y1 = params->image_height;
  if (y1 > params->image_height)
    y1 = params->image_height;
  if (nthreads == 0)
    nthreads = 1;
  std::vector<orc_counters> per(nthreads);
  memset(per.data(), 0, sizeof(orc_counters) * nthreads);
  auto work = [&](uint32_t tid) {
    // row_group_stride N > 1: only the 8-row groups 0, N, 2N, ... counted from row_begin (rt_types.h)
    const uint32_t stride = params->row_group_stride ? params->row_group_stride : 1u;
    for (uint32_t y = y0 + tid; y < y1; y += nthreads) {
      if (((y - y0) >> 3) % stride != 0u)
        continue;
      for (uint32_t x = 0; x < params->image_width; ++x)
        shade_pixel(scene, cam, params, x, y, accum, per[tid], nullptr);
    }
  };
  if (nthreads == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < nthreads; ++t)
      th.emplace_back(work, t);
    for (auto &t : th)
      t.join();
  }
  if (counters)
    for (uint32_t t = 0; t < nthreads; ++t)
      add_counters(*counters, per[t]);
}

extern "C" uint64_t orc_gen_rays(const rt_scene_desc *scene,
                                 const rt_camera_uniforms *cam,
                                 const rt_render_params *params, uint32_t bounce,
                                 rt_ray *rays, uint64_t cap) {
  Capture cap_{rays, cap, 0, bounce};
  orc_counters cn;
  memset(&cn, 0, sizeof cn);
  uint32_t y0 = params->row_begin, y1 = params->row_end;
  if (y0 == 0 && y1 == 0)
    y1 = params->image_height;
  for (uint32_t y = y0; y < y1 && y < params->image_height; ++y)
    for (uint32_t x = 0; x < params->image_width; ++x)
      shade_pixel(scene, cam, params, x, y, nullptr, cn, &cap_);
  return cap_.count;
}
