// rt_kernels.hip — hand-written gfx950 kernels of the render path.
//
//   k_trace_closest : intersect_tlas over a ray batch        (TLAS.slang:14-67)
//   k_render_v0     : compute_main, one lane per pixel       (RayTracing.slang:62-181)
//   k_render_v1     : the same arithmetic per pixel, scheduled as persistent
//                     wave64s that regenerate paths and pull pixels from a queue
//
// Bit-exactness contract with the oracle: fp32, no FMA contraction (file-wide
// pragma + -ffp-contract=off), IEEE divide/sqrt (hipcc default), explicit
// compare-select min/max, the oracle's operation order everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"
#include "rt_device_layout.h"
#include "rt_kernels.h"

#pragma clang fp contract(off)

namespace {

#define RT_BLOCK 256

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// GLSL.std.450 FMin/FMax wording; NOT v_min_f32/v_max_f32 (those drop NaNs).
__device__ __forceinline__ float fmin_(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float fmax_(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ V3 normalize(V3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
__device__ __forceinline__ V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

// (c0*x + c1*y) + (c2*z + c3*w) — glm operator*(mat4, vec4), xyz rows
__device__ __forceinline__ V3 mul_cols(V3 c0, V3 c1, V3 c2, V3 c3, V3 v, float w) {
  return (c0 * v.x + c1 * v.y) + (c2 * v.z + c3 * w);
}

constexpr float kEpsilon = 1.192092896e-07f; // Ray.slang:4
constexpr float kMiss = 1e30f;               // BVHNode.slang:25

struct Hit {
  float t;
  uint32_t tri, inst;
  float u, v;
};

// Per-lane traversal stack: the first RT_LDS_STACK entries live in LDS, laid out
// [entry][lane] so a wave's accesses to one entry hit 64 consecutive banks;
// deeper entries (rare) go to per-lane scratch.
template <int BLOCK, int NLDS> struct StackT {
  uint32_t *lds; // &lds_stack[0][tid]
  uint32_t spill[RT_MAX_STACK - NLDS];
  int sp;
  __device__ __forceinline__ void push(uint32_t ref) {
    if (sp < NLDS)
      lds[sp * BLOCK] = ref;
    else
      spill[sp - NLDS] = ref;
    ++sp;
  }
  __device__ __forceinline__ uint32_t pop() {
    --sp;
    return sp < NLDS ? lds[sp * BLOCK] : spill[sp - NLDS];
  }
};
typedef StackT<RT_BLOCK, RT_LDS_STACK> Stack;

// intersect_aabb, BVHNode.slang:8-26 (DIV) / AABB.hpp:29-48 (RCP)
template <int MODE>
__device__ __forceinline__ float slab(V3 o, V3 d, V3 inv, float4 lo, float4 hi, float t) {
  float tx1, tx2, ty1, ty2, tz1, tz2;
  if (MODE == RT_AABB_DIV) {
    tx1 = (lo.x - o.x) / d.x;
    tx2 = (hi.x - o.x) / d.x;
    ty1 = (lo.y - o.y) / d.y;
    ty2 = (hi.y - o.y) / d.y;
    tz1 = (lo.z - o.z) / d.z;
    tz2 = (hi.z - o.z) / d.z;
  } else {
    tx1 = (lo.x - o.x) * inv.x;
    tx2 = (hi.x - o.x) * inv.x;
    ty1 = (lo.y - o.y) * inv.y;
    ty2 = (hi.y - o.y) * inv.y;
    tz1 = (lo.z - o.z) * inv.z;
    tz2 = (hi.z - o.z) * inv.z;
  }
  float tmin = fmin_(tx1, tx2);
  float tmax = fmax_(tx1, tx2);
  tmin = fmax_(tmin, fmin_(ty1, ty2));
  tmax = fmin_(tmax, fmax_(ty1, ty2));
  tmin = fmax_(tmin, fmin_(tz1, tz2));
  tmax = fmin_(tmax, fmax_(tz1, tz2));
  return (tmax >= tmin && tmin < t && tmax > 0) ? tmin : kMiss;
}

__device__ __forceinline__ V3 inv_direction(V3 d) {
  return v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

// intersect_tlas (TLAS.slang:14-67) with BLASInstance::intersect
// (BVHNode.slang:120-128) and BLAS::intersect (:43-109) folded into one loop
// over one reference stream: pair records for TLAS and BLAS interiors alike,
// an instance leaf switches the current ray to object space and plants a
// sentinel, the sentinel switches back.  The visit order, the near/far rule
// (swap iff dist1 > dist2), the untested roots and the inclusive t-range are
// the reference's.
template <int MODE>
__device__ bool trace(const rt_dev_scene &sc, V3 o, V3 d, float tmin, float tmax, Hit &hit,
                      Stack &st) {
  uint32_t ref = sc.root_ref;
  if (ref == RT_REF_NONE)
    return false;
  V3 co = o, cd = d;
  V3 cinv = MODE == RT_AABB_RCP ? inv_direction(d) : v3(0.f, 0.f, 0.f);
  float closest = tmax;
  bool any = false;
  uint32_t cur_inst = 0;
  st.sp = 0;
  while (true) {
    bool do_pop = false;
    if (!(ref & RT_REF_LEAF)) {
      const float4 *p = reinterpret_cast<const float4 *>(sc.pairs + ref);
      float4 lo0 = p[0], hi0 = p[1], lo1 = p[2], hi1 = p[3];
      float dist1 = slab<MODE>(co, cd, cinv, lo0, hi0, closest);
      float dist2 = slab<MODE>(co, cd, cinv, lo1, hi1, closest);
      uint32_t c1 = __float_as_uint(lo0.w), c2 = __float_as_uint(hi0.w);
      if (dist1 > dist2) {
        float tf = dist1;
        dist1 = dist2;
        dist2 = tf;
        uint32_t tc = c1;
        c1 = c2;
        c2 = tc;
      }
      if (dist1 == kMiss) {
        do_pop = true;
      } else {
        if (dist2 != kMiss)
          st.push(c2);
        ref = c1;
      }
    } else if (ref & RT_REF_INST) {
      cur_inst = ref & 0x3FFFFFFFu;
      const float4 *p = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
      float4 a = p[0], b = p[1], c = p[2], e = p[3];
      V3 c0 = v3(a.x, a.y, a.z), c1 = v3(b.x, b.y, b.z), c2 = v3(c.x, c.y, c.z),
         c3 = v3(e.x, e.y, e.z);
      co = mul_cols(c0, c1, c2, c3, o, 1.f);
      cd = mul_cols(c0, c1, c2, c3, d, 0.f);
      if (MODE == RT_AABB_RCP)
        cinv = inv_direction(cd);
      st.push(RT_REF_SENTINEL);
      ref = __float_as_uint(a.w);
    } else {
      // leaf: triangles are contiguous, the last one carries the terminator.
      uint32_t k = ref & 0x7FFFFFFFu;
      uint32_t last;
      do {
        const float4 *p = reinterpret_cast<const float4 *>(sc.tris + k);
        float4 t0 = p[0], t1 = p[1], t2 = p[2];
        last = __float_as_uint(t1.w);
        ++k;
        // TriangleGeom::hit, Triangle.slang:6-42
        V3 v0 = v3(t0.x, t0.y, t0.z), edge_1 = v3(t1.x, t1.y, t1.z), edge_2 = v3(t2.x, t2.y, t2.z);
        V3 h = cross(cd, edge_2);
        float a = dot(edge_1, h);
        if (a > -kEpsilon && a < kEpsilon)
          continue;
        float f = 1.f / a;
        V3 s = co - v0;
        float u = f * dot(s, h);
        if (u < 0 || u > 1)
          continue;
        V3 q = cross(s, edge_1);
        float v = f * dot(cd, q);
        if (v < 0 || u + v > 1)
          continue;
        float t = f * dot(edge_2, q);
        if (t < kEpsilon || !(tmin <= t && t <= closest))
          continue;
        closest = t;
        any = true;
        hit.t = t;
        hit.u = u;
        hit.v = v;
        hit.tri = __float_as_uint(t0.w);
        hit.inst = cur_inst;
      } while (!last);
      do_pop = true;
    }
    if (do_pop) {
      while (true) {
        if (st.sp == 0)
          return any;
        ref = st.pop();
        if (ref != RT_REF_SENTINEL)
          break;
        co = o;
        cd = d;
        if (MODE == RT_AABB_RCP)
          cinv = inv_direction(d);
      }
    }
  }
}

// ---- RNG (Random.slang) -----------------------------------------------------
__device__ __forceinline__ uint32_t wang_hash(uint32_t &seed) { // Random.slang:7-14
  seed = (seed ^ 61u) ^ (seed >> 16);
  seed *= 9u;
  seed = seed ^ (seed >> 4);
  seed *= 0x27d4eb2du;
  seed = seed ^ (seed >> 15);
  return seed;
}
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
struct Rng {
  uint32_t mode, seed, pixel, sample, slot, dim;
  __device__ __forceinline__ float next() {
    uint32_t h;
    if (mode == RT_RNG_WANG) {
      h = wang_hash(seed);
    } else {
      h = pcg_hash(pixel + pcg_hash(sample + pcg_hash(slot * 64u + dim)));
      ++dim;
    }
    return (float)h / 4294967296.f; // Random.slang:17-19
  }
  __device__ __forceinline__ void set_slot(uint32_t s) {
    slot = s;
    dim = 0;
  }
};

// Deterministic sin/cos shared bit for bit with the oracle (orc_sincos).
__device__ __forceinline__ void sincos_det(float a, float &s_out, float &c_out) {
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
  int q = k & 3;
  float ss = (q & 1) ? c : s;
  float cc = (q & 1) ? s : c;
  s_out = (q & 2) ? -ss : ss;
  c_out = (q == 1 || q == 2) ? -cc : cc;
}

__device__ __forceinline__ V3 rand_unit_vector(Rng &rng) { // Random.slang:34-41
  const float two_pi = 2.0f * 3.14159265359f;
  float z = -1.f + (1.f - -1.f) * rng.next();
  float a = rng.next() * two_pi;
  float r = sqrtf(1.f - z * z);
  float s, c;
  sincos_det(a, s, c);
  return v3(r * c, r * s, z);
}

__device__ __forceinline__ bool near_zero(V3 v) { // Random.slang:51-55
  const float s = 1e-8f;
  return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

__device__ __forceinline__ V3 rand_in_unit_disk(Rng &rng) { // Random.slang:57-63
  while (true) {
    float px = -1.f + (1.f - -1.f) * rng.next();
    float py = -1.f + (1.f - -1.f) * rng.next();
    V3 p = v3(px, py, 0.f);
    if (dot(p, p) < 1.f)
      return p;
  }
}

// nearest / repeat RGBA8-UNORM fetch (Renderer.cpp:347-361, Material.cpp:170)
__device__ __forceinline__ V3 sample_albedo(const rt_dev_scene &sc, uint32_t tex, float u, float v) {
  rt_texture_desc td = sc.textures[tex];
  float fu = floorf(u * (float)td.width);
  float fv = floorf(v * (float)td.height);
  int w = (int)td.width, h = (int)td.height;
  fu = fu - floorf(fu / (float)w) * (float)w;
  fv = fv - floorf(fv / (float)h) * (float)h;
  int ix = (fu == fu) ? (int)fu : 0;
  int iy = (fv == fv) ? (int)fv : 0;
  ix = ix < 0 ? 0 : (ix > w - 1 ? w - 1 : ix);
  iy = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
  uint32_t px = *reinterpret_cast<const uint32_t *>(sc.texels + td.offset +
                                                    4u * ((uint32_t)iy * td.width + (uint32_t)ix));
  return v3((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f,
            (float)((px >> 16) & 255u) / 255.0f);
}

__device__ __forceinline__ V3 reflect(V3 i, V3 n) { return i - (2.0f * dot(n, i)) * n; }

__device__ __forceinline__ float reflectance(float cosine, float ri) { // Material.slang:63-68
  float r0 = (1 - ri) / (1 + ri);
  r0 = r0 * r0;
  float x = 1 - cosine;
  float x2 = x * x;
  float x5 = (x2 * x2) * x;
  return r0 + (1 - r0) * x5;
}

__device__ __forceinline__ V3 rtiow_refract(V3 uv, V3 n, float eta) { // Material.slang:70-75
  float cos_theta = fmin_(dot(neg(uv), n), 1.0f);
  V3 perp = eta * (uv + cos_theta * n);
  V3 par = (-sqrtf(fabsf(1.0f - dot(perp, perp)))) * n;
  return perp + par;
}

// State of one path between two trace() calls.
struct Path {
  V3 o, d;           // current ray (world)
  V3 attenuation;    // RayTracing.slang:89
  V3 sample_radiance; // :90
};

// Camera ray of sample (s_i, s_j): RayTracing.slang:78-90 (+ defocus, Random.slang:65-70)
__device__ __forceinline__ void camera_ray(const rt_camera_uniforms &cam, float recip_sqrt_spp,
                                           uint32_t x, uint32_t y, uint32_t s_i, uint32_t s_j,
                                           Rng &rng, Path &p) {
  float jx = (((float)s_i + rng.next()) * recip_sqrt_spp) - 0.5f;
  float jy = (((float)s_j + rng.next()) * recip_sqrt_spp) - 0.5f;
  V3 pixel_sample = (ld3(cam.pixel00_loc) + (((float)(int)x + jx) * ld3(cam.pixel_delta_u))) +
                    (((float)(int)y + jy) * ld3(cam.pixel_delta_v));
  V3 origin = ld3(cam.camera_center);
  if (cam.defocus_disk_u[3] != 0.f) {
    V3 dk = rand_in_unit_disk(rng);
    origin = (origin + (dk.x * ld3(cam.defocus_disk_u))) + (dk.y * ld3(cam.defocus_disk_v));
  }
  p.o = origin;
  p.d = normalize(pixel_sample - origin);
  p.attenuation = v3(1.f, 1.f, 1.f);
  p.sample_radiance = v3(0.f, 0.f, 0.f);
}

// Everything after intersect_tlas in one bounce: RayTracing.slang:103-167.
// Returns true when the path continues (scattered).
__device__ __forceinline__ bool shade(const rt_dev_scene &sc, bool hit_anything, const Hit &hit,
                                      Rng &rng, Path &p) {
  if (!hit_anything) { // :159-167
    V3 unit = normalize(p.d);
    float a = 0.5f * (unit.y + 1.f);
    V3 sky = v3(0.7f, 0.7f, 0.7f) * (1.f - a) + v3(0.5f, 0.7f, 1.f) * a;
    p.sample_radiance = p.sample_radiance + p.attenuation * sky;
    return false;
  }
  const rt_blas_instance &inst = sc.instances[hit.inst];
  const float *T = inst.transform, *I = inst.inv_transform;
  // hit point in object space, then to world (Triangle.slang:34, RayTracing.slang:117)
  V3 i0 = ld3(I + 0), i1 = ld3(I + 4), i2 = ld3(I + 8), i3 = ld3(I + 12);
  V3 oo = mul_cols(i0, i1, i2, i3, p.o, 1.f);
  V3 od = mul_cols(i0, i1, i2, i3, p.d, 0.f);
  V3 p_obj = oo + hit.t * od;
  const rt_triangle_shading &sh = sc.shading[hit.tri];
  float alpha = 1.f - hit.u - hit.v;
  V3 local_normal =
      normalize((alpha * ld3(sh.n0) + hit.u * ld3(sh.n1)) + hit.v * ld3(sh.n2));
  // mul(transpose(inv), (n,0)): columns of the transpose are the rows of inv
  V3 r0 = v3(I[0], I[4], I[8]), r1 = v3(I[1], I[5], I[9]), r2 = v3(I[2], I[6], I[10]),
     r3 = v3(I[3], I[7], I[11]);
  V3 outward = normalize(mul_cols(r0, r1, r2, r3, local_normal, 0.0f));
  bool front_face = dot(p.d, outward) < 0.f; // HitRecord.slang:17-21
  V3 normal = front_face ? outward : neg(outward);
  float uvx = (alpha * sh.uv0[0] + hit.u * sh.uv1[0]) + hit.v * sh.uv2[0];
  float uvy = (alpha * sh.uv0[1] + hit.u * sh.uv1[1]) + hit.v * sh.uv2[1];
  V3 pw = mul_cols(ld3(T + 0), ld3(T + 4), ld3(T + 8), ld3(T + 12), p_obj, 1.f);

  uint32_t mtype = inst.material.type, mindex = inst.material.index;
  V3 emission = v3(0.f, 0.f, 0.f);
  V3 matt = v3(0.f, 0.f, 0.f);
  V3 out_d = p.d;
  bool scattered = false;
  if (mtype == RT_MATERIAL_LAMBERT) { // Material.slang:23-37
    V3 sd = normal + rand_unit_vector(rng);
    if (near_zero(sd))
      sd = normal;
    out_d = sd;
    matt = sample_albedo(sc, sc.lamberts[mindex], uvx, uvy);
    scattered = true;
  } else if (mtype == RT_MATERIAL_METAL) { // Material.slang:39-57
    float4 m = sc.metals[mindex];
    V3 reflected = reflect(p.d, normal);
    out_d = reflected + (m.w * rand_unit_vector(rng));
    matt = v3(m.x, m.y, m.z);
    scattered = true;
  } else if (mtype == RT_MATERIAL_DIELECTRIC) { // Material.slang:77-100
    float refraction_index = sc.dielectrics[mindex];
    matt = v3(1.f, 1.f, 1.f);
    float ri = front_face ? (1.f / refraction_index) : refraction_index;
    V3 unit = normalize(p.d);
    float cos_theta = fmin_(dot(neg(unit), normal), 1.f);
    float sin_theta = sqrtf(fmax_(0.f, 1.f - cos_theta * cos_theta));
    bool cannot_refract = ri * sin_theta > 1.f;
    if (cannot_refract || reflectance(cos_theta, ri) > rng.next())
      out_d = reflect(unit, normal);
    else
      out_d = rtiow_refract(unit, normal, ri);
    scattered = true;
  } else if (mtype == RT_MATERIAL_EMISSIVE) { // Material.slang:102-108
    float4 e = sc.emissives[mindex];
    emission = v3(e.x, e.y, e.z);
  }
  p.sample_radiance = p.sample_radiance + p.attenuation * emission; // :151
  if (!scattered)
    return false;
  p.attenuation = p.attenuation * matt;
  p.o = pw;
  p.d = out_d;
  return true;
}

__device__ __forceinline__ bool trace_mode(const rt_dev_scene &sc, uint32_t mode, V3 o, V3 d,
                                           float tmin, float tmax, Hit &hit, Stack &st) {
  return mode == RT_AABB_RCP ? trace<RT_AABB_RCP>(sc, o, d, tmin, tmax, hit, st)
                             : trace<RT_AABB_DIV>(sc, o, d, tmin, tmax, hit, st);
}

__device__ __forceinline__ void count_rays(unsigned long long *counter, uint32_t n) {
  // wave-wide sum, one atomic per wave
  for (int off = 32; off > 0; off >>= 1)
    n += __shfl_xor(n, off, 64);
  if ((threadIdx.x & 63) == 0 && n)
    atomicAdd(counter, (unsigned long long)n);
}

// Running mean of RayTracing.slang:174-179.
__device__ __forceinline__ void accumulate(float4 *accum, uint32_t width, uint32_t x, uint32_t y,
                                           V3 radiance, uint32_t sqrt_spp, uint32_t accum_count) {
  float pixel_sample_scale = 1.f / (float)(sqrt_spp * sqrt_spp);
  radiance = radiance * pixel_sample_scale;
  float4 *px = accum + ((size_t)y * width + x);
  float4 prev = *px;
  float fi = (float)accum_count, fi1 = (float)(accum_count + 1u);
  V3 acc = (v3(prev.x, prev.y, prev.z) * fi + radiance) / fi1;
  *px = make_float4(acc.x, acc.y, acc.z, 1.f);
}

__device__ __forceinline__ Rng make_rng(const rt_render_params &pc, uint32_t x, uint32_t y) {
  Rng rng;
  rng.mode = pc.rng_mode;
  rng.seed = (x * 1973u) ^ (y * 9277u) ^ (pc.frame_index * 26699u); // RayTracing.slang:71-72
  rng.pixel = y * pc.image_width + x;
  rng.sample = 0;
  rng.slot = 0;
  rng.dim = 0;
  return rng;
}

} // namespace

// ---------------------------------------------------------------------------
// k_trace_closest: one lane per ray.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_trace_closest(rt_dev_scene sc, const rt_ray *__restrict__ rays, uint64_t n,
                    rt_hit *__restrict__ hits, uint32_t aabb_mode,
                    unsigned long long *ray_counter) {
  __shared__ uint32_t lds_stack[RT_LDS_STACK * RT_BLOCK];
  Stack st;
  st.lds = lds_stack + threadIdx.x;
  uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  uint32_t traced = 0;
  if (i < n) {
    const float4 *rp = reinterpret_cast<const float4 *>(rays + i);
    float4 a = rp[0], b = rp[1];
    Hit hit;
    hit.t = b.w;
    hit.tri = hit.inst = RT_INVALID_ID;
    hit.u = hit.v = 0.f;
    bool any = trace_mode(sc, aabb_mode, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), a.w, b.w, hit, st);
    traced = 1;
    float4 *hp = reinterpret_cast<float4 *>(hits + i);
    hp[0] = make_float4(any ? hit.t : b.w, __uint_as_float(any ? hit.tri : RT_INVALID_ID),
                        __uint_as_float(any ? hit.inst : RT_INVALID_ID), any ? hit.u : 0.f);
    hp[1] = make_float4(any ? hit.v : 0.f, 0.f, 0.f, 0.f);
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_render_v0: the reference's structure — one lane per pixel, sample loop
// around a bounce loop.  A 256-thread block covers a 16x16 pixel tile, each of
// its four waves an 8x8 sub-tile (coherent primary rays per wave); tiles are
// dealt so that blocks sharing an XCD (blockIdx % 8) own one contiguous band of
// the image and therefore one region of the BVH in that XCD's L2.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcd_band_tile(uint32_t bid, uint32_t ntiles) {
  uint32_t xcd = bid & 7u, q = ntiles >> 3, r = ntiles & 7u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (bid >> 3);
}

extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_render_v0(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc,
                float4 *__restrict__ accum, unsigned long long *ray_counter) {
  __shared__ uint32_t lds_stack[RT_LDS_STACK * RT_BLOCK];
  Stack st;
  st.lds = lds_stack + threadIdx.x;
  const uint32_t tiles_x = (pc.image_width + 15u) >> 4;
  const uint32_t tile = xcd_band_tile(blockIdx.x, gridDim.x);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t x = (tile % tiles_x) * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t y = pc.row_begin + (tile / tiles_x) * 16u + (wave >> 1) * 8u + (lane >> 3);
  uint32_t traced = 0;
  if (x < pc.image_width && y < pc.row_end) {
    Rng rng = make_rng(pc, x, y);
    const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
    V3 radiance = v3(0.f, 0.f, 0.f);
    uint32_t sample_in_frame = 0;
    for (uint32_t s_j = 0; s_j < pc.sqrt_spp; ++s_j)
      for (uint32_t s_i = 0; s_i < pc.sqrt_spp; ++s_i, ++sample_in_frame) {
        rng.sample = pc.frame_index * (pc.sqrt_spp * pc.sqrt_spp) + sample_in_frame;
        rng.set_slot(0);
        Path p;
        camera_ray(cam, recip_sqrt_spp, x, y, s_i, s_j, rng, p);
        for (uint32_t d = 0; d < pc.max_depth; ++d) {
          rng.set_slot(d + 1);
          Hit hit;
          bool any = trace_mode(sc, pc.aabb_mode, p.o, p.d, 0.0001f, 1000.f, hit, st);
          ++traced;
          if (!shade(sc, any, hit, rng, p))
            break;
        }
        radiance = radiance + p.sample_radiance;
      }
    accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_render_v1: persistent wave64s with path regeneration.  Per pixel the
// arithmetic and the RNG draw order are those of k_render_v0 (so the image is
// the same bit for bit); what changes is the schedule:
//   * a lane whose path ended (sky, light, depth) starts its pixel's next
//     sample at once instead of idling until the longest path of the wave ends;
//   * a lane whose pixel is finished pulls the next pixel from a global queue
//     (one wave-aggregated atomic per refill), in 8x8-tile order so that the
//     lanes of a wave keep working on neighbouring pixels;
//   * the grid is sized to the chip (resident blocks only), every lane runs
//     until the queue is empty, every wave leaves when its 64 lanes are done.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_render_v1(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc,
                float4 *__restrict__ accum, unsigned long long *ray_counter,
                uint32_t *__restrict__ queue) {
  __shared__ uint32_t lds_stack[RT_LDS_STACK * RT_BLOCK];
  Stack st;
  st.lds = lds_stack + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rows = pc.row_end - pc.row_begin;
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint32_t total_slots = tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628

  bool alive = true, have_pixel = false, have_path = false;
  uint32_t x = 0, y = 0, sample = 0, depth = 0, traced = 0;
  Rng rng = make_rng(pc, 0, 0);
  V3 radiance = v3(0.f, 0.f, 0.f);
  Path p;
  p.o = p.d = p.attenuation = p.sample_radiance = v3(0.f, 0.f, 0.f);

  while (true) {
    // ---- refill: pixels for lanes that have none ----
    while (true) {
      const bool need = alive && !have_pixel;
      const unsigned long long mask = __ballot(need);
      if (mask == 0ull)
        break;
      const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
      uint32_t base = 0;
      if (lane == leader)
        base = atomicAdd(queue, (uint32_t)__popcll(mask));
      base = __shfl(base, (int)leader, 64);
      if (need) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot >= total_slots) {
          alive = false;
        } else {
          const uint32_t tile = slot >> 6, in = slot & 63u;
          x = (tile % tiles_x) * 8u + (in & 7u);
          y = pc.row_begin + (tile / tiles_x) * 8u + (in >> 3);
          if (x < pc.image_width && y < pc.row_end) { // else: edge padding, fetch again
            have_pixel = true;
            have_path = false;
            sample = 0;
            rng = make_rng(pc, x, y);
            radiance = v3(0.f, 0.f, 0.f);
          }
        }
      }
    }
    if (__ballot(alive) == 0ull)
      break;
    // ---- regenerate: next camera ray of the lane's pixel ----
    if (have_pixel && !have_path) {
      rng.sample = pc.frame_index * spp + sample;
      rng.set_slot(0);
      camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
      depth = 0;
      have_path = true;
    }
    // ---- one bounce for every lane that has a path ----
    bool path_ends = have_path;
    if (have_path && depth < pc.max_depth) {
      rng.set_slot(depth + 1);
      Hit hit;
      const bool any = trace_mode(sc, pc.aabb_mode, p.o, p.d, 0.0001f, 1000.f, hit, st);
      ++traced;
      const bool scattered = shade(sc, any, hit, rng, p);
      ++depth;
      path_ends = !scattered || depth >= pc.max_depth;
    }
    if (have_path && path_ends) {
      radiance = radiance + p.sample_radiance; // RayTracing.slang:170
      have_path = false;
      if (++sample == spp) {
        accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
        have_pixel = false;
      }
    }
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_render_v2: persistent wave64s, path regeneration AND resumable traversal.
// Profiling k_render_v1 (profiles/r01_v1_persistent) showed 19 % VALU lane
// utilisation: a wave waits for its longest ray while sky rays finish after a
// handful of nodes.  Here the traversal itself is a per-lane state machine
// (reference, stack pointer, object-space ray, closest hit live in registers /
// LDS between steps): the wave runs traversal steps until `regen_idle` lanes
// have finished their ray (or none is traversing any more); then it shades
// those lanes, regenerates their next ray (next bounce, next sample or
// next pixel from the queue) and resumes — the unfinished lanes simply carry
// on.  Per pixel the sequence of arithmetic and RNG draws is unchanged.
// ---------------------------------------------------------------------------
enum : uint32_t { ST_FETCH = 0, ST_NEWPATH = 1, ST_TRAVERSE = 2, ST_SHADE = 3, ST_DONE = 4 };

//
// NCACHE > 0 (variant 3): the first NCACHE pair records of the scene — the top
// of the TLAS and of the BLASes, which rt_upload_scene numbers first in
// breadth-first order — are copied into LDS once per workgroup; every ray
// walks through them, so about half of all node visits are served by
// ds_read_b128 (LDS latency, no TA/TD/L1 traffic) instead of four divergent
// 16-byte global gathers.  One 1024-thread workgroup per CU then owns the
// whole 160 KiB of LDS: 16 stack entries per lane + 1536 cached records.
template <int MODE, int MINW, int BLOCK, int NLDS, int NCACHE>
__global__ void __launch_bounds__(BLOCK, MINW)
    k_render_v2(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc,
                float4 *__restrict__ accum, unsigned long long *ray_counter,
                uint32_t *__restrict__ queue, uint32_t regen_idle) {
  extern __shared__ __align__(16) uint32_t lds_raw[];
  uint32_t *lds_stack = lds_raw;                                                  // [NLDS][BLOCK]
  const float4 *lds_cache = reinterpret_cast<const float4 *>(lds_raw + NLDS * BLOCK); // [NCACHE][4]
  uint32_t ncache = 0;
  if (NCACHE > 0) {
    ncache = sc.cache_pairs < (uint32_t)NCACHE ? sc.cache_pairs : (uint32_t)NCACHE;
    float4 *dst = reinterpret_cast<float4 *>(lds_raw + NLDS * BLOCK);
    const float4 *src = reinterpret_cast<const float4 *>(sc.pairs);
    for (uint32_t i = threadIdx.x; i < ncache * 4u; i += BLOCK)
      dst[i] = src[i];
    __syncthreads();
  }
  StackT<BLOCK, NLDS> st;
  st.lds = lds_stack + threadIdx.x;
  st.sp = 0;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rows = pc.row_end - pc.row_begin;
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint32_t total_slots = tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628

  uint32_t state = ST_FETCH;
  uint32_t x = 0, y = 0, sample = 0, depth = 0, traced = 0;
  Rng rng = make_rng(pc, 0, 0);
  V3 radiance = v3(0.f, 0.f, 0.f);
  Path p;
  p.o = p.d = p.attenuation = p.sample_radiance = v3(0.f, 0.f, 0.f);
  // traversal state of the lane's current ray
  uint32_t ref = RT_REF_NONE, cur_inst = 0;
  V3 co = v3(0.f, 0.f, 0.f), cd = v3(0.f, 0.f, 0.f), cinv = v3(0.f, 0.f, 0.f);
  float closest = 0.f;
  bool any = false;
  Hit hit;
  hit.t = 0.f, hit.u = hit.v = 0.f, hit.tri = hit.inst = 0;

  while (true) {
    // ---- (1) lanes whose ray is finished: shade, then continue / end the path ----
    if (state == ST_SHADE) {
      const bool scattered = shade(sc, any, hit, rng, p);
      ++depth;
      if (scattered && depth < pc.max_depth) {
        state = ST_NEWPATH + 100u; // marker: start the next bounce below
      } else {
        radiance = radiance + p.sample_radiance; // RayTracing.slang:170
        if (++sample == spp) {
          accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
          state = ST_FETCH;
        } else {
          state = ST_NEWPATH;
        }
      }
    }
    // ---- (2) lanes without a pixel: pull from the queue (wave-uniform loop) ----
    while (true) {
      const bool need = state == ST_FETCH;
      const unsigned long long mask = __ballot(need);
      if (mask == 0ull)
        break;
      const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
      uint32_t base = 0;
      if (lane == leader)
        base = atomicAdd(queue, (uint32_t)__popcll(mask));
      base = __shfl(base, (int)leader, 64);
      if (need) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot >= total_slots) {
          state = ST_DONE;
        } else {
          const uint32_t tile = slot >> 6, in = slot & 63u;
          x = (tile % tiles_x) * 8u + (in & 7u);
          y = pc.row_begin + (tile / tiles_x) * 8u + (in >> 3);
          if (x < pc.image_width && y < pc.row_end) { // else edge padding: fetch again
            state = ST_NEWPATH;
            sample = 0;
            rng = make_rng(pc, x, y);
            radiance = v3(0.f, 0.f, 0.f);
          }
        }
      }
    }
    if (__ballot(state != ST_DONE) == 0ull)
      break;
    // ---- (3) new camera ray / next bounce: arm the traversal ----
    if (state == ST_NEWPATH) {
      rng.sample = pc.frame_index * spp + sample;
      rng.set_slot(0);
      camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
      depth = 0;
      if (pc.max_depth == 0) { // degenerate: the sample contributes nothing (RayTracing.slang:92)
        if (++sample == spp) {
          accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
          state = ST_FETCH;
        }
      } else {
        state = ST_NEWPATH + 100u;
      }
    }
    if (state == ST_NEWPATH + 100u) {
      rng.set_slot(depth + 1);
      ++traced;
      ref = sc.root_ref;
      co = p.o, cd = p.d;
      if (MODE == RT_AABB_RCP)
        cinv = inv_direction(p.d);
      closest = 1000.f; // RayTracing.slang:93-95
      any = false;
      cur_inst = 0;
      st.sp = 0;
      state = ref == RT_REF_NONE ? ST_SHADE : ST_TRAVERSE;
    }
    // ---- (4) traversal steps while enough lanes are traversing ----
    do {
      if (state == ST_TRAVERSE) {
        bool do_pop = false;
        if (!(ref & RT_REF_LEAF)) {
          float4 lo0, hi0, lo1, hi1;
          if (NCACHE > 0 && ref < ncache) {
            const float4 *q = lds_cache + ref * 4u;
            lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
          } else {
            const float4 *q = reinterpret_cast<const float4 *>(sc.pairs + ref);
            lo0 = q[0], hi0 = q[1], lo1 = q[2], hi1 = q[3];
          }
          float dist1 = slab<MODE>(co, cd, cinv, lo0, hi0, closest);
          float dist2 = slab<MODE>(co, cd, cinv, lo1, hi1, closest);
          uint32_t c1 = __float_as_uint(lo0.w), c2 = __float_as_uint(hi0.w);
          if (dist1 > dist2) {
            float tf = dist1;
            dist1 = dist2;
            dist2 = tf;
            uint32_t tc = c1;
            c1 = c2;
            c2 = tc;
          }
          if (dist1 == kMiss) {
            do_pop = true;
          } else {
            if (dist2 != kMiss)
              st.push(c2);
            ref = c1;
          }
        } else if (ref & RT_REF_INST) {
          cur_inst = ref & 0x3FFFFFFFu;
          const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
          float4 a = q[0], b = q[1], c = q[2], e = q[3];
          V3 c0 = v3(a.x, a.y, a.z), c1 = v3(b.x, b.y, b.z), c2 = v3(c.x, c.y, c.z),
             c3 = v3(e.x, e.y, e.z);
          co = mul_cols(c0, c1, c2, c3, p.o, 1.f);
          cd = mul_cols(c0, c1, c2, c3, p.d, 0.f);
          if (MODE == RT_AABB_RCP)
            cinv = inv_direction(cd);
          st.push(RT_REF_SENTINEL);
          ref = __float_as_uint(a.w);
        } else {
          uint32_t k = ref & 0x7FFFFFFFu;
          uint32_t last;
          do {
            const float4 *q = reinterpret_cast<const float4 *>(sc.tris + k);
            float4 t0 = q[0], t1 = q[1], t2 = q[2];
            last = __float_as_uint(t1.w);
            ++k;
            V3 v0 = v3(t0.x, t0.y, t0.z), edge_1 = v3(t1.x, t1.y, t1.z),
               edge_2 = v3(t2.x, t2.y, t2.z);
            V3 h = cross(cd, edge_2);
            float a = dot(edge_1, h);
            if (a > -kEpsilon && a < kEpsilon)
              continue;
            float f = 1.f / a;
            V3 s = co - v0;
            float u = f * dot(s, h);
            if (u < 0 || u > 1)
              continue;
            V3 qq = cross(s, edge_1);
            float v = f * dot(cd, qq);
            if (v < 0 || u + v > 1)
              continue;
            float t = f * dot(edge_2, qq);
            if (t < kEpsilon || !(0.0001f <= t && t <= closest))
              continue;
            closest = t;
            any = true;
            hit.t = t;
            hit.u = u;
            hit.v = v;
            hit.tri = __float_as_uint(t0.w);
            hit.inst = cur_inst;
          } while (!last);
          do_pop = true;
        }
        if (do_pop) {
          while (true) {
            if (st.sp == 0) {
              state = ST_SHADE;
              break;
            }
            ref = st.pop();
            if (ref != RT_REF_SENTINEL)
              break;
            co = p.o;
            cd = p.d;
            if (MODE == RT_AABB_RCP)
              cinv = inv_direction(p.d);
          }
        }
      }
    } while (__ballot(state == ST_TRAVERSE) != 0ull &&
             (uint32_t)__popcll(__ballot(state == ST_SHADE)) < regen_idle);
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// Launchers (host).  Shapes are validated by the caller (rt_hip_api.cpp).
// ---------------------------------------------------------------------------
hipError_t rt_launch_trace(const rt_dev_scene &sc, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits,
                           uint32_t aabb_mode, unsigned long long *d_counter, hipStream_t stream) {
  if (n == 0)
    return hipSuccess;
  uint64_t blocks = (n + RT_BLOCK - 1) / RT_BLOCK;
  if (blocks > 0x7FFFFFFFull)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_trace_closest, dim3((uint32_t)blocks), dim3(RT_BLOCK), 0, stream, sc, d_rays, n,
                     d_hits, aabb_mode, d_counter);
  return hipGetLastError();
}

hipError_t rt_launch_render(uint32_t variant, const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                            const rt_render_params &pc, float *d_accum,
                            unsigned long long *d_counter, uint32_t *d_queue, uint32_t regen_idle,
                            hipStream_t stream) {
  uint32_t rows = pc.row_end - pc.row_begin;
  if (rows == 0 || pc.image_width == 0)
    return hipSuccess;
  if ((variant & 0xFFu) == 0) {
    uint32_t tiles = ((pc.image_width + 15u) >> 4) * ((rows + 15u) >> 4);
    hipLaunchKernelGGL(k_render_v0, dim3(tiles), dim3(RT_BLOCK), 0, stream, sc, cam, pc,
                       reinterpret_cast<float4 *>(d_accum), d_counter);
    return hipGetLastError();
  }
  // persistent grid: only as many blocks as the chip holds at once
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess)
      e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess)
      return e;
  }
  const uint64_t slots = (uint64_t)((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  auto blocks_for = [&](uint32_t waves_per_simd, uint32_t block) { // 4 SIMDs per CU
    uint64_t resident = (uint64_t)cus * (waves_per_simd * 4u / (block / 64u));
    uint64_t need = (slots + block - 1) / block;
    return (uint32_t)(need < resident ? need : resident);
  };
  uint32_t blocks = blocks_for(4, RT_BLOCK);
  hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess)
    return e;
  if ((variant & 0xFFu) == 1) {
    hipLaunchKernelGGL(k_render_v1, dim3(blocks), dim3(RT_BLOCK), 0, stream, sc, cam, pc,
                       reinterpret_cast<float4 *>(d_accum), d_counter, d_queue);
  } else if ((variant & 0xFFu) == 2) {
#define RT_V2(MODE, W)                                                                              \
  hipLaunchKernelGGL((k_render_v2<MODE, W, RT_BLOCK, RT_LDS_STACK, 0>), dim3(blocks), dim3(RT_BLOCK),  \
                     RT_LDS_STACK * RT_BLOCK * 4, stream, sc, cam, pc,                                \
                     reinterpret_cast<float4 *>(d_accum), d_counter, d_queue, regen_idle)
    const uint32_t w = variant >> 16; // experiment knob: min waves per SIMD the kernel is compiled for
    blocks = blocks_for(w == 0 ? 4 : w, RT_BLOCK);
    if (pc.aabb_mode == RT_AABB_RCP) {
      if (w == 5) RT_V2(RT_AABB_RCP, 5);
      else if (w == 6) RT_V2(RT_AABB_RCP, 6);
      else RT_V2(RT_AABB_RCP, 4);
    } else {
      if (w == 5) RT_V2(RT_AABB_DIV, 5);
      else if (w == 6) RT_V2(RT_AABB_DIV, 6);
      else RT_V2(RT_AABB_DIV, 4);
    }
#undef RT_V2
  } else {
    // variant 3: one 1024-thread workgroup per CU owning all 160 KiB of LDS
    constexpr int B3 = 1024, NL3 = 16, NC3 = 1536;
    constexpr size_t lds3 = (size_t)NL3 * B3 * 4 + (size_t)NC3 * 64;
    static_assert(lds3 <= 160 * 1024, "variant 3 LDS budget");
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t ea = hipFuncSetAttribute(
          reinterpret_cast<const void *>(&k_render_v2<RT_AABB_DIV, 4, B3, NL3, NC3>),
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (ea == hipSuccess)
        ea = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&k_render_v2<RT_AABB_RCP, 4, B3, NL3, NC3>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
      if (ea != hipSuccess)
        return ea;
      attr_set = true;
    }
    blocks = blocks_for(4, B3);
    if (pc.aabb_mode == RT_AABB_RCP)
      hipLaunchKernelGGL((k_render_v2<RT_AABB_RCP, 4, B3, NL3, NC3>), dim3(blocks), dim3(B3), lds3, stream,
                         sc, cam, pc, reinterpret_cast<float4 *>(d_accum), d_counter, d_queue, regen_idle);
    else
      hipLaunchKernelGGL((k_render_v2<RT_AABB_DIV, 4, B3, NL3, NC3>), dim3(blocks), dim3(B3), lds3, stream,
                         sc, cam, pc, reinterpret_cast<float4 *>(d_accum), d_counter, d_queue, regen_idle);
  }
  return hipGetLastError();
}
