// rt_device_inl.h — device-side building blocks shared by the kernels of
// rt_kernels.hip (megakernels) and rt_wavefront.hip (wavefront pipeline): fp32
// vector helpers, the slab / triangle tests, the folded TLAS->BLAS traversal,
// RNG, materials and the per-bounce shading of RayTracing.slang.
//
// Bit-exactness contract with the oracle: fp32, no FMA contraction (file-wide
// pragma + -ffp-contract=off), IEEE divide/sqrt (hipcc default), explicit
// compare-select min/max, the oracle's operation order everywhere.
#ifndef RT_DEVICE_INL_H
#define RT_DEVICE_INL_H

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
// The spill array is declared by the kernel (RT_STACK_DECL) and only pointed to
// from here: with the array inside this struct the whole struct — `sp` included —
// is demoted to scratch memory and every push/pop pays scratch loads and stores.
typedef __attribute__((address_space(3))) uint32_t *lds_u32_ptr; // typed LDS pointer: ds_read/ds_write, never flat
template <int BLOCK, int NLDS> struct StackT {
  lds_u32_ptr lds; // &lds_stack[0][tid]
  uint32_t *spill; // per-lane scratch, RT_MAX_STACK - NLDS entries
  int sp;
  __device__ __forceinline__ void push(uint32_t ref) {
    if (sp < NLDS)
      lds[sp * BLOCK] = ref;
    else
      spill[sp - NLDS] = ref;
    ++sp;
  }
  __device__ __forceinline__ uint32_t pop() {
    // An unconditional ds_read plus a rare scratch branch: `c ? lds[..] : spill[..]`
    // becomes a select of two pointers and a flat_load on every pop.
    --sp;
    uint32_t r = lds[(sp < NLDS ? sp : NLDS - 1) * BLOCK];
    if (sp >= NLDS)
      r = spill[sp - NLDS];
    return r;
  }
  // Variant with the top entry cached in a register (`top`; RT_REF_NONE = bottom marker):
  // a push is one ds_write, a pop is a register move plus a ds_read whose result is
  // not needed before the next push/pop, so its latency hides behind the next step.
  __device__ __forceinline__ void push_top(uint32_t &top, uint32_t ref) {
    if (sp < NLDS)
      lds[sp * BLOCK] = top;
    else
      spill[sp - NLDS] = top;
    ++sp;
    top = ref;
  }
  __device__ __forceinline__ uint32_t pop_top(uint32_t &top) { // RT_REF_NONE when the stack is empty
    const uint32_t r = top;
    if (r != RT_REF_NONE) { // the marker sits below every pushed entry, so sp >= 1 here
      --sp;
      uint32_t t = lds[(sp < NLDS ? sp : NLDS - 1) * BLOCK];
      if (sp >= NLDS)
        t = spill[sp - NLDS];
      top = t;
    }
    return r;
  }
};
typedef StackT<RT_BLOCK, RT_LDS_STACK> Stack;
#define RT_STACK_DECL(BLOCK, NLDS, name, lds_base)                                                 \
  uint32_t name##_spill[RT_MAX_STACK - (NLDS)];                                                    \
  StackT<BLOCK, NLDS> name;                                                                        \
  name.lds = (lds_u32_ptr)((lds_base) + threadIdx.x);                                                             \
  name.spill = name##_spill;                                                                       \
  name.sp = 0

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

// One pair record (rt_dev_pair: x[4], y[4], z[4], refs) as the two child boxes
// of the reference's formulation: lo/hi of child 0 and 1, child refs in lo0.w / hi0.w.
__device__ __forceinline__ void unpack_pair(float4 qx, float4 qy, float4 qz, float4 qr, float4 &lo0,
                                            float4 &hi0, float4 &lo1, float4 &hi1) {
  lo0 = make_float4(qx.x, qy.x, qz.x, qr.x);
  hi0 = make_float4(qx.y, qy.y, qz.y, qr.y);
  lo1 = make_float4(qx.z, qy.z, qz.z, 0.f);
  hi1 = make_float4(qx.w, qy.w, qz.w, 0.f);
}

// ---- fast slab test for one pair record, bit-identical to slab<MODE> x2 -------
// DIV mode.  hipcc lowers an IEEE fp32 `n / d` to
//   v_div_scale x2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup
// which is correctly rounded.  For operands of moderate magnitude div_scale,
// div_fmas' scaling and div_fixup are the identity, and what is left is
//   r1 = fma(fma(-d, rcp(d), 1), rcp(d), rcp(d))            (depends on d only)
//   q0 = n*r1; e0 = fma(-d,q0,n); q1 = fma(e0,r1,q0); e1 = fma(-d,q1,n); q = fma(e1,r1,q1)
// of which q1 is already the correctly rounded quotient (see div2_exact).
// The ray's three r1 are computed once per ray space (prep_space), the twelve
// quotients of a pair record then cost 3 packed-fp32 instructions per two.
// Preconditions, checked per ray space (fast_axis_ok) and per scene
// (RT_SCENE_BOXES_MODERATE); a lane that fails them takes slab<MODE>:
//   2^-30 <= |d| <= 2^30; o == 0 or 2^-66 <= |o| <= 2^59; box coordinate 0 or in
//   [2^-66, 2^59].  Then every numerator n = b - o is 0 or 2^-90 <= |n| <= 2^60 (a
//   nonzero difference of two fp32 numbers is at least the smaller ulp), no
//   intermediate leaves the normal range or loses bits to underflow (fp32
//   denormals are enabled), and no quotient is NaN — so v_min/v_max/v_min3/v_max3
//   return what the compare-select min/max of slab<MODE> return, up to the sign
//   of a zero, which only ever feeds comparisons.
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float refined_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ bool fast_axis_ok(float o, float d) {
  const float ad = fabsf(d), ao = fabsf(o);
  return ad >= 0x1p-30f && ad <= 0x1p30f && (o == 0.0f || (ao >= 0x1p-66f && ao <= 0x1p59f));
}
// RCP mode: inv = 1/d is finite and the numerators are finite, so no product is NaN
__device__ __forceinline__ bool fast_axis_ok_rcp(float o, float inv) {
  return fabsf(inv) <= 3.0e38f && fabsf(o) <= 0x1p59f;
}
// ONE correction round is enough (round 2): q = fma(fma(-d, q0, n), r1, q0) with q0 = n * r1 equals the IEEE quotient
// for EVERY pair of fp32 mantissas — checked exhaustively on the GPU, all 2^23 x 2^23 pairs (7.04e13, 35 s;
// experiments/r02_div/div_exhaustive.hip, profiles/r02_div_exhaustive/), together with the exact scaling of r1 over
// the exponent range of the preconditions — so the compiler's second round (e1, q2) is dropped: 3 instead of 5
// packed instructions per two quotients, 18 instead of 30 per pair record.
__device__ __forceinline__ f2v div2_exact(f2v n, float d, float r) {
  const f2v nd = {-d, -d}, rr = {r, r};
  const f2v q0 = n * rr;
#ifdef RT_EXPERIMENT_APPROX_DIV // TIMING EXPERIMENT ONLY (not bit-exact): the upper bound of what a filtered slab test could save
  (void)nd;
  return q0;
#else
  const f2v e0 = __builtin_elementwise_fma(nd, q0, n);
  return __builtin_elementwise_fma(e0, rr, q0);
#endif
}
// Per ray space (world, or an instance's object space): the reciprocals the slab
// test uses and whether slab_pair_fast may be used.  DIV: inv = refined
// reciprocals r1 (meaningful when fast); RCP: inv = 1/d.  Plain values, not a
// struct: a struct handed around by reference ends up in scratch memory.
template <int MODE>
__device__ __forceinline__ void prep_space(V3 o, V3 d, bool scene_ok, V3 &inv, bool &fast) {
  if (MODE == RT_AABB_RCP) {
    inv = inv_direction(d);
    fast = scene_ok && fast_axis_ok_rcp(o.x, inv.x) && fast_axis_ok_rcp(o.y, inv.y) &&
           fast_axis_ok_rcp(o.z, inv.z);
  } else {
    inv = v3(refined_rcp(d.x), refined_rcp(d.y), refined_rcp(d.z));
    fast = scene_ok && fast_axis_ok(o.x, d.x) && fast_axis_ok(o.y, d.y) && fast_axis_ok(o.z, d.z);
  }
}
template <int MODE>
__device__ __forceinline__ void slab_pair_fast(V3 o, V3 d, V3 inv, float4 qx, float4 qy, float4 qz,
                                               float t, float &dist0, float &dist1) {
  const f2v ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
  const f2v nx0 = f2v{qx.x, qx.y} - ox, nx1 = f2v{qx.z, qx.w} - ox;
  const f2v ny0 = f2v{qy.x, qy.y} - oy, ny1 = f2v{qy.z, qy.w} - oy;
  const f2v nz0 = f2v{qz.x, qz.y} - oz, nz1 = f2v{qz.z, qz.w} - oz;
  f2v tx0, tx1, ty0, ty1, tz0, tz1;
  if (MODE == RT_AABB_RCP) {
    const f2v ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    tx0 = nx0 * ix, tx1 = nx1 * ix;
    ty0 = ny0 * iy, ty1 = ny1 * iy;
    tz0 = nz0 * iz, tz1 = nz1 * iz;
  } else {
    tx0 = div2_exact(nx0, d.x, inv.x), tx1 = div2_exact(nx1, d.x, inv.x);
    ty0 = div2_exact(ny0, d.y, inv.y), ty1 = div2_exact(ny1, d.y, inv.y);
    tz0 = div2_exact(nz0, d.z, inv.z), tz1 = div2_exact(nz1, d.z, inv.z);
  }
  const float tmin0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx0.x, tx0.y), __builtin_fminf(ty0.x, ty0.y)),
                                      __builtin_fminf(tz0.x, tz0.y));
  const float tmax0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx0.x, tx0.y), __builtin_fmaxf(ty0.x, ty0.y)),
                                      __builtin_fmaxf(tz0.x, tz0.y));
  const float tmin1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx1.x, tx1.y), __builtin_fminf(ty1.x, ty1.y)),
                                      __builtin_fminf(tz1.x, tz1.y));
  const float tmax1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx1.x, tx1.y), __builtin_fmaxf(ty1.x, ty1.y)),
                                      __builtin_fmaxf(tz1.x, tz1.y));
  dist0 = (tmax0 >= tmin0 && tmin0 < t && tmax0 > 0) ? tmin0 : kMiss;
  dist1 = (tmax1 >= tmin1 && tmin1 < t && tmax1 > 0) ? tmin1 : kMiss;
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
      float4 lo0, hi0, lo1, hi1;
      unpack_pair(p[0], p[1], p[2], p[3], lo0, hi0, lo1, hi1);
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
__device__ __forceinline__ V3 sample_albedo(const rt_dev_scene &sc, uint32_t tex_offset, uint32_t tex_w,
                                            uint32_t tex_h, float u, float v) {
  float fu = floorf(u * (float)tex_w);
  float fv = floorf(v * (float)tex_h);
  int w = (int)tex_w, h = (int)tex_h;
  fu = fu - floorf(fu / (float)w) * (float)w;
  fv = fv - floorf(fv / (float)h) * (float)h;
  int ix = (fu == fu) ? (int)fu : 0;
  int iy = (fv == fv) ? (int)fv : 0;
  ix = ix < 0 ? 0 : (ix > w - 1 ? w - 1 : ix);
  iy = iy < 0 ? 0 : (iy > h - 1 ? h - 1 : iy);
  uint32_t px = *reinterpret_cast<const uint32_t *>(sc.texels + tex_offset +
                                                    4u * ((uint32_t)iy * tex_w + (uint32_t)ix));
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
__device__ __forceinline__ bool shade(const rt_dev_scene &sc, uint32_t flags, bool hit_anything, const Hit &hit,
                                      Rng &rng, Path &p) {
  if (!hit_anything) { // :159-167
    V3 unit = normalize(p.d);
    float a = 0.5f * (unit.y + 1.f);
    const float hz = (flags & RT_RENDER_SKY_WHITE_HORIZON) ? 1.0f : 0.7f; // rt_types.h: the sky behind Screenshot3.png
    V3 sky = v3(hz, hz, hz) * (1.f - a) + v3(0.5f, 0.7f, 1.f) * a;
    p.sample_radiance = p.sample_radiance + p.attenuation * sky;
    return false;
  }
  const rt_dev_inst_shade &inst = sc.inst_shade[hit.inst]; // matrices + resolved material: one record
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
  V3 pw = mul_cols(ld3(T + 0), ld3(T + 4), ld3(T + 8), ld3(T + 12), p_obj, 1.f);

  const uint32_t mtype = inst.mtype;
  const float4 mparam = make_float4(inst.param[0], inst.param[1], inst.param[2], inst.param[3]);
  V3 emission = v3(0.f, 0.f, 0.f);
  V3 matt = v3(0.f, 0.f, 0.f);
  V3 out_d = p.d;
  bool scattered = false;
  if (mtype == RT_MATERIAL_LAMBERT) { // Material.slang:23-37
    V3 sd = normal + rand_unit_vector(rng);
    if (near_zero(sd))
      sd = normal;
    out_d = sd;
    // a 1x1 texture returns its texel for every uv (NaN included): decoded at upload into param, and the triangle's
    // uvs (32 of the record's 80 bytes) are not even fetched
    if (inst.tex_w == 1u && inst.tex_h == 1u) {
      matt = v3(mparam.x, mparam.y, mparam.z);
    } else {
      const float uvx = (alpha * sh.uv0[0] + hit.u * sh.uv1[0]) + hit.v * sh.uv2[0];
      const float uvy = (alpha * sh.uv0[1] + hit.u * sh.uv1[1]) + hit.v * sh.uv2[1];
      matt = sample_albedo(sc, inst.tex_offset, inst.tex_w, inst.tex_h, uvx, uvy);
    }
    scattered = true;
  } else if (mtype == RT_MATERIAL_METAL) { // Material.slang:39-57
    const float4 m = mparam;
    V3 reflected = reflect(p.d, normal);
    out_d = reflected + (m.w * rand_unit_vector(rng));
    matt = v3(m.x, m.y, m.z);
    scattered = true;
  } else if (mtype == RT_MATERIAL_DIELECTRIC) { // Material.slang:77-100
    float refraction_index = mparam.x;
    matt = v3(1.f, 1.f, 1.f);
    float ri = front_face ? (1.f / refraction_index) : refraction_index;
    V3 unit = normalize(p.d);
    float cos_theta = fmin_(dot(neg(unit), normal), 1.f);
    float sin_theta = sqrtf(fmax_(0.f, 1.f - cos_theta * cos_theta));
    bool cannot_refract = ri * sin_theta > 1.f;
    bool reflects; // Material.slang:91, both readings (rt_types.h, RT_RENDER_DIELECTRIC_ALWAYS_DRAWS)
    if (flags & RT_RENDER_DIELECTRIC_ALWAYS_DRAWS) {
      const float xi = rng.next();
      reflects = cannot_refract || reflectance(cos_theta, ri) > xi;
    } else {
      reflects = cannot_refract || reflectance(cos_theta, ri) > rng.next();
    }
    if (reflects)
      out_d = reflect(unit, normal);
    else
      out_d = rtiow_refract(unit, normal, ri);
    scattered = true;
  } else if (mtype == RT_MATERIAL_EMISSIVE) { // Material.slang:102-108
    const float4 e = mparam;
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

#endif
