// rt_render_v7.hip — k_render_v7: the vote-scheduled persistent traversal of
// k_render_v2<SCHED=2> (rt_kernels.hip) rebuilt for occupancy.
//
// Measured on k_render_v2 (profiles/): the traversal is bound by the
// latency of dependent node gathers (L2 miss -> Infinity Cache, about 1 us per
// wave step under load), and throughput grows with every resident wave
// (1 / 2 / 4 waves per SIMD: 1061 / 1885 / 2821 Mrays/s).  k_render_v2 holds 123
// VGPRs, i.e. 4 waves per SIMD, because every lane carries its whole path in
// registers while it traverses.  Here a lane keeps in registers only what a
// traversal step reads (reference, stack top, object-space ray, reciprocals,
// closest t); everything only shading / regeneration touches — path origin and
// direction, attenuation, radiance sums, RNG state, pixel, sample and bounce
// counters, the barycentrics and ids of the closest hit — lives in a
// structure-of-arrays block in HBM (`cold`, [wave][field][lane], fully coalesced) that is
// read when a lane finishes a ray and written when its next ray is armed.  The
// instance-entry and sentinel steps, which need the world-space ray, fetch it
// from there alongside the instance record they load anyway.
//
// Arithmetic, visit order and RNG draws per pixel are those of every other
// variant: the image is bit-identical.
//
// RESULT (C3, 64 spp): the 4-wave build runs at 2517 Mrays/s against 2820 for
// k_render_v2<SCHED=1> (the cold-block traffic costs 10 %), and the builds for 5 / 6 / 8
// waves per SIMD at 2139 / 1758 / 1328: shade() alone needs about 105 VGPRs, so under
// a tighter register limit the compiler spills inside the regeneration phase and the
// extra waves do not pay for it.  Kept as a measured negative result (variant 7); the
// next step on this line is a leaner shade() (staged loads) or a separate shading kernel.
#include "rt_device_inl.h"

namespace {

enum : uint32_t { ST_FETCH = 0, ST_NEWPATH = 1, ST_TRAVERSE = 2, ST_SHADE = 3, ST_DONE = 4, ST_ARM = 5 };

// fields of the cold block
enum : uint32_t {
  // first the fields the traversal loop touches: byte offsets (f * 256) that fit the 12-bit
  // immediate of global_load/store, so they cost no address registers in the loop
  CF_OX = 0, CF_OY, CF_OZ, CF_DX, CF_DY, CF_DZ,       // path ray (world)
  CF_HU, CF_HV, CF_HTRI, CF_HINST,                    // closest hit (t is `closest`)
  CF_AX, CF_AY, CF_AZ,                                // attenuation
  CF_SEED, CF_XY, CF_SAMPLE,                          // RNG state, pixel, sample counter
  CF_DEPTH,                                           // bounce counter
  CF_SX, CF_SY, CF_SZ,                                // sample_radiance
  CF_RX, CF_RY, CF_RZ,                                // radiance (sum over the pixel's samples)
  CF_COUNT
};

} // namespace



template <int MODE, int MINW, int NLDS>
__global__ void __launch_bounds__(RT_BLOCK, MINW)
    k_render_v7(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc, float4 *__restrict__ accum,
                unsigned long long *ray_counter, uint32_t *__restrict__ queue, uint32_t *__restrict__ cold,
                uint32_t regen_idle, uint32_t sched_arg) {
  extern __shared__ __align__(16) uint32_t lds_raw[];
  RT_STACK_DECL(RT_BLOCK, NLDS, st, lds_raw);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rows = rt_virtual_rows(pc);
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint32_t total_slots = tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
  const bool scene_ok = (sc.flags & RT_SCENE_BOXES_MODERATE) != 0u && !(sched_arg & 0x8000u);
  const uint32_t w_pair = (sched_arg & 0xFu) ? (sched_arg & 0xFu) : 1u;
  const uint32_t w_leaf = ((sched_arg >> 4) & 0xFu) ? ((sched_arg >> 4) & 0xFu) : 3u;
  const uint32_t w_inst = ((sched_arg >> 8) & 0xFu) ? ((sched_arg >> 8) & 0xFu) : 6u;

  // cold block, [wave][field][lane]: field f of this lane at my[f * 64] — a constant byte offset
  // (f * 256) from one per-lane pointer, so no per-field address lives in registers, and each
  // access of a wave covers two whole 128-byte lines
  uint32_t *const my = cold + ((size_t)(blockIdx.x * (RT_BLOCK / 64u) + (threadIdx.x >> 6)) * CF_COUNT * 64u + lane);
#define CLD_U(f) (my[(f) * 64u])
#define CLD_F(f) __uint_as_float(my[(f) * 64u])
#define CST_U(f, v) (my[(f) * 64u] = (v))
#define CST_F(f, v) (my[(f) * 64u] = __float_as_uint(v))

  // ---- hot state: what a traversal step reads ----
  uint32_t state = ST_FETCH;
  uint32_t ref = RT_REF_NONE, top = RT_REF_NONE, cur_inst = 0;
  V3 co = v3(0.f, 0.f, 0.f), cd = v3(0.f, 0.f, 0.f), cinv = v3(0.f, 0.f, 0.f);
  bool fast = false, any = false;
  float closest = 0.f;
  uint32_t wave_traced = 0; // rays armed by this lane (summed per wave at the end)

  while (true) {
    // ===== (A) regeneration: lanes whose ray is finished, or that have no pixel =====
    if (state == ST_SHADE || state == ST_FETCH) {
      Path p;
      p.o = p.d = p.attenuation = p.sample_radiance = v3(0.f, 0.f, 0.f);
      V3 radiance = v3(0.f, 0.f, 0.f);
      uint32_t seed = 0, x = 0, y = 0, sample = 0, depth = 0;
      if (state == ST_SHADE) {
        p.o = v3(CLD_F(CF_OX), CLD_F(CF_OY), CLD_F(CF_OZ));
        p.d = v3(CLD_F(CF_DX), CLD_F(CF_DY), CLD_F(CF_DZ));
        p.attenuation = v3(CLD_F(CF_AX), CLD_F(CF_AY), CLD_F(CF_AZ));
        p.sample_radiance = v3(CLD_F(CF_SX), CLD_F(CF_SY), CLD_F(CF_SZ));
        radiance = v3(CLD_F(CF_RX), CLD_F(CF_RY), CLD_F(CF_RZ));
        seed = CLD_U(CF_SEED);
        const uint32_t xy = CLD_U(CF_XY);
        x = xy & 0xFFFFu, y = xy >> 16;
        sample = CLD_U(CF_SAMPLE);
        depth = CLD_U(CF_DEPTH);
        Hit hit;
        hit.t = closest;
        hit.u = hit.v = 0.f, hit.tri = hit.inst = 0;
        if (any) {
          hit.u = CLD_F(CF_HU), hit.v = CLD_F(CF_HV);
          hit.tri = CLD_U(CF_HTRI), hit.inst = CLD_U(CF_HINST);
        }
        Rng rng; // as left by the arm step of this bounce: slot depth+1, dim 0
        rng.mode = pc.rng_mode, rng.seed = seed, rng.pixel = y * pc.image_width + x;
        rng.sample = pc.frame_index * spp + sample;
        rng.set_slot(depth + 1);
        const bool scattered = shade(sc, any, hit, rng, p);
        seed = rng.seed;
        ++depth;
        if (scattered && depth < pc.max_depth) {
          state = ST_ARM;
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
      // lanes without a pixel pull from the queue (uniform over the lanes in this branch)
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
        if (base >= total_slots) { // the queue is exhausted (uniform over the lanes in this loop)
          if (need)
            state = ST_DONE;
          break;
        }
        if (need) {
          const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
          if (slot < total_slots) {
            const uint32_t tile = slot >> 6, in = slot & 63u;
            x = (tile % tiles_x) * 8u + (in & 7u);
            y = rt_row(pc, (tile / tiles_x) * 8u + (in >> 3));
            if (x < pc.image_width && y < pc.row_end) { // else edge padding: fetch again
              sample = 0;
              seed = make_rng(pc, x, y).seed;
              radiance = v3(0.f, 0.f, 0.f);
              if (pc.max_depth == 0) { // degenerate: no sample contributes (RayTracing.slang:92)
                accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
              } else {
                state = ST_NEWPATH;
              }
            }
          } else {
            state = ST_DONE;
          }
        }
      }
      if (state == ST_NEWPATH) {
        Rng rng;
        rng.mode = pc.rng_mode, rng.seed = seed, rng.pixel = y * pc.image_width + x;
        rng.sample = pc.frame_index * spp + sample;
        rng.set_slot(0);
        camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
        seed = rng.seed;
        depth = 0;
        state = ST_ARM;
      }
      if (state == ST_ARM) {
        ++wave_traced;
        ref = sc.root_ref;
        top = RT_REF_NONE;
        st.sp = 0;
        co = p.o, cd = p.d, prep_space<MODE>(co, cd, scene_ok, cinv, fast);
        closest = 1000.f; // RayTracing.slang:93-95
        any = false;
        cur_inst = 0;
        state = ref == RT_REF_NONE ? ST_SHADE : ST_TRAVERSE;
        CST_F(CF_OX, p.o.x), CST_F(CF_OY, p.o.y), CST_F(CF_OZ, p.o.z);
        CST_F(CF_DX, p.d.x), CST_F(CF_DY, p.d.y), CST_F(CF_DZ, p.d.z);
        CST_F(CF_AX, p.attenuation.x), CST_F(CF_AY, p.attenuation.y), CST_F(CF_AZ, p.attenuation.z);
        CST_F(CF_SX, p.sample_radiance.x), CST_F(CF_SY, p.sample_radiance.y), CST_F(CF_SZ, p.sample_radiance.z);
        CST_F(CF_RX, radiance.x), CST_F(CF_RY, radiance.y), CST_F(CF_RZ, radiance.z);
        CST_U(CF_SEED, seed);
        CST_U(CF_XY, x | (y << 16));
        CST_U(CF_SAMPLE, sample);
        CST_U(CF_DEPTH, depth);
      }
    }
    const unsigned long long live = __ballot(state != ST_DONE);
    if (live == 0ull)
      break;
    const uint32_t n_done = 64u - (uint32_t)__popcll(live);

    // ===== (B) traversal steps, one kind per wave iteration (vote) =====
    while (true) {
      const uint32_t kind =
          state != ST_TRAVERSE ? 0u : !(ref & RT_REF_LEAF) ? 1u : (ref & RT_REF_INST) ? 2u : 3u;
      const uint32_t n1 = (uint32_t)__popcll(__ballot(kind == 1u));
      const uint32_t n2 = (uint32_t)__popcll(__ballot(kind == 2u));
      const uint32_t n3 = (uint32_t)__popcll(__ballot(kind == 3u));
      if (n1 + n2 + n3 == 0u || 64u - (n1 + n2 + n3) - n_done >= regen_idle)
        break;
      const uint32_t s1 = n1 * w_pair, s2 = n2 * w_inst, s3 = n3 * w_leaf;
      if (s1 >= s2 && s1 >= s3) {
        if (kind == 1u) { // interior pair: both child boxes, near child next, far child pushed
          const float4 *q = reinterpret_cast<const float4 *>(sc.pairs + ref);
          const float4 qx = q[0], qy = q[1], qz = q[2];
          const float2 r2 = *reinterpret_cast<const float2 *>(q + 3);
          float dist1, dist2;
          if (fast) {
            slab_pair_fast<MODE>(co, cd, cinv, qx, qy, qz, closest, dist1, dist2);
          } else { // the reference's formulation, for rays / scenes outside the fast path's preconditions
            float4 lo0, hi0, lo1, hi1;
            unpack_pair(qx, qy, qz, make_float4(r2.x, r2.y, 0.f, 0.f), lo0, hi0, lo1, hi1);
            dist1 = slab<MODE>(co, cd, cinv, lo0, hi0, closest);
            dist2 = slab<MODE>(co, cd, cinv, lo1, hi1, closest);
          }
          uint32_t c1 = __float_as_uint(r2.x), c2 = __float_as_uint(r2.y);
          if (dist1 > dist2) {
            float tf = dist1;
            dist1 = dist2;
            dist2 = tf;
            uint32_t tc = c1;
            c1 = c2;
            c2 = tc;
          }
          if (dist1 == kMiss) {
            ref = st.pop_top(top);
            if (ref == RT_REF_NONE)
              state = ST_SHADE;
          } else {
            if (dist2 != kMiss)
              st.push_top(top, c2);
            ref = c1;
          }
        }
      } else if (s3 >= s2) {
        if (kind == 3u) { // one triangle of the leaf: TriangleGeom::hit, Triangle.slang:6-42
          const uint32_t k = ref & 0x7FFFFFFFu;
          const float4 *q = reinterpret_cast<const float4 *>(sc.tris + k);
          const float4 t0 = q[0], t1 = q[1], t2 = q[2];
          const bool last = __float_as_uint(t1.w) != 0u;
          const V3 v0 = v3(t0.x, t0.y, t0.z), edge_1 = v3(t1.x, t1.y, t1.z), edge_2 = v3(t2.x, t2.y, t2.z);
          const V3 h = cross(cd, edge_2);
          const float a = dot(edge_1, h);
          if (!(a > -kEpsilon && a < kEpsilon)) {
            const float f = 1.f / a;
            const V3 s = co - v0;
            const float u = f * dot(s, h);
            if (!(u < 0 || u > 1)) {
              const V3 qq = cross(s, edge_1);
              const float v = f * dot(cd, qq);
              if (!(v < 0 || u + v > 1)) {
                const float t = f * dot(edge_2, qq);
                if (!(t < kEpsilon || !(0.0001f <= t && t <= closest))) {
                  closest = t;
                  any = true;
                  CST_F(CF_HU, u), CST_F(CF_HV, v);
                  CST_U(CF_HTRI, __float_as_uint(t0.w)), CST_U(CF_HINST, cur_inst);
                }
              }
            }
          }
          if (last) {
            ref = st.pop_top(top);
            if (ref == RT_REF_NONE)
              state = ST_SHADE;
          } else {
            ref = ref + 1u;
          }
        }
      } else {
        if (kind == 2u) { // change of ray space: enter an instance, or (sentinel) return to world space
          const V3 po = v3(CLD_F(CF_OX), CLD_F(CF_OY), CLD_F(CF_OZ));
          const V3 pd = v3(CLD_F(CF_DX), CLD_F(CF_DY), CLD_F(CF_DZ));
          if (ref == RT_REF_SENTINEL) {
            co = po, cd = pd, prep_space<MODE>(co, cd, scene_ok, cinv, fast);
            ref = st.pop_top(top);
            if (ref == RT_REF_NONE)
              state = ST_SHADE;
          } else {
            cur_inst = ref & 0x3FFFFFFFu;
            const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
            const float4 a = q[0], b = q[1], c = q[2], e = q[3];
            const V3 c0 = v3(a.x, a.y, a.z), c1 = v3(b.x, b.y, b.z), c2 = v3(c.x, c.y, c.z), c3 = v3(e.x, e.y, e.z);
            co = mul_cols(c0, c1, c2, c3, po, 1.f), cd = mul_cols(c0, c1, c2, c3, pd, 0.f);
            prep_space<MODE>(co, cd, scene_ok, cinv, fast);
            st.push_top(top, RT_REF_SENTINEL);
            ref = __float_as_uint(a.w);
          }
        }
      }
    }
  }
  count_rays(ray_counter, wave_traced);
#undef CLD_U
#undef CLD_F
#undef CST_U
#undef CST_F
}

// waves_per_simd: 4, 6 or 8 (the build the launch uses); returns the number of lanes the
// launch will run, i.e. the size of the cold block in units of rt_v7_cold_fields() dwords.
static uint32_t v7_blocks(uint32_t waves_per_simd, uint32_t cus, uint64_t slots) {
  const uint64_t resident = (uint64_t)cus * waves_per_simd; // 256-thread blocks: one wave per SIMD each
  const uint64_t need = (slots + RT_BLOCK - 1) / RT_BLOCK;
  return (uint32_t)(need < resident ? need : resident);
}

size_t rt_v7_cold_bytes(uint32_t cus) { return (size_t)cus * 8u * RT_BLOCK * CF_COUNT * sizeof(uint32_t); }

hipError_t rt_launch_render_v7(uint32_t variant, const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                               const rt_render_params &pc, float *d_accum, unsigned long long *d_counter,
                               uint32_t *d_queue, uint32_t *d_cold, uint32_t cus, uint32_t regen_idle,
                               hipStream_t stream) {
  const uint32_t rows = rt_virtual_rows(pc);
  const uint64_t slots = (uint64_t)((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  if (pc.image_width > 0xFFFFu || pc.image_height > 0xFFFFu)
    return hipErrorInvalidValue; // pixel coordinates are packed 16:16 in the cold block
  const uint32_t sched = ((variant >> 16) & 0xFFFu) | ((variant >> 31) ? 0x8000u : 0u);
  const uint32_t occ = (variant >> 28) & 7u; // 0: 4 waves per SIMD (no spills), 2: the 6-wave build (80 VGPRs, spills)
  const uint32_t w = occ == 2 ? 6u : 4u;
  const uint32_t blocks = v7_blocks(w, cus, slots);
  hipError_t e = hipMemsetAsync(d_queue, 0, 64 * sizeof(uint32_t), stream);
  if (e != hipSuccess)
    return e;
  float4 *acc = reinterpret_cast<float4 *>(d_accum);
  const bool rcp = pc.aabb_mode == RT_AABB_RCP;
#define RT_V7(MODE, W, NL)                                                                          \
  hipLaunchKernelGGL((k_render_v7<MODE, W, NL>), dim3(blocks), dim3(RT_BLOCK), (NL) * RT_BLOCK * 4, \
                     stream, sc, cam, pc, acc, d_counter, d_queue, d_cold, regen_idle, sched)
  if (w == 6) {
    if (rcp) RT_V7(RT_AABB_RCP, 6, 24); else RT_V7(RT_AABB_DIV, 6, 24);
  } else {
    if (rcp) RT_V7(RT_AABB_RCP, 4, 24); else RT_V7(RT_AABB_DIV, 4, 24);
  }
#undef RT_V7
  return hipGetLastError();
}
