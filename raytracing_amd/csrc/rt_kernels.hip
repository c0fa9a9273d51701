// rt_kernels.hip — hand-written gfx950 kernels of the render path.
//
//   k_trace_closest : intersect_tlas over a ray batch        (TLAS.slang:14-67)
//   k_render_v0     : compute_main, one lane per pixel       (RayTracing.slang:62-181)
//   k_render_v8     : the same arithmetic per pixel, scheduled as persistent wave64s with path regeneration, a
//                     resumable per-lane traversal, vote scheduling and joined pair / leaf steps (the default)
//
// The kernels measured and dropped in round 1 (variants 1-7 and 10: persistent without resumable traversal, LDS
// node cache, wavefront pipeline, register-top stack, HBM path state, pooled scheduling) live in
// experiments/r01_variants/ and are not part of the library.
//
// Bit-exactness contract with the oracle: fp32, no FMA contraction (file-wide
// pragma + -ffp-contract=off), IEEE divide/sqrt (hipcc default), explicit
// compare-select min/max, the oracle's operation order everywhere.
#include <atomic>

#include "rt_device_inl.h"

#pragma clang fp contract(off)


// ---------------------------------------------------------------------------
// k_trace_closest: one lane per ray.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_trace_closest(rt_dev_scene sc, const rt_ray *__restrict__ rays, uint64_t n,
                    rt_hit *__restrict__ hits, uint32_t aabb_mode,
                    unsigned long long *ray_counter) {
  __shared__ uint32_t lds_stack[RT_LDS_STACK * RT_BLOCK];
  RT_STACK_DECL(RT_BLOCK, RT_LDS_STACK, st, lds_stack);
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
  RT_STACK_DECL(RT_BLOCK, RT_LDS_STACK, st, lds_stack);
  const uint32_t tiles_x = (pc.image_width + 15u) >> 4;
  const uint32_t tile = xcd_band_tile(blockIdx.x, gridDim.x);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t x = (tile % tiles_x) * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t y = rt_row(pc, (tile / tiles_x) * 16u + (wave >> 1) * 8u + (lane >> 3));
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
          if (!shade(sc, pc.flags, any, hit, rng, p))
            break;
        }
        radiance = radiance + p.sample_radiance;
      }
    accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_render_v8: persistent wave64s, path regeneration, resumable traversal, vote scheduling with joined steps.
// A one-lane-per-pixel kernel waits for its longest ray while sky rays finish after a handful of nodes (13 %
// VALU lane utilisation, profiles/r01_v0_baseline).  Here the grid is sized to the chip, pixels come from a global
// queue in 8x8 tiles, and the traversal itself is a per-lane state machine (reference, stack pointer, object-space
// ray, closest hit live in registers / LDS between steps): the wave runs traversal steps until `regen_idle` lanes
// have finished their ray (or none is traversing any more); then it shades those lanes, regenerates their next
// ray (next bounce, next sample or next pixel from the queue) and resumes — the unfinished lanes simply carry
// on.  Per pixel the sequence of arithmetic and RNG draws is unchanged.
// ---------------------------------------------------------------------------
#ifndef RT_V8_MINW
#define RT_V8_MINW 4 // resident waves per SIMD the default kernel is built for
#endif
#ifdef RT_V2_STATS
__device__ unsigned long long g_v2_stats[32];
#define V2_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define V2_ACC(slot, t0, t1) dbg_cyc[slot] += (t1) - (t0)
#define V2_CNT(slot, pred)                                                                         \
  do {                                                                                             \
    const unsigned long long m_ = __ballot(pred);                                                  \
    if (m_) {                                                                                      \
      ++dbg_exec[slot];                                                                            \
      dbg_lanes[slot] += (uint32_t)__popcll(m_);                                                   \
    }                                                                                              \
  } while (0)
#else
#define V2_T(var)
#define V2_ACC(slot, t0, t1)
#define V2_CNT(slot, pred)
#endif
enum : uint32_t { ST_FETCH = 0, ST_NEWPATH = 1, ST_TRAVERSE = 2, ST_SHADE = 3, ST_DONE = 4 };

template <int MODE, int MINW, int BLOCK, int NLDS>
__global__ void __launch_bounds__(BLOCK, MINW)
    k_render_v8(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc,
                float4 *__restrict__ accum, unsigned long long *ray_counter,
                uint32_t *__restrict__ queue, uint32_t regen_idle, uint32_t sched_arg) {
  extern __shared__ __align__(16) uint32_t lds_raw[];
  uint32_t *lds_stack = lds_raw;                                                  // [NLDS][BLOCK]
  typedef float f4v __attribute__((ext_vector_type(4)));
  RT_STACK_DECL(BLOCK, NLDS, st, lds_stack);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rows = rt_virtual_rows(pc);
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
  // the ray in the space it is being traversed in (prep_space)
  V3 co = v3(0.f, 0.f, 0.f), cd = v3(0.f, 0.f, 0.f), cinv = v3(0.f, 0.f, 0.f);
  bool fast = false;
  const bool scene_ok = (sc.flags & RT_SCENE_BOXES_MODERATE) != 0u && !(sched_arg & 0x8000u);
  float closest = 0.f;
  bool any = false;
  Hit hit;
  hit.t = 0.f, hit.u = hit.v = 0.f, hit.tri = hit.inst = 0;

#ifdef RT_V2_STATS
  // sections: 0 shade, 1 fetch, 2 newpath, 3 arm, 4 pair, 5 inst, 6 leaf(tri iterations), 7 pop
  unsigned long long dbg_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t dbg_exec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dbg_lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long dbg_start = __builtin_amdgcn_s_memtime();
#endif
  while (true) {
    // ---- (1) lanes whose ray is finished: shade, then continue / end the path ----
    V2_T(ta0);
    V2_CNT(0, state == ST_SHADE);
    if (state == ST_SHADE) {
      hit.t = closest; // == the accepted t whenever `any`; keeping it out of the traversal loop saves a register
      const bool scattered = shade(sc, pc.flags, any, hit, rng, p);
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
    V2_T(ta1);
    V2_ACC(0, ta0, ta1);
    V2_CNT(1, state == ST_FETCH);
    // ---- (2) lanes without a pixel: pull from the queue (wave-uniform loop) ----
    // One global queue of 8x8 tiles.  (Per-XCD queue bands, so that an XCD's L2 sees one region of the image,
    // measured 8 % slower twice in round 1: the bands finish at different times.)
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
      if (base >= total_slots) { // the queue is exhausted for everybody
        if (need)
          state = ST_DONE;
        break;
      }
      if (need) {
        const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (slot >= total_slots) {
          state = ST_DONE;
        } else {
          const uint32_t tile = slot >> 6, in = slot & 63u;
          x = (tile % tiles_x) * 8u + (in & 7u);
          y = rt_row(pc, (tile / tiles_x) * 8u + (in >> 3));
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
    V2_T(ta2);
    V2_ACC(1, ta1, ta2);
    V2_CNT(2, state == ST_NEWPATH);
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
    V2_T(ta3);
    V2_ACC(2, ta2, ta3);
    V2_CNT(3, state == ST_NEWPATH + 100u);
    if (state == ST_NEWPATH + 100u) {
      rng.set_slot(depth + 1);
      ++traced;
      ref = sc.root_ref;
      co = p.o, cd = p.d, prep_space<MODE>(co, cd, scene_ok, cinv, fast);
      closest = 1000.f; // RayTracing.slang:93-95
      any = false;
      cur_inst = 0;
      st.sp = 0;
      state = ref == RT_REF_NONE ? ST_SHADE : ST_TRAVERSE;
    }
    V2_T(ta4);
    V2_ACC(3, ta3, ta4);
    const uint32_t n_done = (uint32_t)__popcll(__ballot(state == ST_DONE));
    (void)n_done;
    // ---- (4) traversal steps while enough lanes are traversing ----
    {
      // Vote scheduling: each wave iteration runs ONE kind of step — the one most lanes are waiting for (weighted) —
      // so the instruction stream of an iteration is one section at good lane occupancy instead of every section
      // at poor occupancy.  A BLAS leaf advances one triangle per step.  Pair and leaf steps are JOINED: with
      // one-triangle leaves (the reference's SAH builder makes almost only those) a leaf iteration of its own is one
      // memory round trip for a dozen lanes, so an iteration that the vote gives to the pair lanes also
      // serves the leaf lanes when at least `join_min` of them wait (and the other way round): the loads of
      // both kinds are issued together, so the wave pays one round trip, then the triangle section and the
      // pair section run one after the other.  Per lane the sequence of steps is unchanged.
      // Default weights pair : leaf : instance = 1 : 2 : 3 and join threshold 8 lanes, measured (tools/sweep_v8.sh,
      // sweep_v8b.sh, compare_v8.sh).
      const uint32_t w_pair = (sched_arg & 0xFu) ? (sched_arg & 0xFu) : 1u;
      const uint32_t w_leaf = ((sched_arg >> 4) & 0xFu) ? ((sched_arg >> 4) & 0xFu) : 2u;
      const uint32_t w_inst = ((sched_arg >> 8) & 0xFu) ? ((sched_arg >> 8) & 0xFu) : 3u;
      const uint32_t join_code = (sched_arg >> 12) & 7u;
      const uint32_t join_min = join_code == 0u ? 8u : join_code == 1u ? 1u : join_code == 2u ? 4u : join_code == 3u ? 6u
                              : join_code == 4u ? 12u : join_code == 5u ? 16u : join_code == 6u ? 24u : 65u;
      // `any` and `fast` as per-lane integers inside the step loop: as bools they live in scalar lane masks, and
      // every conditional update costs three scalar instructions where it merges, plus three per loop iteration
      uint32_t anyv = any ? 1u : 0u, fastv = fast ? 1u : 0u;
      while (true) {
        // 1 pair, 2 instance, 3 leaf from the two top bits of the reference (00 -> 1, 10 -> 3, 11 -> 2) by a packed
        // table, 0 when not traversing: no branch
        const uint32_t top2 = ref >> 30;
        const uint32_t kind = (state == ST_TRAVERSE ? 3u : 0u) & (0xB5u >> (top2 * 2u)); // table 01 -> 1 as well
        const uint32_t n1 = (uint32_t)__popcll(__ballot(kind == 1u));
        const uint32_t n2 = (uint32_t)__popcll(__ballot(kind == 2u));
        const uint32_t n3 = (uint32_t)__popcll(__ballot(kind == 3u));
        if (n1 + n2 + n3 == 0u || 64u - (n1 + n2 + n3) - n_done >= regen_idle)
          break;
        const uint32_t s1 = n1 * w_pair, s2 = n2 * w_inst, s3 = n3 * w_leaf;
        bool do_pop = false;
        // which kinds run this iteration, as ONE wave-uniform integer (bit k = kind k); the lanes derive their
        // own predicate from it with three vector instructions.  (Uniform bools combined with per-lane
        // conditions become chains of 64-bit scalar mask instructions: about 30 per iteration, measured.)
        uint32_t sel;
        if (s1 >= s2 && s1 >= s3)
          sel = n3 >= join_min ? 10u : 2u;
        else if (s3 >= s2)
          sel = n1 >= join_min ? 10u : 8u;
        else
          sel = 4u;
        const uint32_t act = ((sel >> kind) & 1u) ? kind : 0u; // this lane's step in this iteration, 0 = wait
        if (sel != 4u) {
          const bool do1 = act == 1u, do3 = act == 3u;
          // One address expression and ONE group of loads for both kinds: the leaf lanes fetch their 48-byte
          // triangle into the registers the pair lanes fetch their record into.  The loads are inline assembly
          // that narrows EXEC itself, so for the compiler they define the registers for every lane: written as
          // C++ loads under `if (do1 || do3)`, the lanes that do not fetch merge an undefined value with the
          // loaded one where the branches join, and the compiler then keeps a second set of 14 registers and
          // copies it every iteration (measured: -13 %).  The compiler does not count these loads: each
          // consuming section begins with its own s_waitcnt vmcnt(0) (loads return in order, so the compiler's
          // counts for its own loads stay sufficient), and whenever a load was issued for a lane that lane enters
          // one of the two sections below in this same iteration, so nothing is left in flight at the loop's end.
          // Lanes outside m13 keep stale register contents, which no section reads for them.
          f4v qx, qy, qz;
          f2v qr;
          {
            // index * stride + base, stride and base selected per lane (no branches)
            const uint32_t stride = kind == 3u ? 48u : 64u;
            const uint64_t base = kind == 3u ? (uint64_t)(uintptr_t)sc.tris : (uint64_t)(uintptr_t)sc.pairs;
            const char *q = reinterpret_cast<const char *>(sc.pairs) +
                            ((uint64_t)(ref & 0x7FFFFFFFu) * stride + (base - (uint64_t)(uintptr_t)sc.pairs));
            const unsigned long long m1 = __ballot(do1), m13 = __ballot(act != 0u);
            unsigned long long saved;
            asm volatile("; RT_V8_LOADS_BEGIN (tools/check_v8_asm.py)\n\t"
                         "s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b64 exec, %[m13]\n\t"
                         "global_load_dwordx4 %[a], %[q], off\n\t"
                         "global_load_dwordx4 %[b], %[q], off offset:16\n\t"
                         "global_load_dwordx4 %[c], %[q], off offset:32\n\t"
                         "s_mov_b64 exec, %[m1]\n\t"
                         "global_load_dwordx2 %[d], %[q], off offset:48\n\t"
                         "s_mov_b64 exec, %[sv]\n\t"
                         "; RT_V8_LOADS_END"
                         : [a] "=&v"(qx), [b] "=&v"(qy), [c] "=&v"(qz), [d] "=&v"(qr), [sv] "=&s"(saved)
                         : [q] "v"(q), [m13] "s"(m13), [m1] "s"(m1)
                         : "memory");
          }
          if (do3) { // one triangle of the leaf: TriangleGeom::hit, Triangle.slang:6-42
            asm volatile("s_waitcnt vmcnt(0) ; RT_V8_WAIT_TRI" : "+v"(qx), "+v"(qy), "+v"(qz));
            if (__float_as_uint(qy.w))
              do_pop = true; // last triangle of the leaf
            else
              ref = ref + 1u;
            // the test without its early exits (same operations on the accepted path, so the same bits): four
            // nested branches cost about 20 scalar instructions and make every value they may change a merge
            V3 v0 = v3(qx.x, qx.y, qx.z), edge_1 = v3(qy.x, qy.y, qy.z), edge_2 = v3(qz.x, qz.y, qz.z);
            V3 h = cross(cd, edge_2);
            float a = dot(edge_1, h);
            float f = 1.f / a;
            V3 s = co - v0;
            float u = f * dot(s, h);
            V3 qq = cross(s, edge_1);
            float v = f * dot(cd, qq);
            float t = f * dot(edge_2, qq);
            const bool accept = !(a > -kEpsilon && a < kEpsilon) & !(u < 0 || u > 1) & !(v < 0 || u + v > 1) &
                                !(t < kEpsilon || !(0.0001f <= t && t <= closest));
            closest = accept ? t : closest;
            anyv = accept ? 1u : anyv;
            hit.u = accept ? u : hit.u, hit.v = accept ? v : hit.v;
            hit.tri = accept ? __float_as_uint(qx.w) : hit.tri, hit.inst = accept ? cur_inst : hit.inst;
          }
          if (do1) {
            asm volatile("s_waitcnt vmcnt(0) ; RT_V8_WAIT_PAIR" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qr));
            const float4 px = make_float4(qx.x, qx.y, qx.z, qx.w), py = make_float4(qy.x, qy.y, qy.z, qy.w),
                         pz = make_float4(qz.x, qz.y, qz.z, qz.w), pr = make_float4(qr.x, qr.y, 0.f, 0.f);
            float dist1, dist2;
            if (fastv) {
              slab_pair_fast<MODE>(co, cd, cinv, px, py, pz, closest, dist1, dist2);
            } else { // the reference's formulation, for rays / scenes outside the fast path's preconditions
              float4 lo0, hi0, lo1, hi1;
              unpack_pair(px, py, pz, pr, lo0, hi0, lo1, hi1);
              dist1 = slab<MODE>(co, cd, cinv, lo0, hi0, closest);
              dist2 = slab<MODE>(co, cd, cinv, lo1, hi1, closest);
            }
            uint32_t c1 = __float_as_uint(qr.x), c2 = __float_as_uint(qr.y);
            if (dist1 > dist2) {
              float tf = dist1;
              dist1 = dist2;
              dist2 = tf;
              uint32_t tc = c1;
              c1 = c2;
              c2 = tc;
            }
            // branch-light form of "miss: pop; else descend into the near child and push the far one": nested
            // branches cost scalar instructions at every level, and the spill path of the stack is rare enough
            // for a wave-uniform test
            const bool hit1 = dist1 != kMiss, hit2 = dist2 != kMiss; // sorted: hit2 implies hit1
            do_pop = !hit1;
            ref = hit1 ? c1 : ref;
            if (__ballot(hit2 && st.sp >= NLDS) != 0ull) {
              if (hit2)
                st.push(c2);
            } else if (hit2) {
              st.lds[st.sp * BLOCK] = c2;
              ++st.sp;
            }
          }
        } else {
          if (act == 2u) {
            cur_inst = ref & 0x3FFFFFFFu;
            const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
            float4 a = q[0], b = q[1], c = q[2], e = q[3];
            V3 c0 = v3(a.x, a.y, a.z), c1 = v3(b.x, b.y, b.z), c2 = v3(c.x, c.y, c.z),
               c3 = v3(e.x, e.y, e.z);
            co = mul_cols(c0, c1, c2, c3, p.o, 1.f), cd = mul_cols(c0, c1, c2, c3, p.d, 0.f);
            bool f2;
            prep_space<MODE>(co, cd, scene_ok, cinv, f2);
            fastv = f2 ? 1u : 0u;
            st.push(RT_REF_SENTINEL);
            ref = __float_as_uint(a.w);
          }
        }
        if (do_pop) { // at most two pops: the entry under a sentinel is a TLAS reference, never another sentinel
          if (st.sp == 0) {
            state = ST_SHADE;
          } else {
            ref = st.pop();
            if (ref == RT_REF_SENTINEL) { // back to world space
              bool f2;
              co = p.o, cd = p.d, prep_space<MODE>(co, cd, scene_ok, cinv, f2);
              fastv = f2 ? 1u : 0u;
              if (st.sp == 0)
                state = ST_SHADE;
              else
                ref = st.pop();
            }
          }
        }
      }
      any = anyv != 0u, fast = fastv != 0u;
    }
  }
#ifdef RT_V2_STATS
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&g_v2_stats[i], dbg_cyc[i]);
      atomicAdd(&g_v2_stats[8 + i], (unsigned long long)dbg_exec[i]);
      atomicAdd(&g_v2_stats[16 + i], (unsigned long long)dbg_lanes[i]);
    }
    atomicAdd(&g_v2_stats[24], __builtin_amdgcn_s_memtime() - dbg_start);
    atomicAdd(&g_v2_stats[25], 1ull);
  }
#endif
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_probe_walk: measurement aid, not part of the render path.  Every lane walks
// the real pair records from the root, picking a pseudo-random child per step
// and restarting at the root when it reaches a leaf: the memory access pattern
// of the traversal (dependent, divergent 64-byte gathers) with almost no
// arithmetic.  Its step rate is the ceiling the memory path sets for the
// traversal kernels (roofline).  loads = 1..4 dwordx4 per step.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_probe_walk(rt_dev_scene sc, uint32_t steps, uint32_t loads, uint32_t start_ref,
                 unsigned long long *out) {
  uint32_t ref = start_ref;
  uint32_t h = (blockIdx.x * RT_BLOCK + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  unsigned long long walked = 0;
  // loads >> 8: number of lanes per wave that walk (0 = all 64); the others idle, as divergence would
  const uint32_t live = (loads >> 8) ? (loads >> 8) : 64u;
  loads &= 0xFFu;
  if ((threadIdx.x & 63u) >= live)
    steps = 0;
  if (loads == 5u) {
    // quad-cooperative gather: the four lanes of a quad fetch one lane's 64-byte record with ONE
    // dwordx4 instruction (16 B each, 64 contiguous bytes = one L1 access instead of four), four
    // instructions serve the quad's four records; the child refs (piece 3) are handed to the owner.
    // Measured 1.96x the step rate of per-lane gathers at full lane occupancy; inside the render
    // kernel (55 % of the lanes at a pair step, LDS-DMA staging) it was 8 % slower and was dropped.
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 3u, qbase = lane & ~3u;
    for (uint32_t i = 0; i < steps; ++i) {
      uint32_t rx = 0, ry = 0;
#pragma unroll
      for (uint32_t j = 0; j < 4u; ++j) {
        const uint32_t r = (uint32_t)__shfl((int)ref, (int)(qbase + j), 64);
        const float4 piece = *(reinterpret_cast<const float4 *>(sc.pairs + r) + sub);
        acc += piece.z;
        const uint32_t bx = (uint32_t)__shfl((int)__float_as_uint(piece.x), (int)(qbase + 3u), 64);
        const uint32_t by = (uint32_t)__shfl((int)__float_as_uint(piece.y), (int)(qbase + 3u), 64);
        if (sub == j)
          rx = bx, ry = by;
      }
      h = h * 1664525u + 1013904223u;
      uint32_t c = (h >> 16) & 1u ? ry : rx;
      if ((c & (RT_REF_LEAF | RT_REF_INST)) == (RT_REF_LEAF | RT_REF_INST) && c != RT_REF_SENTINEL && c != RT_REF_NONE)
        c = sc.inst_trace[c & 0x3FFFFFFFu].root_ref;
      ref = (c & RT_REF_LEAF) ? start_ref : c;
      ++walked;
    }
    steps = 0;
  }
  for (uint32_t i = 0; i < steps; ++i) {
    const float4 *q = reinterpret_cast<const float4 *>(sc.pairs + ref);
    float4 lo0 = q[0], hi0 = q[3]; // x quad and the refs quad
    acc += lo0.x;
    if (loads > 2) {
      float4 lo1 = q[1];
      acc += lo1.y;
    }
    if (loads > 3) {
      float4 hi1 = q[2];
      acc += hi1.z;
    }
    h = h * 1664525u + 1013904223u;
    uint32_t c = (h >> 16) & 1u ? __float_as_uint(hi0.y) : __float_as_uint(hi0.x);
    if ((c & (RT_REF_LEAF | RT_REF_INST)) == (RT_REF_LEAF | RT_REF_INST) && c != RT_REF_SENTINEL && c != RT_REF_NONE)
      c = sc.inst_trace[c & 0x3FFFFFFFu].root_ref; // descend into the instance's BLAS
    ref = (c & RT_REF_LEAF) ? start_ref : c;
    ++walked;
  }
  if (acc == 123.456f)
    walked += 1; // keep the loads alive
  for (int off = 32; off > 0; off >>= 1)
    walked += __shfl_xor(walked, off, 64);
  if ((threadIdx.x & 63) == 0)
    atomicAdd(out, walked);
}

// Test hook for the exact fast division of slab_pair_fast: q_fast[i] = div2_exact,
// q_ieee[i] = the compiler's IEEE n/d, for arbitrary operand pairs.
extern "C" __global__ void __launch_bounds__(RT_BLOCK)
    k_debug_fast_div(const float *__restrict__ n, const float *__restrict__ d, uint64_t count,
                     float *__restrict__ q_fast, float *__restrict__ q_ieee) {
  const uint64_t i = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (i >= count)
    return;
  const float nn = n[i], dd = d[i];
  const f2v q = div2_exact(f2v{nn, nn}, dd, refined_rcp(dd));
  q_fast[i] = q.y;
  q_ieee[i] = nn / dd;
}

hipError_t rt_launch_debug_fast_div(const float *d_n, const float *d_d, uint64_t count, float *d_fast,
                                    float *d_ieee, hipStream_t stream) {
  if (count == 0)
    return hipSuccess;
  hipLaunchKernelGGL(k_debug_fast_div, dim3((uint32_t)((count + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0,
                     stream, d_n, d_d, count, d_fast, d_ieee);
  return hipGetLastError();
}

hipError_t rt_launch_probe_walk(const rt_dev_scene &sc, uint32_t blocks, uint32_t steps, uint32_t loads,
                                unsigned long long *d_out, hipStream_t stream) {
  uint32_t start = sc.root_ref;
  if (start == RT_REF_NONE || (start & RT_REF_LEAF))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_probe_walk, dim3(blocks), dim3(RT_BLOCK), 0, stream, sc, steps, loads, start, d_out);
  return hipGetLastError();
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

// variant: low byte 0 = k_render_v0, 8 = k_render_v8; for 8, bits 16..27 = vote weights (pair, leaf, instance; 0 =
// default), bits 28..30 = join threshold code, bit 31 = never take the fast slab path (test aid).
hipError_t rt_launch_render(uint32_t variant, const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                            const rt_render_params &pc, float *d_accum,
                            unsigned long long *d_counter, uint32_t *d_queue, uint32_t regen_idle,
                            uint32_t cus, hipStream_t stream) {
  uint32_t rows = rt_virtual_rows(pc);
  if (rows == 0 || pc.image_width == 0)
    return hipSuccess;
  const uint32_t kernel = variant & 0xFFu;
  if (kernel == 0) {
    uint32_t tiles = ((pc.image_width + 15u) >> 4) * ((rows + 15u) >> 4);
    hipLaunchKernelGGL(k_render_v0, dim3(tiles), dim3(RT_BLOCK), 0, stream, sc, cam, pc,
                       reinterpret_cast<float4 *>(d_accum), d_counter);
    return hipGetLastError();
  }
  if (kernel != 8 || cus == 0)
    return hipErrorInvalidValue;
  // persistent grid: only as many blocks as the chip holds at once (cus = compute units of the device, 4 SIMDs each)
  const uint64_t slots = (uint64_t)((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  const uint64_t resident = (uint64_t)cus * (RT_V8_MINW * 4u / (RT_BLOCK / 64u));
  const uint64_t need = (slots + RT_BLOCK - 1) / RT_BLOCK;
  const uint32_t blocks = (uint32_t)(need < resident ? need : resident);
  const uint32_t sched = ((variant >> 16) & 0xFFFu) | ((variant >> 31) ? 0x8000u : 0u) | (((variant >> 28) & 7u) << 12);
  hipError_t e = hipMemsetAsync(d_queue, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess)
    return e;
  const size_t lds_bytes = RT_LDS_STACK * RT_BLOCK * 4;
  if (pc.aabb_mode == RT_AABB_RCP)
    hipLaunchKernelGGL((k_render_v8<RT_AABB_RCP, RT_V8_MINW, RT_BLOCK, RT_LDS_STACK>), dim3(blocks), dim3(RT_BLOCK),
                       lds_bytes, stream, sc, cam, pc, reinterpret_cast<float4 *>(d_accum), d_counter, d_queue,
                       regen_idle, sched);
  else
    hipLaunchKernelGGL((k_render_v8<RT_AABB_DIV, RT_V8_MINW, RT_BLOCK, RT_LDS_STACK>), dim3(blocks), dim3(RT_BLOCK),
                       lds_bytes, stream, sc, cam, pc, reinterpret_cast<float4 *>(d_accum), d_counter, d_queue,
                       regen_idle, sched);
#ifdef RT_V2_STATS
  {
    static const char *names[8] = {"shade", "fetch", "newpath", "arm", "pair", "inst", "leaf", "pop"};
    unsigned long long h[32], z[32] = {0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_v2_stats), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v2_stats), z, sizeof z);
    fprintf(stderr, "[v8 stats] waves %llu wave_ticks %llu\n", h[25], h[24]);
    for (int i = 0; i < 8; ++i)
      fprintf(stderr, "[v8 stats] %-8s ticks %14llu (%5.1f%%) exec %12llu lanes %14llu (%.1f/exec)\n", names[i],
              h[i], 100.0 * (double)h[i] / (double)(h[24] ? h[24] : 1), h[8 + i], h[16 + i],
              h[8 + i] ? (double)h[16 + i] / (double)h[8 + i] : 0.0);
  }
#endif
  return hipGetLastError();
}
