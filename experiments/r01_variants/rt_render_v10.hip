// rt_render_v10.hip — k_render_v10: pooled scheduling.  A wave owns POOL = 128 paths instead of 64.
//
// Measured on the default kernel: the VALU pipe is 67 % busy but only 39 % of the lanes of an issued
// instruction do useful work — a lane whose path is at a leaf waits while the wave runs a pair step, and so on.  The
// lever left is lanes per instruction.  Here the paths are decoupled from the lanes: the traversal state of every
// path (reference, stack pointer, object-space ray, closest t, flags, current instance; a 12-entry stack) lives in
// LDS, indexed by path slot ([field][slot], 11 KB per wave), the path state only shading touches in an HBM block as in
// variant 7.  Every iteration the wave classifies its 128 slots, picks ONE kind of step (pair / leaf triangle /
// instance entry or exit / shade-and-regenerate), gathers up to 64 slots that wait for that kind — with two
// candidates per lane a pair step is practically always full — and each lane runs the step for the slot it was
// handed, reading and writing that slot's state in LDS.  One wave per workgroup, fourteen workgroups per compute
// unit (154 of the 160 KiB of LDS; 126 VGPRs).
//
// Per path the sequence of node visits, triangle tests and RNG draws is exactly that of every other variant: the
// image is bit-identical.
//
// RESULT (C3, 64 spp): 2330 Mrays/s against 2950-3000 for the default kernel.  The pooling works as intended — 27 % fewer
// wave iterations (57 M against 78 M per 16-spp launch), pair steps at 45 instead of 35 active lanes, leaf steps at 29
// instead of 13, VALU lane utilisation 53 % instead of 39 % (rocprofv3 PMC) — but the bookkeeping costs what it saves:
// classifying the slots, handing them out and moving the slot state through LDS brings an iteration to 192 VALU + 141
// scalar instructions against 140 + 92 and adds three LDS round trips to its dependency chain, so the wave-level
// instruction count is unchanged (11.0 G) and an iteration takes 2.4 us instead of 1.3 us (-DRT_V10_STATS for the
// counts).  Measured on the way: kind bytes instead of a classification from (state, reference) and bursts of pair
// steps from registers +4 %; LDS stack of 9 / 11 / 12 / 13 / 15 entries (16 / 15 / 14 / 13 / 10 workgroups per compute
// unit; 2.7 % ... 0.001 % of the pushes spill to HBM): 2275 / 2284 / 2325 / 2318 / 2078.  Kept as a measured variant
// (10); what it needs to pay off is a scheduler of a few instructions (per-kind slot queues maintained incrementally).
#include <stdio.h>

#include "rt_device_inl.h"

namespace {

constexpr uint32_t POOL = 128;  // path slots per wave
constexpr uint32_t NSTK = 12;   // stack entries per slot in LDS; deeper ones in the HBM block
constexpr uint32_t SPILL = RT_MAX_STACK - NSTK;

enum : uint32_t { PS_FETCH = 0, PS_TRAVERSE = 2, PS_SHADE = 3, PS_DONE = 4 };

// hot fields, [field][slot] in LDS.  HF_PACK: bits 0-2 state, bit 3 any hit, bit 4 fast slab path allowed,
// bits 8-15 stack pointer, bits 16-31 current instance (the reference's TLAS holds at most 2^15 instances)
enum : uint32_t { HF_REF = 0, HF_PACK, HF_CLOSEST, HF_OX, HF_OY, HF_OZ, HF_DX, HF_DY, HF_DZ, HF_COUNT };
// the kind of step a slot waits for, one byte per slot, written by whoever changes the slot (so that the per-iteration
// classification is one LDS read per slot): 1 pair, 2 instance entry / exit, 3 leaf triangle, 4 shade, 5 needs a
// pixel, 0 done
constexpr uint32_t KIND_DWORDS = POOL / 4;
constexpr uint32_t LDS_DWORDS = (HF_COUNT + NSTK) * POOL + 64 + KIND_DWORDS; // + the compaction list of one step + kinds

// cold fields, [wave][field][slot] in HBM (as in rt_render_v7.hip)
enum : uint32_t {
  CF_OX = 0, CF_OY, CF_OZ, CF_DX, CF_DY, CF_DZ, // path ray (world)
  CF_HU, CF_HV, CF_HTRI, CF_HINST,              // closest hit (t is HF_CLOSEST)
  CF_AX, CF_AY, CF_AZ,                          // attenuation
  CF_SEED, CF_XY, CF_SAMPLE, CF_DEPTH,          // RNG state, pixel, sample and bounce counters
  CF_SX, CF_SY, CF_SZ,                          // sample_radiance
  CF_RX, CF_RY, CF_RZ,                          // radiance (sum over the pixel's samples)
  CF_COUNT
};
constexpr size_t COLD_DWORDS_PER_WAVE = (size_t)(CF_COUNT + SPILL) * POOL;

typedef __attribute__((address_space(3))) uint32_t *lds_ptr;

#ifdef RT_V10_STATS
__device__ unsigned long long g_v10_stats[16]; // [K]: executions, [8 + K]: active lanes; [6]: stack spills; [7]: pushes
#endif

} // namespace

template <int MODE>
__global__ void __launch_bounds__(64, 4)
    k_render_v10(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc, float4 *__restrict__ accum,
                 unsigned long long *ray_counter, uint32_t *__restrict__ queue, uint32_t *__restrict__ cold,
                 uint32_t regen_min, uint32_t sched_arg, uint32_t burst, uint32_t burst_min) {
  extern __shared__ __align__(16) uint32_t lds_raw[];
  const lds_ptr hot = (lds_ptr)lds_raw;                          // [HF_COUNT][POOL]
  const lds_ptr stk = (lds_ptr)lds_raw + HF_COUNT * POOL;        // [NSTK][POOL]
  const lds_ptr list = (lds_ptr)lds_raw + (HF_COUNT + NSTK) * POOL; // [64]
  typedef __attribute__((address_space(3))) uint8_t *lds_u8;
  const lds_u8 kinds = (lds_u8)((lds_ptr)lds_raw + (HF_COUNT + NSTK) * POOL + 64); // [POOL]
  auto kind_of = [](uint32_t state, uint32_t ref) -> uint32_t {
    return state == PS_TRAVERSE ? (!(ref & RT_REF_LEAF) ? 1u : (ref & RT_REF_INST) ? 2u : 3u)
           : state == PS_SHADE ? 4u : state == PS_FETCH ? 5u : 0u;
  };
  const uint32_t lane = threadIdx.x;
  const uint32_t rows = rt_virtual_rows(pc);
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint32_t total_slots = tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
  const bool scene_ok = (sc.flags & RT_SCENE_BOXES_MODERATE) != 0u && !(sched_arg & 0x8000u);
  const uint32_t w_pair = (sched_arg & 0xFu) ? (sched_arg & 0xFu) : 2u;
  const uint32_t w_leaf = ((sched_arg >> 4) & 0xFu) ? ((sched_arg >> 4) & 0xFu) : 3u;
  const uint32_t w_inst = ((sched_arg >> 8) & 0xFu) ? ((sched_arg >> 8) & 0xFu) : 4u;
  uint32_t *const cold_wave = cold + (size_t)blockIdx.x * COLD_DWORDS_PER_WAVE;
  uint32_t *const spill_wave = cold_wave + (size_t)CF_COUNT * POOL; // [slot][SPILL]

#define HLD(f, s) (hot[(f) * POOL + (s)])
#define HLDF(f, s) __uint_as_float(hot[(f) * POOL + (s)])
#define HST(f, s, v) (hot[(f) * POOL + (s)] = (v))
#define HSTF(f, s, v) (hot[(f) * POOL + (s)] = __float_as_uint(v))
#define CLD_U(f, s) (cold_wave[(f) * POOL + (s)])
#define CLD_F(f, s) __uint_as_float(cold_wave[(f) * POOL + (s)])
#define CST_U(f, s, v) (cold_wave[(f) * POOL + (s)] = (v))
#define CST_F(f, s, v) (cold_wave[(f) * POOL + (s)] = __float_as_uint(v))

  // the stack of a slot: push / pop on LDS entries, spill entries in HBM
  auto push = [&](uint32_t slot, uint32_t &sp, uint32_t value) {
#ifdef RT_V10_STATS
    atomicAdd(&g_v10_stats[sp < NSTK ? 7 : 6], 1ull);
#endif
    if (sp < NSTK)
      stk[sp * POOL + slot] = value;
    else
      spill_wave[(size_t)slot * SPILL + (sp - NSTK)] = value;
    ++sp;
  };
  auto pop = [&](uint32_t slot, uint32_t &sp) -> uint32_t { // sp > 0
    --sp;
    uint32_t r = stk[(sp < NSTK ? sp : NSTK - 1u) * POOL + slot];
    if (sp >= NSTK)
      r = spill_wave[(size_t)slot * SPILL + (sp - NSTK)];
    return r;
  };

  for (uint32_t s = lane; s < POOL; s += 64u) { // every slot starts without a pixel
    HST(HF_PACK, s, PS_FETCH);
    HST(HF_REF, s, RT_REF_NONE);
    kinds[s] = 5u;
  }
  asm volatile("" ::: "memory"); // LDS operations of one wave execute in order: only the compiler must not reorder
  __builtin_amdgcn_wave_barrier();

  uint32_t wave_traced = 0;
  bool queue_dry = false; // uniform: the pixel queue returned an index past the end

  while (true) {
    // ---- classify the two slots this lane watches: kinds 1 pair, 2 instance / sentinel, 3 leaf, 4 regenerate ----
    uint32_t kindv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t k = kinds[lane + 64u * h];
      kindv[h] = k == 5u ? (queue_dry ? 0u : 4u) : k;
    }
    uint32_t cnt[5];
#pragma unroll
    for (uint32_t k = 1; k <= 4; ++k)
      cnt[k] = (uint32_t)__popcll(__ballot(kindv[0] == k)) + (uint32_t)__popcll(__ballot(kindv[1] == k));
    const uint32_t n_trav = cnt[1] + cnt[2] + cnt[3];
    if (n_trav + cnt[4] == 0u)
      break; // every slot is done (or waits for a pixel that will not come)
    uint32_t K;
    if (cnt[4] >= regen_min || n_trav == 0u) {
      K = 4u;
    } else {
      const uint32_t c1 = cnt[1] < 64u ? cnt[1] : 64u, c2 = cnt[2] < 64u ? cnt[2] : 64u, c3 = cnt[3] < 64u ? cnt[3] : 64u;
      const uint32_t s1 = c1 * w_pair, s2 = c2 * w_inst, s3 = c3 * w_leaf;
      K = (s1 >= s2 && s1 >= s3) ? 1u : (s3 >= s2 ? 3u : 2u);
    }
    // ---- hand out up to 64 slots of kind K: the lane's own first slot, else one of the second slots ----
    const bool own = kindv[0] == K, second = kindv[1] == K;
    const unsigned long long m_own = __ballot(own), m_second = __ballot(second);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (second)
      list[__popcll(m_second & below)] = lane + 64u;
    asm volatile("" ::: "memory"); // LDS operations of one wave execute in order: only the compiler must not reorder
    __builtin_amdgcn_wave_barrier();
    const uint32_t free_rank = (uint32_t)__popcll(~m_own & below);
    const bool take = !own && free_rank < (uint32_t)__popcll(m_second);
    const uint32_t slot = own ? lane : list[take ? free_rank : 0u];
    const bool active = own || take;
    __builtin_amdgcn_wave_barrier();
#ifdef RT_V10_STATS
    {
      const unsigned long long n_act = (unsigned long long)__popcll(__ballot(active));
      if (lane == 0) {
        atomicAdd(&g_v10_stats[K], 1ull);
        atomicAdd(&g_v10_stats[8 + K], n_act);
      }
    }
#endif

    if (K == 1u) {
      // interior pair: both child boxes, near child next, far child pushed.  A burst: the lanes whose slot is at a pair
      // again afterwards repeat the step from registers (no hand-out, no LDS round trip) while enough of them remain.
      uint32_t ref = 0, fl = 0, sp = 0;
      V3 co = v3(0.f, 0.f, 0.f), cd = co, cinv = co;
      float closest = 0.f;
      if (active) {
        ref = HLD(HF_REF, slot), fl = HLD(HF_PACK, slot);
        sp = (fl >> 8) & 0xFFu;
        co = v3(HLDF(HF_OX, slot), HLDF(HF_OY, slot), HLDF(HF_OZ, slot));
        cd = v3(HLDF(HF_DX, slot), HLDF(HF_DY, slot), HLDF(HF_DZ, slot));
        // the reciprocals of the ray space are recomputed (3 instructions an axis) rather than kept in LDS
        cinv = MODE == RT_AABB_RCP ? inv_direction(cd) : v3(refined_rcp(cd.x), refined_rcp(cd.y), refined_rcp(cd.z));
        closest = HLDF(HF_CLOSEST, slot);
      }
      bool want = active;
      for (uint32_t b = 0;; ++b) {
        if (want) {
          const float4 *q = reinterpret_cast<const float4 *>(sc.pairs + ref);
          const float4 qx = q[0], qy = q[1], qz = q[2];
          const float2 r2 = *reinterpret_cast<const float2 *>(q + 3);
          float dist1, dist2;
          if (fl & 16u) {
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
            if (sp == 0u)
              fl = (fl & ~7u) | PS_SHADE;
            else
              ref = pop(slot, sp);
          } else {
            if (dist2 != kMiss)
              push(slot, sp, c2);
            ref = c1;
          }
          want = (fl & 7u) == PS_TRAVERSE && !(ref & RT_REF_LEAF);
        }
        if (b + 1u >= burst || (uint32_t)__popcll(__ballot(want)) < burst_min)
          break;
      }
      if (active) {
        HST(HF_REF, slot, ref);
        HST(HF_PACK, slot, (fl & 0xFFFF00FFu) | (sp << 8));
        kinds[slot] = (uint8_t)kind_of(fl & 7u, ref);
      }
    } else if (K == 3u) {
      if (active) { // one triangle of the leaf: TriangleGeom::hit, Triangle.slang:6-42
        uint32_t ref = HLD(HF_REF, slot), fl = HLD(HF_PACK, slot);
        uint32_t sp = (fl >> 8) & 0xFFu;
        const V3 co = v3(HLDF(HF_OX, slot), HLDF(HF_OY, slot), HLDF(HF_OZ, slot));
        const V3 cd = v3(HLDF(HF_DX, slot), HLDF(HF_DY, slot), HLDF(HF_DZ, slot));
        const float closest = HLDF(HF_CLOSEST, slot);
        const float4 *q = reinterpret_cast<const float4 *>(sc.tris + (ref & 0x7FFFFFFFu));
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
                HSTF(HF_CLOSEST, slot, t);
                fl |= 8u;
                CST_F(CF_HU, slot, u), CST_F(CF_HV, slot, v);
                CST_U(CF_HTRI, slot, __float_as_uint(t0.w)), CST_U(CF_HINST, slot, fl >> 16);
              }
            }
          }
        }
        if (last) {
          if (sp == 0u)
            fl = (fl & ~7u) | PS_SHADE;
          else
            ref = pop(slot, sp);
        } else {
          ref = ref + 1u;
        }
        HST(HF_REF, slot, ref);
        HST(HF_PACK, slot, (fl & 0xFFFF00FFu) | (sp << 8));
        kinds[slot] = (uint8_t)kind_of(fl & 7u, ref);
      }
    } else if (K == 2u) {
      if (active) { // change of ray space: enter an instance, or (sentinel) return to world space
        uint32_t ref = HLD(HF_REF, slot), fl = HLD(HF_PACK, slot);
        uint32_t sp = (fl >> 8) & 0xFFu;
        const V3 po = v3(CLD_F(CF_OX, slot), CLD_F(CF_OY, slot), CLD_F(CF_OZ, slot));
        const V3 pd = v3(CLD_F(CF_DX, slot), CLD_F(CF_DY, slot), CLD_F(CF_DZ, slot));
        V3 co, cd, cinv;
        bool fast;
        if (ref == RT_REF_SENTINEL) {
          co = po, cd = pd, prep_space<MODE>(co, cd, scene_ok, cinv, fast);
          if (sp == 0u)
            fl = (fl & ~7u) | PS_SHADE;
          else
            ref = pop(slot, sp);
        } else {
          const uint32_t inst = ref & 0x3FFFFFFFu;
          fl = (fl & 0xFFFFu) | (inst << 16);
          const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + inst);
          const float4 a = q[0], b = q[1], c = q[2], e = q[3];
          const V3 c0 = v3(a.x, a.y, a.z), c1 = v3(b.x, b.y, b.z), c2 = v3(c.x, c.y, c.z), c3 = v3(e.x, e.y, e.z);
          co = mul_cols(c0, c1, c2, c3, po, 1.f), cd = mul_cols(c0, c1, c2, c3, pd, 0.f);
          prep_space<MODE>(co, cd, scene_ok, cinv, fast);
          push(slot, sp, RT_REF_SENTINEL);
          ref = __float_as_uint(a.w);
        }
        fl = (fl & ~16u) | (fast ? 16u : 0u);
        HSTF(HF_OX, slot, co.x), HSTF(HF_OY, slot, co.y), HSTF(HF_OZ, slot, co.z);
        HSTF(HF_DX, slot, cd.x), HSTF(HF_DY, slot, cd.y), HSTF(HF_DZ, slot, cd.z);
        HST(HF_REF, slot, ref);
        HST(HF_PACK, slot, (fl & 0xFFFF00FFu) | (sp << 8));
        kinds[slot] = (uint8_t)kind_of(fl & 7u, ref);
      }
    } else {
      // ===== regeneration: slots whose ray is finished, or that have no pixel =====
      uint32_t state = 7u; // lanes without a slot take no part
      uint32_t fl = 0;
      if (active) {
        fl = HLD(HF_PACK, slot);
        state = fl & 7u;
      }
      Path p;
      p.o = p.d = p.attenuation = p.sample_radiance = v3(0.f, 0.f, 0.f);
      V3 radiance = v3(0.f, 0.f, 0.f);
      uint32_t seed = 0, x = 0, y = 0, sample = 0, depth = 0;
      bool arm = false, newpath = false;
      if (state == PS_SHADE) {
        p.o = v3(CLD_F(CF_OX, slot), CLD_F(CF_OY, slot), CLD_F(CF_OZ, slot));
        p.d = v3(CLD_F(CF_DX, slot), CLD_F(CF_DY, slot), CLD_F(CF_DZ, slot));
        p.attenuation = v3(CLD_F(CF_AX, slot), CLD_F(CF_AY, slot), CLD_F(CF_AZ, slot));
        p.sample_radiance = v3(CLD_F(CF_SX, slot), CLD_F(CF_SY, slot), CLD_F(CF_SZ, slot));
        radiance = v3(CLD_F(CF_RX, slot), CLD_F(CF_RY, slot), CLD_F(CF_RZ, slot));
        seed = CLD_U(CF_SEED, slot);
        const uint32_t xy = CLD_U(CF_XY, slot);
        x = xy & 0xFFFFu, y = xy >> 16;
        sample = CLD_U(CF_SAMPLE, slot);
        depth = CLD_U(CF_DEPTH, slot);
        const bool any = (fl & 8u) != 0u;
        Hit hit;
        hit.t = HLDF(HF_CLOSEST, slot);
        hit.u = hit.v = 0.f, hit.tri = hit.inst = 0;
        if (any) {
          hit.u = CLD_F(CF_HU, slot), hit.v = CLD_F(CF_HV, slot);
          hit.tri = CLD_U(CF_HTRI, slot), hit.inst = CLD_U(CF_HINST, slot);
        }
        Rng rng; // as left by the arm step of this bounce: slot depth+1, dim 0
        rng.mode = pc.rng_mode, rng.seed = seed, rng.pixel = y * pc.image_width + x;
        rng.sample = pc.frame_index * spp + sample;
        rng.set_slot(depth + 1);
        const bool scattered = shade(sc, any, hit, rng, p);
        seed = rng.seed;
        ++depth;
        if (scattered && depth < pc.max_depth) {
          arm = true;
        } else {
          radiance = radiance + p.sample_radiance; // RayTracing.slang:170
          if (++sample == spp) {
            accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
            state = PS_FETCH;
          } else {
            newpath = true;
          }
        }
      }
      // slots without a pixel pull from the queue; the loop is uniform over the wave and ends when no lane needs a
      // pixel any more or the queue is exhausted
      while (true) {
        const bool need = state == PS_FETCH && !newpath && !arm;
        const unsigned long long mask = __ballot(need);
        if (mask == 0ull)
          break;
        const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
        uint32_t base = 0;
        if (lane == leader)
          base = atomicAdd(queue, (uint32_t)__popcll(mask));
        base = __shfl(base, (int)leader, 64);
        if (base >= total_slots) {
          queue_dry = true;
          if (need)
            state = PS_DONE;
          break;
        }
        if (need) {
          const uint32_t qs = base + (uint32_t)__popcll(mask & below);
          if (qs < total_slots) {
            const uint32_t tile = qs >> 6, in = qs & 63u;
            x = (tile % tiles_x) * 8u + (in & 7u);
            y = rt_row(pc, (tile / tiles_x) * 8u + (in >> 3));
            if (x < pc.image_width && y < pc.row_end) { // else edge padding: fetch again
              sample = 0;
              seed = make_rng(pc, x, y).seed;
              radiance = v3(0.f, 0.f, 0.f);
              if (pc.max_depth == 0) { // degenerate: no sample contributes (RayTracing.slang:92)
                accumulate(accum, pc.image_width, x, y, radiance, pc.sqrt_spp, pc.accum_count);
              } else {
                newpath = true;
              }
            }
          } else {
            state = PS_DONE;
          }
        }
      }
      if (newpath) {
        Rng rng;
        rng.mode = pc.rng_mode, rng.seed = seed, rng.pixel = y * pc.image_width + x;
        rng.sample = pc.frame_index * spp + sample;
        rng.set_slot(0);
        camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
        seed = rng.seed;
        depth = 0;
        arm = true;
      }
      if (arm) {
        ++wave_traced;
        V3 co = p.o, cd = p.d, cinv;
        bool fast;
        prep_space<MODE>(co, cd, scene_ok, cinv, fast);
        const uint32_t ref = sc.root_ref;
        HST(HF_REF, slot, ref);
        HSTF(HF_CLOSEST, slot, 1000.f); // RayTracing.slang:93-95
        HSTF(HF_OX, slot, co.x), HSTF(HF_OY, slot, co.y), HSTF(HF_OZ, slot, co.z);
        HSTF(HF_DX, slot, cd.x), HSTF(HF_DY, slot, cd.y), HSTF(HF_DZ, slot, cd.z);
        HST(HF_PACK, slot, (ref == RT_REF_NONE ? PS_SHADE : PS_TRAVERSE) | (fast ? 16u : 0u)); // sp 0, instance 0
        kinds[slot] = (uint8_t)kind_of(ref == RT_REF_NONE ? PS_SHADE : PS_TRAVERSE, ref);
        CST_F(CF_OX, slot, p.o.x), CST_F(CF_OY, slot, p.o.y), CST_F(CF_OZ, slot, p.o.z);
        CST_F(CF_DX, slot, p.d.x), CST_F(CF_DY, slot, p.d.y), CST_F(CF_DZ, slot, p.d.z);
        CST_F(CF_AX, slot, p.attenuation.x), CST_F(CF_AY, slot, p.attenuation.y), CST_F(CF_AZ, slot, p.attenuation.z);
        CST_F(CF_SX, slot, p.sample_radiance.x), CST_F(CF_SY, slot, p.sample_radiance.y), CST_F(CF_SZ, slot, p.sample_radiance.z);
        CST_F(CF_RX, slot, radiance.x), CST_F(CF_RY, slot, radiance.y), CST_F(CF_RZ, slot, radiance.z);
        CST_U(CF_SEED, slot, seed);
        CST_U(CF_XY, slot, x | (y << 16));
        CST_U(CF_SAMPLE, slot, sample);
        CST_U(CF_DEPTH, slot, depth);
      } else if (active) {
        HST(HF_PACK, slot, state == PS_DONE ? PS_DONE : PS_FETCH); // no pixel (yet), or none left
        kinds[slot] = state == PS_DONE ? 0u : 5u;
      }
    }
    // the next classification reads what this step wrote (other lanes' slots): LDS operations of one wave execute in
    // order, the fence keeps the compiler from moving them
    asm volatile("" ::: "memory"); // LDS operations of one wave execute in order: only the compiler must not reorder
    __builtin_amdgcn_wave_barrier();
  }
  count_rays(ray_counter, wave_traced);
#undef HLD
#undef HLDF
#undef HST
#undef HSTF
#undef CLD_U
#undef CLD_F
#undef CST_U
#undef CST_F
}

constexpr uint32_t WAVES_PER_CU = 14; // LDS bound: 14 x 11,264 B
size_t rt_v10_cold_bytes(uint32_t cus) { return (size_t)cus * WAVES_PER_CU * COLD_DWORDS_PER_WAVE * sizeof(uint32_t); }

hipError_t rt_launch_render_v10(uint32_t variant, const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                                const rt_render_params &pc, float *d_accum, unsigned long long *d_counter,
                                uint32_t *d_queue, uint32_t *d_cold, uint32_t cus, hipStream_t stream) {
  const uint32_t rows = rt_virtual_rows(pc);
  const uint64_t slots = (uint64_t)((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  if (pc.image_width > 0xFFFFu || pc.image_height > 0xFFFFu)
    return hipErrorInvalidValue; // pixel coordinates are packed 16:16 in the cold block
  const uint32_t sched = ((variant >> 16) & 0xFFFu) | ((variant >> 31) ? 0x8000u : 0u);
  const uint32_t burst = ((variant >> 28) & 7u) ? ((variant >> 28) & 7u) : 2u; // pair steps per hand-out (1 / 2 / 3 / 6: 2199 / 2275 / 2259 / 2273)
  const uint32_t burst_min = 24u;                                               // ... while this many lanes go on
  uint32_t regen_min = (variant >> 8) & 0xFFu;
  if (regen_min == 0 || regen_min > POOL)
    regen_min = 48u; // measured: 16 / 32 / 48 / 64 -> 1984 / 2132 / 2212 / 2180 Mrays/s
  const uint64_t resident = (uint64_t)cus * WAVES_PER_CU, need = (slots + POOL - 1) / POOL;
  const uint32_t blocks = (uint32_t)(need < resident ? need : resident);
  if (blocks == 0)
    return hipSuccess;
  hipError_t e = hipMemsetAsync(d_queue, 0, 64 * sizeof(uint32_t), stream);
  if (e != hipSuccess)
    return e;
  float4 *acc = reinterpret_cast<float4 *>(d_accum);
  const size_t lds = LDS_DWORDS * sizeof(uint32_t);
  if (pc.aabb_mode == RT_AABB_RCP)
    hipLaunchKernelGGL((k_render_v10<RT_AABB_RCP>), dim3(blocks), dim3(64), lds, stream, sc, cam, pc, acc, d_counter,
                       d_queue, d_cold, regen_min, sched, burst, burst_min);
  else
    hipLaunchKernelGGL((k_render_v10<RT_AABB_DIV>), dim3(blocks), dim3(64), lds, stream, sc, cam, pc, acc, d_counter,
                       d_queue, d_cold, regen_min, sched, burst, burst_min);
#ifdef RT_V10_STATS
  {
    unsigned long long h[16], z[16] = {0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_v10_stats), sizeof h);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_v10_stats), z, sizeof z);
    static const char *names[5] = {"", "pair", "inst", "leaf", "regen"};
    for (int k = 1; k <= 4; ++k)
      fprintf(stderr, "[v10 stats] %-6s exec %12llu lanes %.1f/exec\n", names[k], h[k], h[k] ? (double)h[8 + k] / (double)h[k] : 0.0);
    fprintf(stderr, "[v10 stats] pushes in LDS %llu, spilled %llu\n", h[7], h[6]);
  }
#endif
  return hipGetLastError();
}
