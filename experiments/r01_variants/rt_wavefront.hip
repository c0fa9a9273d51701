// rt_wavefront.hip — wavefront formulation of the render path (kernel variant 4).
//
// Why: PMC profiles of the megakernels (profiles/r01_*) show that the slab
// tests are NOT what the VALU spends its time on: in k_render_v2 more wave
// instructions go to shading / ray generation / leaf handling executed with
// 10-25 % of the lanes than to the pair steps.  Splitting the path tracer at
// the ray boundary removes that code from the divergent loop:
//
//   k_wf_init    one lane per pixel: RNG seed, first camera ray            (RayTracing.slang:66-90)
//   k_wf_extend  persistent wave64s: every lane pulls rays from the round's
//                queue and traverses them; a finished lane only stores its
//                hit record and pulls the next ray                         (TLAS.slang:14-67)
//   k_wf_shade   one lane per live path, all 64 lanes busy: material
//                switch, sky, path termination, next sample's camera ray,
//                running-mean write-out, compaction of the live list       (RayTracing.slang:103-179)
//
// One pixel has at most one ray in flight (the reference's RNG state is
// sequential per pixel), so a frame takes at most spp * max_depth rounds of
// (extend, shade); every pixel performs exactly the arithmetic of compute_main
// in the same order, so the image is bit-identical to the megakernels'.
// Path state lives in HBM between rounds (116 B per pixel, SoA, 16-byte lanes).
#include "rt_device_inl.h"
#include "rt_wavefront.h"
#include <stdio.h>

#pragma clang fp contract(off)

#define RT_WF_CHUNK 256u // rays a wave takes from the round's queue per atomic

namespace {

__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, bool want, uint32_t lane) {
  // wave-aggregated atomic: returns this lane's slot (valid when want)
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull)
    return 0;
  const uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
  uint32_t base = 0;
  if (lane == leader)
    base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, (int)leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

struct PathState {
  V3 attenuation, sample_radiance, radiance;
  uint32_t seed, sample, depth;
};

__device__ __forceinline__ void store_state(const rt_wf_buffers &b, uint32_t p, const PathState &s) {
  b.st_a[p] = make_float4(s.attenuation.x, s.attenuation.y, s.attenuation.z, s.sample_radiance.x);
  b.st_b[p] = make_float4(s.sample_radiance.y, s.sample_radiance.z, s.radiance.x, s.radiance.y);
  b.st_c[p] = make_float4(s.radiance.z, __uint_as_float(s.seed),
                          __uint_as_float((s.sample << 8) | (s.depth & 255u)), 0.f);
}

__device__ __forceinline__ PathState load_state(const rt_wf_buffers &b, uint32_t p) {
  float4 a = b.st_a[p], bb = b.st_b[p], c = b.st_c[p];
  PathState s;
  s.attenuation = v3(a.x, a.y, a.z);
  s.sample_radiance = v3(a.w, bb.x, bb.y);
  s.radiance = v3(bb.z, bb.w, c.x);
  s.seed = __float_as_uint(c.y);
  uint32_t sd = __float_as_uint(c.z);
  s.sample = sd >> 8;
  s.depth = sd & 255u;
  return s;
}

__device__ __forceinline__ Rng rng_of(const rt_render_params &pc, uint32_t p, const PathState &s,
                                      uint32_t slot, uint32_t spp) {
  Rng rng;
  rng.mode = pc.rng_mode;
  rng.seed = s.seed;
  rng.pixel = p; // y * image_width + x, as make_rng
  rng.sample = pc.frame_index * spp + s.sample;
  rng.slot = slot;
  rng.dim = 0;
  return rng;
}

} // namespace

// ---------------------------------------------------------------------------
// k_wf_init: first camera ray of every pixel of rows [row_begin, row_end), in
// 8x8-tile order so that neighbouring queue entries are neighbouring pixels.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256)
    k_wf_init(rt_wf_buffers b, rt_camera_uniforms cam, rt_render_params pc) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t rows = rt_virtual_rows(pc);
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint32_t total_slots = tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  const uint32_t tile = slot >> 6, in = slot & 63u;
  const uint32_t x = (tile % tiles_x) * 8u + (in & 7u);
  const uint32_t y = rt_row(pc, (tile / tiles_x) * 8u + (in >> 3));
  const bool valid = slot < total_slots && x < pc.image_width && y < pc.row_end;
  const uint32_t p = y * pc.image_width + x;
  if (valid) {
    Rng rng = make_rng(pc, x, y);
    rng.sample = pc.frame_index * (pc.sqrt_spp * pc.sqrt_spp);
    rng.set_slot(0);
    Path path;
    camera_ray(cam, 1.f / (float)pc.sqrt_spp, x, y, 0, 0, rng, path);
    PathState s;
    s.attenuation = path.attenuation;
    s.sample_radiance = path.sample_radiance;
    s.radiance = v3(0.f, 0.f, 0.f);
    s.seed = rng.seed;
    s.sample = 0;
    s.depth = 0;
    store_state(b, p, s);
    b.ray_o[p] = make_float4(path.o.x, path.o.y, path.o.z, 0.f);
    b.ray_d[p] = make_float4(path.d.x, path.d.y, path.d.z, 0.f);
  }
  const uint32_t at = wave_append(b.count + 0, valid, lane);
  if (valid)
    b.list[0][at] = p;
}

// ---------------------------------------------------------------------------
// k_wf_extend: closest hit for every ray of the round.  Persistent: the grid is
// sized to the chip; each lane pulls ray indices from the round's queue until
// it is empty.  The traversal is the megakernels' (same visit order, same
// arithmetic), scheduled "while-while": the wave stays in the pair loop while
// at least `pair_min` lanes have an interior node to test, then lets the lanes
// waiting at leaves (triangles / instances) run once; lanes whose ray is done
// store the hit and are refilled once `refill_min` of them are idle.
// ---------------------------------------------------------------------------
template <int MODE, int MINW, int NLDS>
__global__ void __launch_bounds__(256, MINW)
    k_wf_extend(rt_dev_scene sc, rt_wf_buffers b, uint32_t round, uint32_t pair_min,
                uint32_t refill_min, unsigned long long *ray_counter) {
  __shared__ uint32_t lds_stack[NLDS * 256];
  const uint32_t nrays = b.count[round % 3u];
  if (nrays == 0)
    return;
  uint32_t *head = b.head + (round & 1u);
  const uint32_t *list = b.list[round & 1u];
  RT_STACK_DECL(256, NLDS, st, lds_stack);
  const uint32_t lane = threadIdx.x & 63u;

#ifdef RT_WF_STATS
  uint32_t dbg_pair_iters = 0, dbg_pair_lanes = 0, dbg_leaf_iters = 0, dbg_leaf_lanes = 0;
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
  bool active = false, exhausted = false;
  bool queue_empty = false;               // wave-uniform
  uint32_t chunk_next = 0, chunk_end = 0; // wave-uniform: this wave's slice of the queue
  uint32_t p = 0, traced = 0;
  V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 0.f);
  uint32_t ref = RT_REF_NONE, cur_inst = 0;
  V3 co = o, cd = d, cinv = o;
  float closest = 0.f;
  Hit hit;
  hit.t = 0.f, hit.u = hit.v = 0.f, hit.tri = RT_INVALID_ID, hit.inst = RT_INVALID_ID;

  while (true) {
    // ---- refill idle lanes from the wave's chunk of the queue ----
    // One returning atomic per RT_WF_CHUNK rays, not per refill: a single
    // device-scope head word sustains only ~90 dequeues/us (MI355X_MICROARCH.md,
    // row 'dequeue'), which at one atomic per 16 rays was the whole round time.
    {
      bool want = !active && !exhausted;
      while (true) {
        const unsigned long long mask = __ballot(want);
        if (mask == 0ull)
          break;
        if (chunk_next == chunk_end) {
          if (queue_empty) {
            if (want)
              exhausted = true;
            break;
          }
          uint32_t base = 0;
          if (lane == 0)
            base = atomicAdd(head, RT_WF_CHUNK);
          base = __builtin_amdgcn_readfirstlane(base);
          chunk_next = base < nrays ? base : nrays;
          chunk_end = base + RT_WF_CHUNK < nrays ? base + RT_WF_CHUNK : nrays;
          if (chunk_next == chunk_end)
            queue_empty = true;
          continue;
        }
        const uint32_t avail = chunk_end - chunk_next;
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t n = (uint32_t)__popcll(mask);
        const bool got = want && rank < avail;
        const uint32_t idx = chunk_next + rank;
        chunk_next += n < avail ? n : avail;
        if (got) {
          want = false;
          p = list[idx];
          float4 ro = b.ray_o[p], rd = b.ray_d[p];
          o = v3(ro.x, ro.y, ro.z);
          d = v3(rd.x, rd.y, rd.z);
          co = o, cd = d;
          if (MODE == RT_AABB_RCP)
            cinv = inv_direction(d);
          closest = 1000.f; // RayTracing.slang:93-95
          hit.tri = RT_INVALID_ID;
          hit.inst = RT_INVALID_ID;
          hit.t = 1000.f, hit.u = hit.v = 0.f;
          cur_inst = 0;
          st.sp = 0;
          ref = sc.root_ref;
          ++traced;
          active = ref != RT_REF_NONE;
          if (!active) { // empty scene: miss
            b.hit_a[p] = make_float4(hit.t, hit.u, hit.v, __uint_as_float(RT_INVALID_ID));
            b.hit_inst[p] = RT_INVALID_ID;
          }
        }
      }
    }
    if (__ballot(active) == 0ull) {
      if (__ballot(!exhausted) == 0ull)
        break;
      continue;
    }
    // ---- traverse ----
    while (true) {
      // pair loop
      while (true) {
        const bool at_pair = active && !(ref & RT_REF_LEAF);
        const unsigned long long m = __ballot(at_pair);
        if (m == 0ull)
          break;
        if ((uint32_t)__popcll(m) < pair_min && __ballot(active && (ref & RT_REF_LEAF)) != 0ull)
          break;
#ifdef RT_WF_STATS
        ++dbg_pair_iters;
        dbg_pair_lanes += (uint32_t)__popcll(m);
#endif
        if (at_pair) {
          const float4 *q = reinterpret_cast<const float4 *>(sc.pairs + ref);
          float4 lo0, hi0, lo1, hi1;
          unpack_pair(q[0], q[1], q[2], q[3], lo0, hi0, lo1, hi1);
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
          if (dist1 != kMiss) {
            if (dist2 != kMiss)
              st.push(c2);
            ref = c1;
          } else {
            while (true) { // pop
              if (st.sp == 0) {
                active = false;
                break;
              }
              ref = st.pop();
              if (ref != RT_REF_SENTINEL)
                break;
              co = o, cd = d;
              if (MODE == RT_AABB_RCP)
                cinv = inv_direction(d);
            }
          }
        }
      }
#ifdef RT_WF_STATS
      ++dbg_leaf_iters;
      dbg_leaf_lanes += (uint32_t)__popcll(__ballot(active && (ref & RT_REF_LEAF)));
#endif
      // leaf step for the lanes waiting at a leaf
      if (active && (ref & RT_REF_LEAF)) {
        if (ref & RT_REF_INST) {
          cur_inst = ref & 0x3FFFFFFFu;
          const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
          float4 a = q[0], bq = q[1], c = q[2], e = q[3];
          V3 c0 = v3(a.x, a.y, a.z), c1 = v3(bq.x, bq.y, bq.z), c2 = v3(c.x, c.y, c.z),
             c3 = v3(e.x, e.y, e.z);
          co = mul_cols(c0, c1, c2, c3, o, 1.f);
          cd = mul_cols(c0, c1, c2, c3, d, 0.f);
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
            hit.t = t;
            hit.u = u;
            hit.v = v;
            hit.tri = __float_as_uint(t0.w);
            hit.inst = cur_inst;
          } while (!last);
          while (true) { // pop
            if (st.sp == 0) {
              active = false;
              break;
            }
            ref = st.pop();
            if (ref != RT_REF_SENTINEL)
              break;
            co = o, cd = d;
            if (MODE == RT_AABB_RCP)
              cinv = inv_direction(d);
          }
        }
      }
      const unsigned long long act = __ballot(active);
      if (act == 0ull)
        break;
      if ((uint32_t)__popcll(__ballot(!active && !exhausted)) >= refill_min)
        break;
    }
    // ---- finished lanes: store the hit record ----
    if (!active && !exhausted && ref != RT_REF_NONE) {
      b.hit_a[p] = make_float4(hit.t, hit.u, hit.v, __uint_as_float(hit.tri));
      b.hit_inst[p] = hit.inst;
      ref = RT_REF_NONE; // stored
    }
  }
#ifdef RT_WF_STATS
  if (lane == 0) {
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(b.count + 16);
    atomicAdd(dbg + 0, (unsigned long long)dbg_pair_iters);
    atomicAdd(dbg + 1, (unsigned long long)dbg_pair_lanes);
    atomicAdd(dbg + 2, (unsigned long long)dbg_leaf_iters);
    atomicAdd(dbg + 3, (unsigned long long)dbg_leaf_lanes);
    atomicAdd(dbg + 4, (unsigned long long)(__builtin_amdgcn_s_memtime() - dbg_t0));
    atomicAdd(dbg + 5, 1ull);
  }
#endif
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_wf_shade: everything of one bounce after intersect_tlas, for every live
// path of the round; generates the next ray (scatter or next sample's camera
// ray), finishes pixels (running mean, RayTracing.slang:174-179) and builds the
// next round's compacted list.  Also resets the counters the next round needs.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256)
    k_wf_shade(rt_dev_scene sc, rt_wf_buffers b, rt_camera_uniforms cam, rt_render_params pc,
               float4 *__restrict__ accum, uint32_t round) {
  const uint32_t n = b.count[round % 3u];
  uint32_t *next_count = b.count + (round + 1u) % 3u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    b.count[(round + 2u) % 3u] = 0; // target of the NEXT round's shade
    b.head[(round + 1u) & 1u] = 0;  // queue head of the NEXT round's extend
  }
  if (n == 0)
    return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t *list = b.list[round & 1u];
  uint32_t *next_list = b.list[(round + 1u) & 1u];
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp;
  const uint32_t stride = gridDim.x * 256u;
  __shared__ uint32_t wave_total[4], wave_base[4];
  // every thread of the block iterates the same number of times (barriers inside)
  for (uint32_t base = blockIdx.x * 256u; base < n; base += stride) {
    const uint32_t i0 = base + (threadIdx.x & ~63u);
    const uint32_t i = i0 + lane;
    const bool live = i < n;
    bool again = false;
    uint32_t p = 0;
    if (live) {
      p = list[i];
      PathState s = load_state(b, p);
      float4 ro = b.ray_o[p], rd = b.ray_d[p], ha = b.hit_a[p];
      Path path;
      path.o = v3(ro.x, ro.y, ro.z);
      path.d = v3(rd.x, rd.y, rd.z);
      path.attenuation = s.attenuation;
      path.sample_radiance = s.sample_radiance;
      Hit hit;
      hit.t = ha.x, hit.u = ha.y, hit.v = ha.z;
      hit.tri = __float_as_uint(ha.w);
      hit.inst = b.hit_inst[p];
      const bool any = hit.tri != RT_INVALID_ID;
      Rng rng = rng_of(pc, p, s, s.depth + 1u, spp);
      const bool scattered = shade(sc, any, hit, rng, path);
      s.seed = rng.seed;
      ++s.depth;
      again = true;
      if (!(scattered && s.depth < pc.max_depth)) {
        s.radiance = s.radiance + path.sample_radiance; // RayTracing.slang:170
        ++s.sample;
        if (s.sample == spp) {
          const uint32_t x = p % pc.image_width, y = p / pc.image_width;
          accumulate(accum, pc.image_width, x, y, s.radiance, pc.sqrt_spp, pc.accum_count);
          again = false;
        } else {
          const uint32_t x = p % pc.image_width, y = p / pc.image_width;
          rng = rng_of(pc, p, s, 0, spp);
          camera_ray(cam, recip_sqrt_spp, x, y, s.sample % pc.sqrt_spp, s.sample / pc.sqrt_spp, rng,
                     path);
          s.seed = rng.seed;
          s.depth = 0;
        }
      }
      if (again) {
        s.attenuation = path.attenuation;
        s.sample_radiance = path.sample_radiance;
        store_state(b, p, s);
        b.ray_o[p] = make_float4(path.o.x, path.o.y, path.o.z, 0.f);
        b.ray_d[p] = make_float4(path.d.x, path.d.y, path.d.z, 0.f);
      }
    }
    // block-aggregated compaction: one atomic per 256 paths
    const unsigned long long mask = __ballot(again);
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0)
      wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t t0 = wave_total[0], t1 = wave_total[1], t2 = wave_total[2], t3 = wave_total[3];
      const uint32_t total = t0 + t1 + t2 + t3;
      const uint32_t base_out = total ? atomicAdd(next_count, total) : 0u;
      wave_base[0] = base_out, wave_base[1] = base_out + t0, wave_base[2] = base_out + t0 + t1;
      wave_base[3] = base_out + t0 + t1 + t2;
    }
    __syncthreads();
    if (again)
      next_list[wave_base[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = p;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Host launcher: init, then spp * max_depth rounds of (extend, shade).  Rounds
// past the last live path find an empty list and return at once.
// ---------------------------------------------------------------------------
size_t rt_wf_bytes(uint32_t width, uint32_t height) {
  const size_t P = (size_t)width * height;
  return P * (6 * 16 + 4 + 2 * 4) + 256; // 6 float4 arrays, hit_inst, two lists, counters
}

rt_wf_buffers rt_wf_carve(void *mem, uint32_t width, uint32_t height) {
  const size_t P = (size_t)width * height;
  rt_wf_buffers b;
  uint8_t *at = static_cast<uint8_t *>(mem);
  b.count = reinterpret_cast<uint32_t *>(at);
  b.head = b.count + 4;
  at += 256;
  float4 **f4[6] = {&b.ray_o, &b.ray_d, &b.hit_a, &b.st_a, &b.st_b, &b.st_c};
  for (auto ptr : f4) {
    *ptr = reinterpret_cast<float4 *>(at);
    at += P * 16;
  }
  b.hit_inst = reinterpret_cast<uint32_t *>(at);
  at += P * 4;
  b.list[0] = reinterpret_cast<uint32_t *>(at);
  at += P * 4;
  b.list[1] = reinterpret_cast<uint32_t *>(at);
  return b;
}

hipError_t rt_launch_render_wavefront(const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                                      const rt_render_params &pc, float *d_accum,
                                      unsigned long long *d_counter, const rt_wf_buffers &b,
                                      uint32_t pair_min, uint32_t refill_min, uint32_t occupancy_knob,
                                      uint32_t cus, hipStream_t stream) {
  if (cus == 0)
    return hipErrorInvalidValue;
  const uint32_t rows = rt_virtual_rows(pc);
  const uint32_t slots = ((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  hipError_t e = hipMemsetAsync(b.count, 0, 256, stream);
  if (e != hipSuccess)
    return e;
  hipLaunchKernelGGL(k_wf_init, dim3((slots + 255u) / 256u), dim3(256), 0, stream, b, cam, pc);
  const uint32_t waves = occupancy_knob == 6 || occupancy_knob == 8 ? occupancy_knob : 4u;
  const uint32_t persistent = (uint32_t)cus * waves; // blocks of 256 threads per CU = waves per SIMD
  uint32_t grid = (slots + 255u) / 256u;
  if (grid > persistent)
    grid = persistent;
  const uint32_t rounds = pc.sqrt_spp * pc.sqrt_spp * pc.max_depth;
#define RT_EXT(MODE, W, NL)                                                                         \
  hipLaunchKernelGGL((k_wf_extend<MODE, W, NL>), dim3(grid), dim3(256), 0, stream, sc, b, r, pair_min, \
                     refill_min, d_counter)
  for (uint32_t r = 0; r < rounds; ++r) {
    if (pc.aabb_mode == RT_AABB_RCP) {
      if (waves == 8) RT_EXT(RT_AABB_RCP, 8, 16);
      else if (waves == 6) RT_EXT(RT_AABB_RCP, 6, 24);
      else RT_EXT(RT_AABB_RCP, 4, 24);
    } else {
      if (waves == 8) RT_EXT(RT_AABB_DIV, 8, 16);
      else if (waves == 6) RT_EXT(RT_AABB_DIV, 6, 24);
      else RT_EXT(RT_AABB_DIV, 4, 24);
    }
    hipLaunchKernelGGL(k_wf_shade, dim3(grid), dim3(256), 0, stream, sc, b, cam, pc,
                       reinterpret_cast<float4 *>(d_accum), r);
  }
#undef RT_EXT
#ifdef RT_WF_STATS
  {
    unsigned long long h[6];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h, b.count + 16, sizeof h, hipMemcpyDeviceToHost);
    fprintf(stderr, "[wf stats] pair_iters %llu pair_lanes %llu leaf_iters %llu leaf_lanes %llu wave_cycles %llu waves %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  }
#endif
  return hipGetLastError();
}
