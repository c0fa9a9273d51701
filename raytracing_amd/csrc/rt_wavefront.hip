// rt_wavefront.hip — kernel 9: the sample-parallel wavefront pipeline for the keyed PCG RNG.
//
// The reference's RNG (a Wang hash whose state runs through all samples and bounces of a pixel,
// Random.slang:7-19, RayTracing.slang:71-72) allows one path per pixel in flight, which is why the default kernel is
// a megakernel: ray generation, traversal, shading and accumulation in one set of registers (109 VGPRs, 4 waves per
// SIMD).  With the keyed PCG RNG of north_star — every draw a hash of (pixel, sample, bounce, dimension) — the
// samples of a pixel are independent, so the path can be split where its register needs differ:
//
//   k_wf_shade (init)  fills the path slots with camera rays                                 (RayTracing.slang:78-90)
//   k_wf_trace         closest hit of every live path's current ray: traversal state only    (TLAS.slang, BVHNode.slang)
//   k_wf_shade         shades the hits at full lane occupancy, one lane per path, writes the scattered ray back into
//                      the slot or ends the path and starts the slot's next one              (RayTracing.slang:103-170)
//   k_wf_resolve       sums every pixel's sample radiances IN SAMPLE ORDER and folds them into the running mean
//                                                                                            (RayTracing.slang:170-179)
//
// Path state lives in HBM between the kernels (84 B per slot, millions of slots: the design leans on MI355X's HBM
// capacity and bandwidth instead of registers).  A path is the pair (pixel, sample); path ids run tile-major —
// id = (tile * chunk_samples + sample) * 64 + pixel_in_tile — so consecutive ids are the 64 pixels of one 8x8 tile:
// the camera rays a wave starts together are coherent, and the resolve kernel reads its samples coalesced.  A window
// of 256 slots belongs to one workgroup of the shade kernel for the whole chunk: when a path ends the workgroup
// starts the next one of its own share of the ids, so there are no global lists and no global compaction, the state
// traffic of the shade kernel is coalesced, and every round traces about `slots` rays until the chunk runs out.
// Inside its window the shade pass keeps the paths ordered (those that go on, then the new camera rays: k_wf_shade).
// Per pixel the arithmetic, the RNG keys and the order of the radiance sum are those of k_render_v0 in PCG mode, so
// the image is the same bit for bit.
//
// End of round 3: NO ROUNDS.  The kernels above run as rounds over all slots with the host between them (kept:
// rt_debug_wavefront_persistent(ctx, 0)).  By default a launch is now ONE dispatch of k_wf_trace in its PERSIST mode —
// every resident wave owns a window of 512 slots and fills it with new paths, traces it and shades it in a loop
// (wp_fill / the step loop / wp_shade) — followed by k_wf_resolve; the arithmetic per path is the same, so the image is.
// The first 320 pair records (the breadth-first top of the scene) are read from an LDS copy (NCACHE).
//
// Round 3: FRAME BATCHES and the reference's own RNG.  The reference reaches high sample counts by progressive frames
// of sqrt_spp^2 (= 1) samples each, re-seeding the Wang state per (x, y, frame_index) (RayTracing.slang:71-72,
// Renderer.cpp:21, 625, 651): the samples of ONE frame of a pixel are a sequence, but (pixel, frame) pairs are
// independent.  rt_render_frames hands a launch `frame_count` consecutive frames, and the unit of work in a slot — an
// ENTRY — is
//   keyed PCG:  (pixel, frame, sample): every sample independent, as before; entry j = frame * spp + sample
//   Wang:       (pixel, frame): the frame's samples run one after the other IN the slot, the Wang state (4 B per slot)
//               carried from bounce to bounce and from sample to sample, the frame's radiance summed in sample order
//               in the slot (16 B per slot, touched when a sample ends, only if spp > 1); entry j = frame
// k_wf_resolve then walks a pixel's entries in order, closes a frame every `per_frame` entries and applies the running
// mean (RayTracing.slang:174-179) frame by frame — the same operations in the same order as frame_count launches.
#include <vector>

#include "rt_device_inl.h"
#include "rt_wavefront.h"

#pragma clang fp contract(off)

namespace {

// ctrl words.  Same-address atomics cost about 20 ns each device-wide (measured: a first version with one id counter,
// one list counter and one queue head took 7 ms to hand 8 M paths to 131 K waves), so every hot counter is striped
// over WF_STRIPES addresses and touched once per wave or per 512-slot chunk.
// Atomics execute at the L2 per cache line, so the stripes of one counter sit WF_PAD words (256 B) apart: counters
// packed into one line serialise like a single address.
#define WF_STRIPES 64u
#ifndef WF_SHADE_BLOCK
#define WF_SHADE_BLOCK 256 // lanes of a k_wf_shade workgroup = slots of its window
#endif
#define WF_PAD 64u
enum : uint32_t {
  // live paths after a shade pass, per stripe (summed by the host; the lengths of the live-slot list's segments).
  // Two sets: pass p counts into set p & 1 while the trace pass before it still reads set (p - 1) & 1.  The heads sit
  // between them so that the heads and either set are cleared with one memset.
  WF_ALIVE0 = 0,
  WF_HEAD = WF_ALIVE0 + WF_STRIPES * WF_PAD, // [WF_STRIPES * WF_PAD] next slot of each band of the slot space (k_wf_trace)
  WF_USED = WF_HEAD + 32,                   // 64 bits, one per band whose slots are all handed out (a cache line of its own
                                            // inside band 0's padding; cleared with the heads)
  WF_ALIVE1 = WF_HEAD + WF_STRIPES * WF_PAD,
  WF_CTRL_WORDS = WF_ALIVE1 + WF_STRIPES * WF_PAD
};
__host__ __device__ inline uint32_t wf_alive_set(uint32_t pass) { return (pass & 1u) ? WF_ALIVE1 : WF_ALIVE0; }
#define WF_DEAD 0xFFFFFFFFu // depth word of a slot without a path
// Depth word bit 30: the slot's `rad` entry holds the path's sample radiance.  Clear = the radiance is +0, +0, +0 and
// `rad` is neither written nor read — the state of practically every path until its last bounce, since only an emitter
// or the sky adds to it (RayTracing.slang:151,165) and both end the path; a path whose radiance is anything else (a
// NaN from an infinite attenuation times a zero emission) sets the bit and goes through `rad`.  Saves 32 of the
// ~230 B of HBM traffic per ray in the shade kernel.  (max_depth >= 2^30 is refused by the launcher.)
#define WF_HAS_RAD 0x40000000u
// Depth word bit 29: the slot's `att` entry holds the path's attenuation.  Clear = a fresh path, attenuation 1, 1, 1
// (RayTracing.slang:88): not written by the regeneration, not read by the path's first shade.
#define WF_HAS_ATT 0x20000000u
// Depth word bit 28: the slot's ray is SUSPENDED — a trace wave that ran out of fresh slots with only a few rays left
// wrote their traversal state (node cursor, closest hit, stack) to the pool record whose index sits in hit_v[slot]
// and ended; the shade pass leaves such a slot alone and the next round's trace pass resumes the ray where it stopped,
// in a wave with full lanes.  (max_depth >= 2^28 is refused by the launcher.)
#define WF_SUSPENDED 0x10000000u
#define WF_SUSP_STEPS 64u
#define WF_REC_WORDS 64u                    // pool record: 8 words of state + the stack entries, 256 B
#define WF_REC_STACK (WF_REC_WORDS - 8u)    // deepest stack a record holds; a wave with a deeper one does not suspend

struct WfChunk {
  uint32_t s0;       // first entry of the chunk (entries count through the launch's frames, see the file comment)
  uint32_t cs;       // entries in the chunk
  uint32_t tiles_x;  // 8x8 tiles per (virtual) row of tiles
  uint32_t total;    // path ids of the chunk: tiles * cs * 64
  uint32_t slots;    // path slots in use for this chunk
  uint32_t slot_span; // slots per band of the trace kernel's queue (a multiple of 64)
  uint32_t list_span; // entries per stripe of the live-slot list: the slots of the shade workgroups of one stripe
  uint32_t per_frame; // entries per frame: spp (keyed PCG) or 1 (Wang)
};

__device__ __forceinline__ void wf_pixel_of(const rt_render_params &pc, const WfChunk &ch, uint32_t id, uint32_t &x,
                                            uint32_t &y, uint32_t &entry) {
  const uint32_t in = id & 63u, ts = id >> 6; // ts = tile * cs + entry_in_chunk
  const uint32_t tile = ts / ch.cs;
  entry = ch.s0 + (ts - tile * ch.cs);
  x = (tile % ch.tiles_x) * 8u + (in & 7u);
  y = rt_row(pc, (tile / ch.tiles_x) * 8u + (in >> 3));
}

// Path id number p of workgroup blk's share of the chunk's ids: ids go out in groups of 64 (one 8x8 tile at one entry),
// the workgroup's j-th group is group blk + j * blocks of the chunk — its paths are spread over the whole image (equal
// work).  (Runs of 2 / 4 / 8 consecutive groups per workgroup, which keep a window's paths in one place of the scene:
// trace time unchanged, shade +2 ... 6 %, measured in round 3 — locality between the rays of a window is not what the
// trace pass lacks.)
__device__ __forceinline__ uint64_t wf_share_id(uint32_t blk, uint32_t blocks, uint32_t p) {
  return ((uint64_t)blk + (uint64_t)(p >> 6) * blocks) * 64u + (p & 63u);
}

// Seed of the reference's RNG for frame `frame` of the launch (RayTracing.slang:71-72 with frame_index + frame).
__device__ __forceinline__ uint32_t wf_wang_seed(const rt_render_params &pc, uint32_t x, uint32_t y, uint32_t frame) {
  return (x * 1973u) ^ (y * 9277u) ^ ((pc.frame_index + frame) * 26699u);
}

// Starts new paths in the slots of the lanes with want == true.  No global atomics (same-line atomics run at about
// 50 M/s device-wide; a first version with id counters shared between waves spent more time on them than on
// shading): every WORKGROUP owns a private, strided share of the chunk's id space — its j-th group of 64 ids is
// group blockIdx + j * gridDim of the chunk, so a workgroup's paths are spread over the whole image (equal work)
// while the ids one wave starts together are the pixels of one tile and sample (coherent camera rays).  The
// workgroup's position in its share lives in b.block_pos[blockIdx], touched by one thread per round.  Ids on tile
// padding outside the image are skipped (the lane draws again).  Returns true for lanes that wrote a camera ray.
// Must be called by all threads of the workgroup.
// The new path's ray is handed back in new_o / new_d (and its Wang state in new_seed): the caller stores it where the
// path is to live (k_wf_shade: the compacted position).
template <bool WANG>
__device__ __forceinline__ bool wf_regenerate(const rt_camera_uniforms &cam, const rt_render_params &pc,
                                              const WfChunk &ch, const rt_wf_buffers &b, bool want, float4 &new_o,
                                              float4 &new_d, uint32_t &new_seed) {
  __shared__ uint32_t s_cnt[WF_SHADE_BLOCK / 64], s_base;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
  bool started = false;
  while (true) {
    const unsigned long long mask = __ballot(want);
    if (lane == 0)
      s_cnt[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < WF_SHADE_BLOCK / 64; ++w) {
      before += w < wave ? s_cnt[w] : 0u;
      total += s_cnt[w];
    }
    if (total == 0)
      break; // uniform over the workgroup
    if (threadIdx.x == 0) {
      const uint32_t pos = b.block_pos[blockIdx.x];
      b.block_pos[blockIdx.x] = pos + total;
      s_base = pos;
    }
    __syncthreads();
    if (want) {
      const uint32_t p = s_base + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      const uint64_t id64 = wf_share_id(blockIdx.x, gridDim.x, p);
      if (id64 >= ch.total) {
        want = false; // the workgroup's share is used up (later positions lie even further out)
      } else {
        const uint32_t id = (uint32_t)id64;
        uint32_t x, y, entry;
        wf_pixel_of(pc, ch, id, x, y, entry);
        if (x < pc.image_width && y < pc.row_end) { // else tile padding: draw another id
          Rng rng = make_rng(pc, x, y);
          Path p;
          if (WANG) { // entry = frame: its first sample; the state goes on through the frame's bounces and samples
            rng.seed = wf_wang_seed(pc, x, y, entry);
            camera_ray(cam, recip_sqrt_spp, x, y, 0u, 0u, rng, p);
            new_seed = rng.seed;
          } else {      // entry = frame * spp + sample: the absolute sample index keys the draws
            const uint32_t sample = entry % spp;
            rng.sample = pc.frame_index * spp + entry;
            rng.set_slot(0);
            camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
          }
          new_o = make_float4(p.o.x, p.o.y, p.o.z, __uint_as_float(id));
          new_d = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(0u));
          want = false; // (attenuation 1 and sample radiance +0: WF_HAS_ATT / WF_HAS_RAD clear, nothing to store)
          started = true;
        }
      }
    }
    __syncthreads(); // s_cnt / s_base are rewritten by the next turn
  }
  return started;
}

// ---------------------------------------------------------------------------
// k_wf_shade: one lane per path slot; workgroup w works on the WINDOW of slots [w * 256, w * 256 + 256) for the whole
// chunk (no global lists, no global compaction: while the chunk has ids left every slot is refilled the moment its
// path ends, so all slots stay busy; only the last rounds thin out).  INIT: nothing to shade, every slot just starts a
// path (the first fill of a chunk).
//
// Where a path lives inside its window is free — a ray's trace does not depend on the lane that runs it — and the
// pass uses that (pass bit 30, "compaction"): the paths that go on are written to the front of the window in lane
// order, the camera rays of the new paths behind them, dead slots last.  A trace wave works through a window from its
// front (one share = one window), so it first holds scattered rays only and then the new paths' camera rays, which
// are the pixels of one or two 8x8 tiles at one sample: 64 coherent rays step through the same records in the same
// states (the triangle section of the step loop runs for all of them at once instead of for a dozen lanes), where the
// in-place scheme mixed 28 of them with 36 scattered rays in every wave.  All loads of the window happen before the
// first barrier and all stores after it; a wave's stores are runs of consecutive slots, so the traffic stays coalesced.
// Slots that cannot move keep their place: suspended rays (the pool record names the slot) and the slots beyond
// ch.slots in the last window.  Without the bit (and for the reference's RNG at spp > 1, whose per-slot sample sum
// would have to move along) every path stays in the slot of its lane.
// ---------------------------------------------------------------------------
template <bool INIT, bool WANG>
__global__ void __launch_bounds__(WF_SHADE_BLOCK)
    k_wf_shade(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc, rt_wf_buffers b, WfChunk ch, uint32_t pass) {
  constexpr uint32_t WAVES = WF_SHADE_BLOCK / 64;
  __shared__ uint32_t s_on[WAVES], s_pin[WAVES], s_new[WAVES], s_dead[WAVES], s_live[WAVES], s_at;
  __shared__ uint16_t s_free[WF_SHADE_BLOCK];
  __shared__ uint8_t s_was_dead[WF_SHADE_BLOCK];
  const uint32_t base = blockIdx.x * WF_SHADE_BLOCK, slot = base + threadIdx.x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool in_range = slot < ch.slots;
  const bool compact = (pass & 0x40000000u) != 0u;
  bool ended = false, goes_on = false, pinned = !in_range, was_dead = false;
  // what a path that goes on leaves in its slot
  float4 out_o = make_float4(0.f, 0.f, 0.f, 0.f), out_d = out_o, out_a = out_o, out_r = out_o, out_fs = out_o;
  bool put_att = false, put_rad = false, put_fs = false;
  uint32_t out_seed = 0;
  if (INIT) {
    ended = in_range;
  } else if (in_range) {
    const float4 rd = b.ray_d[slot];
    uint32_t depth = __float_as_uint(rd.w);
    if (depth != WF_DEAD && (depth & WF_SUSPENDED)) {
      goes_on = pinned = true; // the ray is still being traced (k_wf_trace resumes it next round): nothing to shade yet
    } else if (depth != WF_DEAD) {
      const bool has_rad = (depth & WF_HAS_RAD) != 0u, has_att = (depth & WF_HAS_ATT) != 0u;
      depth &= ~(WF_HAS_RAD | WF_HAS_ATT);
      const float4 ro = b.ray_o[slot], h = b.hit[slot];
      const float4 at = has_att ? b.att[slot] : make_float4(1.f, 1.f, 1.f, 0.f);
      const float4 ra = has_rad ? b.rad[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t id = __float_as_uint(ro.w);
      uint32_t x, y, entry;
      wf_pixel_of(pc, ch, id, x, y, entry);
      Rng rng = make_rng(pc, x, y);
      if (WANG) {
        rng.seed = b.seed[slot]; // the pixel's sequential state, where the bounce before left it
      } else {
        rng.sample = pc.frame_index * (pc.sqrt_spp * pc.sqrt_spp) + entry;
        rng.set_slot(depth + 1u);
      }
      Path p;
      p.o = v3(ro.x, ro.y, ro.z), p.d = v3(rd.x, rd.y, rd.z);
      p.attenuation = v3(at.x, at.y, at.z), p.sample_radiance = v3(ra.x, ra.y, ra.z);
      Hit hit;
      hit.t = h.x, hit.tri = __float_as_uint(h.y), hit.inst = __float_as_uint(h.z), hit.u = h.w;
      const bool any = hit.tri != RT_INVALID_ID;
      hit.v = any ? b.hit_v[slot] : 0.f;
      const bool scattered = shade(sc, pc.flags, any, hit, rng, p);
      ++depth;
      if (scattered && depth < pc.max_depth) {
        put_rad = (__float_as_uint(p.sample_radiance.x) | __float_as_uint(p.sample_radiance.y) |
                   __float_as_uint(p.sample_radiance.z)) != 0u; // anything but +0, +0, +0
        out_o = make_float4(p.o.x, p.o.y, p.o.z, ro.w);
        out_d = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(depth | WF_HAS_ATT | (put_rad ? WF_HAS_RAD : 0u)));
        out_a = make_float4(p.attenuation.x, p.attenuation.y, p.attenuation.z, 0.f), put_att = true;
        if (put_rad)
          out_r = make_float4(p.sample_radiance.x, p.sample_radiance.y, p.sample_radiance.z, 0.f);
        out_seed = rng.seed;
        goes_on = true;
      } else if (WANG) { // the sample ended: RayTracing.slang:170 in the slot, then the frame's next sample or its end
        const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
        V3 sum = v3(0.f, 0.f, 0.f);
        uint32_t s_idx = 0;
        if (spp > 1u) {
          const float4 fs = b.fsum[slot];
          sum = v3(fs.x, fs.y, fs.z), s_idx = __float_as_uint(fs.w);
        }
        sum = sum + p.sample_radiance;
        if (s_idx + 1u < spp) { // the next sample of this pixel and frame continues the RNG stream (:76-77)
          ++s_idx;
          Path np;
          camera_ray(cam, 1.f / (float)pc.sqrt_spp, x, y, s_idx % pc.sqrt_spp, s_idx / pc.sqrt_spp, rng, np);
          out_o = make_float4(np.o.x, np.o.y, np.o.z, ro.w);
          out_d = make_float4(np.d.x, np.d.y, np.d.z, __uint_as_float(0u));
          out_seed = rng.seed;
          out_fs = make_float4(sum.x, sum.y, sum.z, __uint_as_float(s_idx)), put_fs = true; // (spp > 1: no compaction)
          goes_on = true;
        } else {
          b.sr[id] = make_float4(sum.x, sum.y, sum.z, 0.f); // the frame's radiance
          ended = true;
        }
      } else { // RayTracing.slang:170: the sample's radiance, summed per pixel in sample order by k_wf_resolve
        b.sr[id] = make_float4(p.sample_radiance.x, p.sample_radiance.y, p.sample_radiance.z, 0.f);
        ended = true;
      }
    } else {
      was_dead = true;
    }
  }
  // ---- the paths that go on: to the front of the window (every load of the window above, every store below) ----
  const bool moves_on = goes_on && !pinned;
  const unsigned long long m_on = __ballot(moves_on), m_pin = __ballot(pinned);
  if (lane == 0)
    s_on[wave] = (uint32_t)__popcll(m_on), s_pin[wave] = (uint32_t)__popcll(m_pin);
  s_was_dead[threadIdx.x] = was_dead ? 1 : 0;
  __syncthreads();
  uint32_t on_before = 0, on_total = 0, pin_before = 0, pin_total = 0;
#pragma unroll
  for (uint32_t w = 0; w < WAVES; ++w) {
    on_before += w < wave ? s_on[w] : 0u, on_total += s_on[w];
    pin_before += w < wave ? s_pin[w] : 0u, pin_total += s_pin[w];
  }
  const bool direct = !compact || pin_total == 0u; // position k of the window is slot base + k
  if (compact && pin_total != 0u) { // (uniform) the k-th slot that may be written: s_free[k]
    if (!pinned)
      s_free[threadIdx.x - pin_before - (uint32_t)__popcll(m_pin & below)] = (uint16_t)threadIdx.x;
    __syncthreads();
  }
  auto place = [&](uint32_t k) { return !compact ? slot : direct ? base + k : base + s_free[k]; };
  uint32_t dst = slot; // where this lane's path lives after the pass
  if (moves_on) {
    dst = place(on_before + (uint32_t)__popcll(m_on & below));
    b.ray_o[dst] = out_o;
    b.ray_d[dst] = out_d;
    if (put_att)
      b.att[dst] = out_a;
    if (put_rad)
      b.rad[dst] = out_r;
    if (WANG)
      b.seed[dst] = out_seed;
    if (put_fs)
      b.fsum[dst] = out_fs;
  }
  // ---- new paths behind them, dead slots last ----
  float4 new_o = make_float4(0.f, 0.f, 0.f, 0.f), new_d = new_o;
  uint32_t new_seed = 0;
  const bool started = wf_regenerate<WANG>(cam, pc, ch, b, ended, new_o, new_d, new_seed);
  const bool dead = !pinned && !moves_on && !started; // ended without a successor, or dead before
  const unsigned long long m_new = __ballot(started), m_dead = __ballot(dead);
  if (lane == 0)
    s_new[wave] = (uint32_t)__popcll(m_new), s_dead[wave] = (uint32_t)__popcll(m_dead);
  // Live paths after this pass: one atomic per workgroup on its stripe's counter (the host sums the stripes), and what
  // the atomic returns places the workgroup's live slots in the stripe's segment of the live-slot list, which the
  // trace pass walks instead of the slots once most of them are dead.
  const bool live = goes_on || started;
  const unsigned long long lm = __ballot(live);
  if (lane == 0)
    s_live[wave] = (uint32_t)__popcll(lm);
  __syncthreads();
  uint32_t new_before = 0, new_total = 0, dead_before = 0, before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < WAVES; ++w) {
    new_before += w < wave ? s_new[w] : 0u, new_total += s_new[w];
    dead_before += w < wave ? s_dead[w] : 0u;
    before += w < wave ? s_live[w] : 0u, total += s_live[w];
  }
  if (started) {
    dst = place(on_total + new_before + (uint32_t)__popcll(m_new & below));
    b.ray_o[dst] = new_o;
    b.ray_d[dst] = new_d;
    if (WANG) {
      b.seed[dst] = new_seed;
      if (pc.sqrt_spp > 1u)
        b.fsum[dst] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
    }
  } else if (dead) {
    dst = place(on_total + new_total + dead_before + (uint32_t)__popcll(m_dead & below));
    // (a position that was dead before the pass needs no second mark: without compaction that is the lane's own)
    if (!s_was_dead[dst - base])
      b.ray_d[dst] = make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_DEAD));
  }
  if (total == 0)
    return; // uniform over the workgroup
  const uint32_t stripe = blockIdx.x % WF_STRIPES;
  if (threadIdx.x == 0)
    s_at = atomicAdd(b.ctrl + wf_alive_set(pass) + stripe * WF_PAD, total);
  __syncthreads();
  if (live && (pass >> 31)) // (bit 31: the host wants the list — not while every slot is refilled anyway)
    b.live[(size_t)stripe * ch.list_span + s_at + before + (uint32_t)__popcll(lm & below)] = dst;
}

// ---------------------------------------------------------------------------
// PERSIST mode of k_wf_trace: no rounds (experiments/r03_persistent/ has the measurements).  Every resident wave owns a
// window of WP_WINDOW path slots for the whole launch and loops: fill the free positions with new paths (camera rays
// of whole 64-id groups, drawn from striped counters), trace the window's rays (the step loop, lanes refilling from
// the window, stragglers suspended), shade the window (one lane per slot, 64 at a time; paths that go on are written
// to the window's front in place).  The path state never leaves its wave (277 MB for the whole GPU instead of the
// rounds' 5.9 GB streaming through HBM twice per round), there is no kernel boundary per round and the host never waits.
// ---------------------------------------------------------------------------
// slots per wave.  C3, 256 spp, ms per launch (full frame / one rank's eighth): 128 254.4 / 32.7, 256 232.7 / 30.4,
// 512 226.6 / 29.9, 1024 225.2 / 31.7 — against 242.5 / 36.4 for the rounds (experiments/r03_persistent/)
#define WP_WINDOW 512u
#define WP_NO_GROUP 0xFFFFFFFFu

// Path-state accesses of the fill and shade phases are NON-TEMPORAL: the state of a window (43 KB) comes back to its
// wave long after the L2 has forgotten it (768 windows per XCD against 4 MB), so caching it there only displaces node
// records — with the persistent mode's shade phases running beside other waves' trace phases the L2 hit rate fell from
// 65 to 57 %.  C3, 256 spp: 227.8 -> 220.6 ms with the hint on every state load and store here (on the loads and the
// result stores only: 222.4; also on the trace phase's ray loads and hit stores: 223.0 — those stay plain).  A scene
// that fits the L2s has no node records to protect and its windows' state does come back from the L2: there the hint
// costs 3-4 % (C1 10119 -> 9827, C2 13705 -> 13204 Mrays/s), so it is a launch argument (`nt`, wave-uniform), set for
// scenes of more than WP_NT_SCENE_BYTES of pair records and triangles.
#define WP_NT_SCENE_BYTES (32u << 20)
typedef float wp_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wp_ld4(const float4 *p, bool nt) {
  if (!nt)
    return *p;
  wp_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wp_f4 *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void wp_st4(float4 *p, float4 a, bool nt) {
  if (!nt) {
    *p = a;
    return;
  }
  wp_f4 v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, reinterpret_cast<wp_f4 *>(p));
}
__device__ __forceinline__ float wp_ld1(const float *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

struct WpState {
  uint32_t base = 0;     // first slot of the wave's window
  uint32_t n_on = 0;     // paths at the window's front that go on (after a shade)
  uint32_t n_live = 0;   // slots in use (after a fill)
  uint32_t stripe = 0;   // the striped group counter this wave draws from
  uint32_t pend_group = 0;
  unsigned long long pend_mask = 0;  // lanes of pend_group whose paths have not been started yet
  unsigned long long used_seen = 0;  // stripes known to be used up
  uint32_t ids_left = 1;
};

// the next group of 64 path ids (one 8x8 tile at one entry), or WP_NO_GROUP.  The 64 striped counters are eight per
// class of workgroups that share an XCD and with it an L2 (blockIdx % 8; MI355X_MICROARCH: blocks are dealt round-robin
// over the XCDs).  Two ways to deal the groups to the classes, chosen per launch:
//  * interleaved (BANDS false): stripe s hands out groups s, s + 64, ... — every class works on the same tiles at the
//    same time (an eighth of their samples each) and the launch's cost spreads evenly whatever the image looks like;
//  * BANDS: class c works through its own eighth of the groups, a band of tile rows (stripe 8 c + j: groups
//    c * per + j, + 8, ...), so the eight L2s hold the records of eight parts of the scene instead of eight copies of
//    one: C3 at 256 spp 220.1 -> 217.7 ms (L2 hit rate 58.1 -> 59.3 %, FETCH_SIZE 498 -> 464 GB).  But bands of
//    unequal cost only level out by waves moving on to the next class's stripes, and a launch with few groups per wave
//    has nothing left to move to: C3 at 1 / 4 / 16 spp 2.9 -> 4.0 / 6.0 -> 6.8 / 16.1 -> 16.4 ms; and a scene that fits
//    the L2s has nothing to gain (C2, 256 spp: 51.9 -> 54.0 ms).  Pieces of 1 ... 2048 tiles dealt round-robin to the
//    classes (tools/sweep_band_tiles.sh on an experimental build) were no better than interleaved.
// So bands are for launches of at least WP_BANDS_MIN_PATHS paths over a scene of more than WP_NT_SCENE_BYTES.  The
// choice is a template parameter because as a launch argument it cost what it gained (one more scalar kept across
// the step loop: 220.4 against 218.4 ms).
#define WP_BANDS_MIN_PATHS 100000000ull
#define WP_KIND_BANDS 4 // added to the kernel's WANG parameter (kinds 0 and 1)
template <bool BANDS>
__device__ __forceinline__ uint32_t wp_draw_group(const rt_wf_buffers &b, const WfChunk &ch, WpState &wp) {
  const uint32_t groups = ch.total >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  while (true) {
    const unsigned long long avail = ~wp.used_seen;
    if (avail == 0ull)
      return WP_NO_GROUP;
    const unsigned long long rot = wp.stripe ? ((avail >> wp.stripe) | (avail << (64u - wp.stripe))) : avail;
    wp.stripe = (wp.stripe + (uint32_t)__ffsll((long long)rot) - 1u) & 63u;
    uint32_t k = 0;
    if (lane == 0)
      k = atomicAdd(b.ctrl + WF_HEAD + wp.stripe * WF_PAD, 1u);
    k = (uint32_t)__builtin_amdgcn_readlane((int)k, 0);
    if (BANDS) {
      const uint32_t per = (groups + 7u) / 8u, c = wp.stripe >> 3;
      const uint64_t end = (uint64_t)(c + 1u) * per < groups ? (uint64_t)(c + 1u) * per : groups;
      const uint64_t g = (uint64_t)c * per + (wp.stripe & 7u) + (uint64_t)k * 8u;
      if (g < end)
        return (uint32_t)g;
    } else {
      const uint64_t g = (uint64_t)wp.stripe + (uint64_t)k * WF_STRIPES;
      if (g < groups)
        return (uint32_t)g;
    }
    wp.used_seen |= 1ull << wp.stripe;
  }
}

template <int WANG, bool BANDS> // 0: keyed PCG; 1: the reference's RNG at one sample per frame; 2: at more (the slot carries the frame's sum)
__device__ __forceinline__ void wp_fill(const rt_camera_uniforms &cam, const rt_render_params &pc, const WfChunk &ch,
                                        const rt_wf_buffers &b, WpState &wp, bool nt) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
  uint32_t fill = wp.n_on;
  while (fill < WP_WINDOW) {
    if (wp.pend_mask == 0ull) {
      if (!wp.ids_left)
        break;
      const uint32_t g = wp_draw_group<BANDS>(b, ch, wp);
      if (g == WP_NO_GROUP) {
        wp.ids_left = 0u;
        break;
      }
      wp.pend_group = g;
      uint32_t x, y, entry;
      wf_pixel_of(pc, ch, g * 64u + lane, x, y, entry);
      const unsigned long long m = __ballot(x < pc.image_width && y < pc.row_end); // (else tile padding)
      wp.pend_mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32);
      continue;
    }
    const uint32_t room = WP_WINDOW - fill;
    const uint32_t rank = (uint32_t)__popcll(wp.pend_mask & below);
    const bool go = ((wp.pend_mask >> lane) & 1ull) != 0ull && rank < room;
    if (go) {
      const uint32_t id = wp.pend_group * 64u + lane;
      uint32_t x, y, entry;
      wf_pixel_of(pc, ch, id, x, y, entry);
      Rng rng = make_rng(pc, x, y);
      Path p;
      const uint32_t dst = wp.base + fill + rank;
      if (WANG) { // entry = frame: its first sample; the state goes on through the frame's bounces and samples
        rng.seed = wf_wang_seed(pc, x, y, entry);
        camera_ray(cam, recip_sqrt_spp, x, y, 0u, 0u, rng, p);
        b.seed[dst] = rng.seed;
        if (WANG == 2)
          b.fsum[dst] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0u));
      } else {    // entry = frame * spp + sample: the absolute sample index keys the draws
        const uint32_t sample = entry % spp;
        rng.sample = pc.frame_index * spp + entry;
        rng.set_slot(0);
        camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
      }
      wp_st4(b.ray_o + dst, make_float4(p.o.x, p.o.y, p.o.z, __uint_as_float(id)), nt);
      wp_st4(b.ray_d + dst, make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(0u)), nt);
    }
    const unsigned long long went = __ballot(go);
    fill += (uint32_t)__popcll(went);
    wp.pend_mask &= ~went;
    wp.pend_mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wp.pend_mask) |
                   ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wp.pend_mask >> 32)) << 32);
  }
  wp.n_live = (uint32_t)__builtin_amdgcn_readfirstlane((int)fill);
}

// shades the window's traced rays, 64 slots at a time (the arithmetic of k_wf_shade); the paths that go on (and the
// suspended rays, untouched) are written to the front of the window IN PLACE: position n_on + rank <= the slot being
// read, and a batch's loads are complete before its stores.  Everything a path owns moves with it: ray, attenuation,
// radiance, and — reference's RNG — its state and (spp > 1) the frame's sample sum; a suspended ray also takes the
// index of its pool record (hit_v) along.
template <int WANG>
__device__ __forceinline__ void wp_shade(const rt_dev_scene &sc, const rt_camera_uniforms &cam, const rt_render_params &pc,
                                         const rt_wf_buffers &b, const WfChunk &ch, WpState &wp, bool nt) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
  constexpr bool sums = WANG == 2; // the slot carries the frame's radiance over the samples done so far
  uint32_t n_on = 0;
  for (uint32_t g0 = 0; g0 < wp.n_live; g0 += 64u) {
    const uint32_t slot = wp.base + g0 + lane;
    bool goes_on = false, put_att = false, put_rad = false, put_v = false;
    float4 out_o = make_float4(0.f, 0.f, 0.f, 0.f), out_d = out_o, out_a = out_o, out_r = out_o, out_fs = out_o;
    float out_v = 0.f;
    uint32_t out_seed = 0;
    if (g0 + lane < wp.n_live) {
      const float4 rd = wp_ld4(b.ray_d + slot, nt);
      uint32_t depth = __float_as_uint(rd.w);
      const bool has_rad = (depth & WF_HAS_RAD) != 0u, has_att = (depth & WF_HAS_ATT) != 0u;
      const float4 ro = wp_ld4(b.ray_o + slot, nt);
      const float4 at = has_att ? wp_ld4(b.att + slot, nt) : make_float4(1.f, 1.f, 1.f, 0.f);
      const float4 ra = has_rad ? wp_ld4(b.rad + slot, nt) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (WANG)
        out_seed = b.seed[slot];
      if (sums)
        out_fs = b.fsum[slot];
      if (depth & WF_SUSPENDED) { // still being traced: moves along as it is, with the index of its pool record
        goes_on = true;
        out_o = ro, out_d = rd, out_a = at, out_r = ra, put_att = has_att, put_rad = has_rad;
        out_v = b.hit_v[slot], put_v = true;
      } else {
        depth &= ~(WF_HAS_RAD | WF_HAS_ATT);
        const float4 h = wp_ld4(b.hit + slot, nt);
        const uint32_t id = __float_as_uint(ro.w);
        uint32_t x, y, entry;
        wf_pixel_of(pc, ch, id, x, y, entry);
        Rng rng = make_rng(pc, x, y);
        if (WANG) {
          rng.seed = out_seed; // the pixel's sequential state, where the bounce before left it
        } else {
          rng.sample = pc.frame_index * spp + entry;
          rng.set_slot(depth + 1u);
        }
        Path p;
        p.o = v3(ro.x, ro.y, ro.z), p.d = v3(rd.x, rd.y, rd.z);
        p.attenuation = v3(at.x, at.y, at.z), p.sample_radiance = v3(ra.x, ra.y, ra.z);
        Hit hit;
        hit.t = h.x, hit.tri = __float_as_uint(h.y), hit.inst = __float_as_uint(h.z), hit.u = h.w;
        const bool any = hit.tri != RT_INVALID_ID;
        hit.v = any ? wp_ld1(b.hit_v + slot, nt) : 0.f;
        const bool scattered = shade(sc, pc.flags, any, hit, rng, p);
        ++depth;
        if (scattered && depth < pc.max_depth) {
          put_rad = (__float_as_uint(p.sample_radiance.x) | __float_as_uint(p.sample_radiance.y) |
                     __float_as_uint(p.sample_radiance.z)) != 0u; // anything but +0, +0, +0
          out_o = make_float4(p.o.x, p.o.y, p.o.z, ro.w);
          out_d = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(depth | WF_HAS_ATT | (put_rad ? WF_HAS_RAD : 0u)));
          out_a = make_float4(p.attenuation.x, p.attenuation.y, p.attenuation.z, 0.f), put_att = true;
          out_r = make_float4(p.sample_radiance.x, p.sample_radiance.y, p.sample_radiance.z, 0.f);
          out_seed = rng.seed;
          goes_on = true;
        } else if (WANG) { // the sample ended: RayTracing.slang:170 in the slot, then the frame's next sample or its end
          V3 sum = v3(out_fs.x, out_fs.y, out_fs.z); // (+0 at spp 1)
          uint32_t s_idx = __float_as_uint(out_fs.w);
          sum = sum + p.sample_radiance;
          if (sums && s_idx + 1u < spp) { // the next sample of this pixel and frame continues the RNG stream (:76-77)
            ++s_idx;
            Path np;
            camera_ray(cam, 1.f / (float)pc.sqrt_spp, x, y, s_idx % pc.sqrt_spp, s_idx / pc.sqrt_spp, rng, np);
            out_o = make_float4(np.o.x, np.o.y, np.o.z, ro.w);
            out_d = make_float4(np.d.x, np.d.y, np.d.z, __uint_as_float(0u));
            out_seed = rng.seed;
            out_fs = make_float4(sum.x, sum.y, sum.z, __uint_as_float(s_idx));
            goes_on = true;
          } else {
            wp_st4(b.sr + id, make_float4(sum.x, sum.y, sum.z, 0.f), nt); // the frame's radiance
          }
        } else { // RayTracing.slang:170: the sample's radiance, summed per pixel in sample order by k_wf_resolve
          wp_st4(b.sr + id, make_float4(p.sample_radiance.x, p.sample_radiance.y, p.sample_radiance.z, 0.f), nt);
        }
      }
    }
    const unsigned long long m_on = __ballot(goes_on);
    if (goes_on) {
      const uint32_t dst = wp.base + n_on + (uint32_t)__popcll(m_on & below);
      wp_st4(b.ray_o + dst, out_o, nt);
      wp_st4(b.ray_d + dst, out_d, nt);
      if (put_att)
        wp_st4(b.att + dst, out_a, nt);
      if (put_rad)
        wp_st4(b.rad + dst, out_r, nt);
      if (put_v)
        b.hit_v[dst] = out_v;
      if (WANG)
        b.seed[dst] = out_seed;
      if (sums)
        b.fsum[dst] = out_fs;
    }
    n_on += (uint32_t)__popcll(m_on);
  }
  wp.n_on = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_on);
}

// ---------------------------------------------------------------------------
// k_wf_trace: persistent wave64s; a lane traces one ray at a time and takes the next live slot when it is done.  The
// step loop is k_render_v8's (vote scheduling, joined pair / leaf steps, exact fast slab division); what a lane carries
// besides it is only its slot and the world-space ray.  Slots are handed out in chunks of WF_GRAB consecutive slots
// per atomic from striped queue heads.
// ---------------------------------------------------------------------------
enum : uint32_t { WT_FETCH = 0, WT_TRAVERSE = 2, WT_FINISHED = 3, WT_EXIT = 4 };
// Slots per share.  Two costs pull against each other.  A share is what a wave works through alone: 256 slots are 4
// rays per lane one after the other, about 0.4 ms, and the waves of a round finish up to that far apart (measured
// with per-wave clocks, experiments/r02_trace_rounds/: with 512 the average wave of a 2 ms round was busy 76 % of it).
// A SMALL share costs more than its atomic (two round trips at device scope, about 6 us): the lanes of a wave then
// hold rays of several shares from different parts of the image, and the step loop itself slows down by 15 % at 64
// — neighbouring slots are neighbouring pixels of one tile, and their rays share node records.  Measured 64 ... 4096 on
// full frames and on shards (tools/bench_shard.py): 256 is best on all (512: -0.7 ... -4 %, 192 / 384: -1 ... -2 %,
// 128: -5 %, 64: -17 %); shares that shrink towards the end of a band (guided self-scheduling) lost 2 ... 25 %, and so
// did asking for the next share in advance (a wave then holds two) and skipping the look at the used-up bits.
#define WF_GRAB 256u

//
// COOP: quad-cooperative record fetch.  A lane that gathers its own 64-byte record with four 16-byte loads costs the
// CU's texture path one cache-line lookup per lane and instruction: 256 per wave step, which is where the per-lane
// gather saturates (k_probe_walk: 139 G record visits/s chip-wide; this kernel ran at 115 G).  Here the four lanes of a
// quad fetch ONE lane's record per instruction, 16 bytes each — one line lookup per quad, 64 per wave step (the probe
// reaches 270-300 G visits/s this way) — and the four instructions serve the quad's four records.  The data cannot
// stay in the fetching lanes' registers (a lane would hold pieces of four different records), so the loads are
// LDS-DMA (global_load_lds_dwordx4: destination = M0 + 16 * lane): instruction j drops the complete record of quad
// q's lane j at staging + j * WF_COOP_STRIDE + 64 * q, and every lane reads its own record back with ds_read_b128.
// WF_COOP_STRIDE = 1040 = 1024 + 16 keeps the four lanes of a quad on different banks.  Instruction j runs only for
// the quads whose lane j takes a step (EXEC from a DPP compare), addresses are 32-bit offsets from one base
// (rt_wf_geom) so that a DPP add forms them in one instruction.
#define WF_COOP_STRIDE 1040u
#define WF_COOP_BYTES (4u * WF_COOP_STRIDE) // per wave

template <int MODE, int MINW, int BLOCK, int NLDS, bool COOP, int NCACHE, bool PERSIST, int WANG>
__global__ void __launch_bounds__(BLOCK, MINW)
    k_wf_trace(rt_dev_scene sc, rt_wf_geom geom, rt_wf_buffers b, WfChunk ch, unsigned long long *ray_counter,
               uint32_t regen_idle, uint32_t sched_arg, uint32_t susp_arg, rt_camera_uniforms cam, rt_render_params pc) {
  extern __shared__ __align__(16) uint32_t lds_raw[];
  uint32_t *lds_stack = lds_raw; // [NLDS][BLOCK]
  typedef float f4v __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const f4v *lds_f4_ptr;
  RT_STACK_DECL(BLOCK, NLDS, st, lds_stack);
  const uint32_t lane = threadIdx.x & 63u;
  // staging area of this wave behind the stacks; M0 values (LDS byte addresses) of the four instructions
  uint32_t *stage_words = lds_raw + NLDS * BLOCK + (threadIdx.x >> 6) * (WF_COOP_BYTES / 4u);
  const uint32_t stage_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_u32_ptr)stage_words);
  const lds_f4_ptr my_rec = (lds_f4_ptr)(lds_u32_ptr)(stage_words + (lane & 3u) * (WF_COOP_STRIDE / 4u) + (lane >> 2) * 16u);
  const uint32_t sub16 = (lane & 3u) * 16u;
  // NCACHE: the first records of the pair array — the breadth-first top of the scene (rt_hip_api.cpp), which every ray
  // walks through — copied to LDS behind the stacks once per workgroup; a pair step of a lane whose record is among
  // them reads LDS instead of gathering through the L1 (WF_CACHE above the kernel).
  static_assert(!(COOP && NCACHE), "the staging area and the record cache share the space behind the stacks");
  const uint32_t n_cached = NCACHE ? (sc.cache_pairs < (uint32_t)NCACHE ? sc.cache_pairs : (uint32_t)NCACHE) : 0u;
  const uint32_t cache_lds = (uint32_t)(uintptr_t)(lds_u32_ptr)(lds_raw + NLDS * BLOCK);
  if (NCACHE) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sc.pairs);
    uint4 *dst = reinterpret_cast<uint4 *>(lds_raw + NLDS * BLOCK);
    for (uint32_t i = threadIdx.x; i < n_cached * 4u; i += BLOCK)
      dst[i] = src[i];
    __syncthreads();
  }

  uint32_t state = WT_FETCH, slot = 0, traced = 0;
  V3 wo = v3(0.f, 0.f, 0.f), wd = v3(0.f, 0.f, 0.f); // the ray in world space
  uint32_t ref = RT_REF_NONE, cur_inst = 0;
  V3 co = v3(0.f, 0.f, 0.f), cd = v3(0.f, 0.f, 0.f), cinv = v3(0.f, 0.f, 0.f);
  const bool scene_ok = (sc.flags & RT_SCENE_BOXES_MODERATE) != 0u && !(sched_arg & 0x8000u);
  float closest = 0.f, hit_u = 0.f, hit_v = 0.f;
  uint32_t hit_tri = 0, hit_inst = 0;
  uint32_t anyv = 0, fastv = 0;
  // the wave's share of the slot space (wave-uniform): [grab_next, grab_end), refilled from the band heads
  // (readfirstlane / readlane below: these values are the same in all lanes, and saying so keeps the whole share
  // bookkeeping in scalar registers — 7 vector registers less)
  uint32_t grab_next = 0, grab_end = 0;
  uint32_t band = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) % WF_STRIPES));
  // suspension (WF_SUSPENDED): susp_arg bits 0-7 = a wave with this many rays or fewer left and no fresh slots gives
  // them up (0 = never), bit 8 = which of the two record pools this round writes (it reads the other)
  // A wave suspends only after WF_SUSP_STEPS traversal steps of its own: every round then moves some ray forward, so
  // the rounds end (a wave handed three resumed rays and nothing else would otherwise give them up again at once).
  uint32_t susp_min = susp_arg & 0xFFu, q_empty = 0, steps = 0; // all wave-uniform
  const uint32_t share = susp_arg >> 16; // bits 16-31: slots per share
  // bit 9: walk the live-slot list of the shade pass before (band = stripe, its length from that pass's counters,
  // bit 10 = which set) instead of the slots themselves
  const bool by_list = (susp_arg & 0x200u) != 0u;
  const uint32_t *const list_len = b.ctrl + wf_alive_set((susp_arg >> 10) & 1u);
  const uint32_t band_span = by_list ? ch.list_span : ch.slot_span;
  // bands this wave knows to be used up (those beyond the slot space from the start)
  const uint32_t n_bands = by_list ? WF_STRIPES : (ch.slots + ch.slot_span - 1u) / ch.slot_span;
  unsigned long long used_seen = n_bands >= 64u ? 0ull : ~((1ull << n_bands) - 1ull);
  // (PERSIST: a wave resumes in its next trace phase what it suspended in the last one — one pool)
  uint32_t *const pool_w = b.pool + (PERSIST ? (size_t)0 : (size_t)((susp_arg >> 8) & 1u) * b.pool_records * WF_REC_WORDS);
  const uint32_t *const pool_r = b.pool + (PERSIST ? (size_t)0 : (size_t)(((susp_arg >> 8) & 1u) ^ 1u) * b.pool_records * WF_REC_WORDS);
  // PERSIST: the wave owns the window of WP_WINDOW slots at wp_base for the whole launch and alternates between
  // filling it, tracing it and shading it (wp_fill / wp_shade below)
  WpState wp;
  const bool wp_nt = (sched_arg & 0x10000u) != 0u; // (PERSIST: non-temporal state accesses in fill and shade)
  wp.base = (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * WP_WINDOW;
  static_assert(WF_STRIPES == 64, "eight stripes for each of the eight XCD classes");
  wp.stripe = (blockIdx.x % 8u) * 8u + ((blockIdx.x / 8u) * (BLOCK / 64) + (threadIdx.x >> 6)) % 8u;

  const uint32_t w_pair = (sched_arg & 0xFu) ? (sched_arg & 0xFu) : 1u;
  const uint32_t w_leaf = ((sched_arg >> 4) & 0xFu) ? ((sched_arg >> 4) & 0xFu) : 2u;
  const uint32_t w_inst = ((sched_arg >> 8) & 0xFu) ? ((sched_arg >> 8) & 0xFu) : 3u;
  const uint32_t join_code = (sched_arg >> 12) & 7u;
  const uint32_t join_min = join_code == 0u ? 8u : join_code == 1u ? 1u : join_code == 2u ? 4u : join_code == 3u ? 6u
                          : join_code == 4u ? 12u : join_code == 5u ? 16u : join_code == 6u ? 24u : 65u;

  if (PERSIST && blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6) >= share) // (PERSIST: bits 16-31 = windows of the launch)
    return;
  do { // (PERSIST: one turn per fill / trace / shade of the window; otherwise once)
  if (PERSIST) {
    wp_fill<(WANG & 3), (WANG & WP_KIND_BANDS) != 0>(cam, pc, ch, b, wp, wp_nt);
    if (wp.n_live == 0u)
      break;
    grab_next = wp.base, grab_end = wp.base + wp.n_live;
    q_empty = 0u, steps = 0u, susp_min = susp_arg & 0xFFu;
    // every lane starts the phase without a ray: nothing of the last phase is live (said explicitly, or the
    // traversal state would occupy registers through the fill and the shade)
    state = WT_FETCH, slot = 0;
    wo = wd = co = cd = cinv = v3(0.f, 0.f, 0.f);
    ref = RT_REF_NONE, cur_inst = 0, closest = hit_u = hit_v = 0.f, hit_tri = hit_inst = anyv = fastv = 0u;
    st.sp = 0;
  }
  while (true) {
    // ---- (1) finished rays: the hit record goes to the slot ----
    if (state == WT_FINISHED) {
      b.hit[slot] = make_float4(closest, __uint_as_float(anyv ? hit_tri : RT_INVALID_ID), __uint_as_float(hit_inst), hit_u);
      if (anyv)
        b.hit_v[slot] = hit_v;
      state = WT_FETCH;
    }
    // ---- (2) lanes without a ray: next live slots (wave-uniform loop; dead slots are skipped) ----
    while (true) {
      const bool need = state == WT_FETCH;
      const unsigned long long mask = __ballot(need);
      if (mask == 0ull)
        break;
      if (PERSIST && grab_next >= grab_end) { // the window has been handed out
        if (need)
          state = WT_EXIT;
        q_empty = 1u;
        break;
      }
      if (!PERSIST && grab_next >= grab_end) { // the wave's share is used up: `share` more slots from a band head
        // which bands have slots left: one 64-bit word, a bit per band that a wave found used up (a look at the 64
        // heads themselves cost 64 reads at device scope per share, as much as the rounds' tails)
        // (looking only after the wave's own band has run out — one round trip per share instead of two — measured
        // 4-12 % SLOWER: kept as it is)
        const unsigned long long seen_now = __hip_atomic_load(reinterpret_cast<unsigned long long *>(b.ctrl + WF_USED),
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        used_seen |= (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seen_now) |
                     ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seen_now >> 32)) << 32);
        const unsigned long long avail = ~used_seen;
        if (avail == 0ull) {
          if (need)
            state = WT_EXIT;
          q_empty = 1u;
          break;
        }
        const unsigned long long rot = band ? ((avail >> band) | (avail << (64u - band))) : avail;
        band = (band + (uint32_t)__ffsll((long long)rot) - 1u) & 63u;
        const uint32_t lo = band * band_span;
        const uint32_t hi = by_list ? lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)list_len[band * WF_PAD])
                                    : lo + band_span < ch.slots ? lo + band_span : ch.slots;
        uint32_t base = 0;
        if (lane == 0)
          base = atomicAdd(b.ctrl + WF_HEAD + band * WF_PAD, share);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, 0);
        if (base >= hi - lo) { // the band is used up: say so, look again
          used_seen |= 1ull << band;
          if (lane == 0)
            atomicOr(reinterpret_cast<unsigned long long *>(b.ctrl + WF_USED), 1ull << band);
          continue;
        }
        grab_next = lo + base;
        grab_end = grab_next + share < hi ? grab_next + share : hi;
      }
      const uint32_t take = (uint32_t)__popcll(mask);
      if (need) {
        const uint32_t at = grab_next + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (at < grab_end) {
          const uint32_t s = by_list ? b.live[at] : at;
          // both halves of the ray at once (the origin of a dead slot is read for nothing: a second round trip for
          // every live one costs more)
          const float4 rd = b.ray_d[s];
          float4 ro = b.ray_o[s];
          asm volatile("" : "+v"(ro.x), "+v"(ro.y), "+v"(ro.z));
          if (__float_as_uint(rd.w) != WF_DEAD) {
            slot = s;
            wo = v3(ro.x, ro.y, ro.z), wd = v3(rd.x, rd.y, rd.z);
            co = wo, cd = wd;
            if (__float_as_uint(rd.w) & WF_SUSPENDED) { // a ray the round before gave up: take its state back
              const uint32_t *rec = pool_r + (size_t)__float_as_uint(b.hit_v[s]) * WF_REC_WORDS;
              const uint4 ra = *reinterpret_cast<const uint4 *>(rec), rb = *reinterpret_cast<const uint4 *>(rec + 4);
              ref = ra.x, cur_inst = ra.y & 0x3FFFFFFFu, anyv = ra.y >> 31;
              closest = __uint_as_float(ra.z), hit_u = __uint_as_float(ra.w), hit_v = __uint_as_float(rb.x);
              hit_tri = rb.y, hit_inst = rb.z;
              st.sp = 0;
              for (uint32_t k = 0; k < rb.w; ++k)
                st.push(rec[8u + k]);
              if (ra.y & 0x40000000u) { // it was inside an instance: the same object-space ray as on entry
                const float4 *q = reinterpret_cast<const float4 *>(sc.inst_trace + cur_inst);
                float4 a = q[0], bb = q[1], c = q[2], e = q[3];
                V3 c0 = v3(a.x, a.y, a.z), c1 = v3(bb.x, bb.y, bb.z), c2 = v3(c.x, c.y, c.z), c3 = v3(e.x, e.y, e.z);
                co = mul_cols(c0, c1, c2, c3, wo, 1.f), cd = mul_cols(c0, c1, c2, c3, wd, 0.f);
              }
              reinterpret_cast<uint32_t *>(b.ray_d + s)[3] = __float_as_uint(rd.w) & ~WF_SUSPENDED;
              state = WT_TRAVERSE;
            } else {
              ref = sc.root_ref;
              closest = 1000.f; // RayTracing.slang:93-95
              anyv = 0u;
              cur_inst = 0;
              st.sp = 0;
              ++traced;
              state = ref == RT_REF_NONE ? WT_FINISHED : WT_TRAVERSE;
            }
            bool f2;
            prep_space<MODE>(co, cd, scene_ok, cinv, f2);
            fastv = f2 ? 1u : 0u;
          }
        }
      }
      grab_next += take;
    }
    susp_min = (uint32_t)__builtin_amdgcn_readfirstlane((int)susp_min);
    if (q_empty && susp_min && steps >= WF_SUSP_STEPS) { // no fresh slots any more: every lane is traversing or done
      const bool live = state == WT_TRAVERSE;
      const unsigned long long lm = __ballot(live);
      if (lm != 0ull && (uint32_t)__popcll(lm) <= susp_min) {
        if (__ballot(live && st.sp > (int)WF_REC_STACK) != 0ull) {
          susp_min = 0; // a stack deeper than a record: this wave sees its rays through
        } else if (live) {
          uint32_t tid = threadIdx.x;
          asm volatile("" : "+v"(tid)); // keeps the record address out of the registers the step loop needs
          const uint32_t r = blockIdx.x * BLOCK + tid;
          uint32_t *rec = pool_w + (size_t)r * WF_REC_WORDS;
          uint32_t inside = 0;
          for (int k = 0; k < st.sp; ++k) {
            const uint32_t e = k < NLDS ? st.lds[k * BLOCK] : st.spill[k - NLDS];
            inside |= e == RT_REF_SENTINEL ? 0x40000000u : 0u;
            rec[8 + k] = e;
          }
          *reinterpret_cast<uint4 *>(rec) = make_uint4(ref, cur_inst | inside | (anyv << 31), __float_as_uint(closest), __float_as_uint(hit_u));
          *reinterpret_cast<uint4 *>(rec + 4) = make_uint4(__float_as_uint(hit_v), hit_tri, hit_inst, (uint32_t)st.sp);
          b.hit_v[slot] = __uint_as_float(r);
          uint32_t *dw = reinterpret_cast<uint32_t *>(b.ray_d + slot) + 3;
          *dw = *dw | WF_SUSPENDED;
          state = WT_EXIT;
          if (lane == (uint32_t)__ffsll((long long)lm) - 1u)
            atomicAdd(ray_counter + 1, (unsigned long long)__popcll(lm)); // statistics: rays suspended
        }
      }
    }
    if (__ballot(state != WT_EXIT) == 0ull)
      break;
    const uint32_t n_done = (uint32_t)__popcll(__ballot(state == WT_EXIT));
    // ---- (3) traversal steps while enough lanes are traversing (k_render_v8's loop) ----
    while (true) {
      const uint32_t top2 = ref >> 30;
      const uint32_t kind = (state == WT_TRAVERSE ? 3u : 0u) & (0xB5u >> (top2 * 2u));
      const uint32_t n1 = (uint32_t)__popcll(__ballot(kind == 1u));
      const uint32_t n2 = (uint32_t)__popcll(__ballot(kind == 2u));
      const uint32_t n3 = (uint32_t)__popcll(__ballot(kind == 3u));
      if (n1 + n2 + n3 == 0u || 64u - (n1 + n2 + n3) - n_done >= regen_idle || (q_empty && steps >= WF_SUSP_STEPS && n1 + n2 + n3 <= susp_min))
        break;
      ++steps;
      const uint32_t s1 = n1 * w_pair, s2 = n2 * w_inst, s3 = n3 * w_leaf;
      bool do_pop = false;
      uint32_t sel;
      if (s1 >= s2 && s1 >= s3)
        sel = n3 >= join_min ? 10u : 2u;
      else if (s3 >= s2)
        sel = n1 >= join_min ? 10u : 8u;
      else
        sel = 4u;
      const uint32_t act = ((sel >> kind) & 1u) ? kind : 0u;
      if (sel != 4u) {
        const bool do1 = act == 1u, do3 = act == 3u;
        f4v qx, qy, qz;
        f2v qr;
        if (COOP) {
          const uint32_t rec_off = kind == 3u ? geom.tris_off + (ref & 0x7FFFFFFFu) * 48u : geom.pairs_off + ref * 64u;
          uint32_t t0, t1, t2, t3, keep_m0;
          unsigned long long saved, k0, k1, k2, k3;
          asm volatile("s_mov_b64 %[sv], exec\n\t"
                       "s_mov_b32 %[km0], m0\n\t"
                       "s_mov_b64 exec, -1\n\t"
                       "s_waitcnt lgkmcnt(0)\n\t" // the previous step's reads of the staging area are done
                       "s_nop 3\n\t"
                       "v_mov_b32_dpp %[t0], %[act] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cmp_ne_u32_e32 vcc, 0, %[t0]\n\t"
                       "v_add_u32_dpp %[t0], %[off], %[sub] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                       "s_mov_b64 %[k0], vcc\n\t"
                       "v_mov_b32_dpp %[t1], %[act] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
                       "v_add_u32_dpp %[t1], %[off], %[sub] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                       "s_mov_b64 %[k1], vcc\n\t"
                       "v_mov_b32_dpp %[t2], %[act] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cmp_ne_u32_e32 vcc, 0, %[t2]\n\t"
                       "v_add_u32_dpp %[t2], %[off], %[sub] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                       "s_mov_b64 %[k2], vcc\n\t"
                       "v_mov_b32_dpp %[t3], %[act] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                       "v_cmp_ne_u32_e32 vcc, 0, %[t3]\n\t"
                       "v_add_u32_dpp %[t3], %[off], %[sub] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                       "s_mov_b64 %[k3], vcc\n\t"
                       "s_mov_b32 m0, %[l0]\n\t"
                       "s_mov_b64 exec, %[k0]\n\t"
                       "s_nop 0\n\t"
                       "global_load_lds_dwordx4 %[t0], %[base]\n\t"
                       "s_add_u32 m0, %[l0], 1040\n\t"
                       "s_mov_b64 exec, %[k1]\n\t"
                       "s_nop 0\n\t"
                       "global_load_lds_dwordx4 %[t1], %[base]\n\t"
                       "s_add_u32 m0, %[l0], 2080\n\t"
                       "s_mov_b64 exec, %[k2]\n\t"
                       "s_nop 0\n\t"
                       "global_load_lds_dwordx4 %[t2], %[base]\n\t"
                       "s_add_u32 m0, %[l0], 3120\n\t"
                       "s_mov_b64 exec, %[k3]\n\t"
                       "s_nop 0\n\t"
                       "global_load_lds_dwordx4 %[t3], %[base]\n\t"
                       "s_mov_b32 m0, %[km0]\n\t"
                       "s_mov_b64 exec, %[sv]\n\t"
                       "s_waitcnt vmcnt(0)"
                       : [sv] "=&s"(saved), [km0] "=&s"(keep_m0), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3),
                         [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                       : [act] "v"(act), [off] "v"(rec_off), [sub] "v"(sub16), [l0] "s"(stage_lds),
                         [base] "s"(geom.base)
                       : "memory", "vcc", "scc");
          // the lane's own record, written by its quad (read by every lane: a conditional read makes the registers
          // merges with an undefined value, which costs a second register set; idle lanes read stale bytes)
          qx = my_rec[0], qy = my_rec[1], qz = my_rec[2];
          const f4v qw = my_rec[3];
          qr = f2v{qw.x, qw.y};
        } else {
          // the shared, EXEC-narrowing load group of k_render_v8 (see there for why it is inline assembly)
          const uint32_t stride = kind == 3u ? 48u : 64u;
          const uint64_t base = kind == 3u ? (uint64_t)(uintptr_t)sc.tris : (uint64_t)(uintptr_t)sc.pairs;
          const char *q = reinterpret_cast<const char *>(sc.pairs) +
                          ((uint64_t)(ref & 0x7FFFFFFFu) * stride + (base - (uint64_t)(uintptr_t)sc.pairs));
          const unsigned long long m1 = __ballot(do1), m13 = __ballot(act != 0u);
          unsigned long long saved;
          if (NCACHE) {
            // lanes whose pair record sits in the LDS copy read it from there: the same destination registers, disjoint
            // lanes (a load writes only the lanes of its EXEC), both kinds in flight until the consumers' waits
            const unsigned long long mc = __ballot(do1 && ref < n_cached);
            const unsigned long long g13 = m13 & ~mc, g1 = m1 & ~mc;
            const uint32_t la = cache_lds + (ref << 6);
            asm volatile("; RT_V8_LOADS_BEGIN (tools/check_v8_asm.py)\n\t"
                         "s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b64 exec, %[g13]\n\t"
                         "global_load_dwordx4 %[a], %[q], off\n\t"
                         "global_load_dwordx4 %[b], %[q], off offset:16\n\t"
                         "global_load_dwordx4 %[c], %[q], off offset:32\n\t"
                         "s_mov_b64 exec, %[g1]\n\t"
                         "global_load_dwordx2 %[d], %[q], off offset:48\n\t"
                         "s_mov_b64 exec, %[mc]\n\t"
                         "ds_read_b128 %[a], %[la]\n\t"
                         "ds_read_b128 %[b], %[la] offset:16\n\t"
                         "ds_read_b128 %[c], %[la] offset:32\n\t"
                         "ds_read_b64 %[d], %[la] offset:48\n\t"
                         "s_mov_b64 exec, %[sv]\n\t"
                         "; RT_V8_LOADS_END"
                         : [a] "=&v"(qx), [b] "=&v"(qy), [c] "=&v"(qz), [d] "=&v"(qr), [sv] "=&s"(saved)
                         : [q] "v"(q), [la] "v"(la), [g13] "s"(g13), [g1] "s"(g1), [mc] "s"(mc)
                         : "memory");
          } else {
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
        }
        if (do3) { // one triangle of the leaf: TriangleGeom::hit, Triangle.slang:6-42 (branch-free form, see k_render_v8)
          if (NCACHE) // (also the LDS reads of the pair lanes: they write other lanes of the same registers)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0) ; RT_V8_WAIT_TRI" : "+v"(qx), "+v"(qy), "+v"(qz));
          else
            asm volatile("s_waitcnt vmcnt(0) ; RT_V8_WAIT_TRI" : "+v"(qx), "+v"(qy), "+v"(qz));
          if (__float_as_uint(qy.w))
            do_pop = true;
          else
            ref = ref + 1u;
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
          hit_u = accept ? u : hit_u, hit_v = accept ? v : hit_v;
          hit_tri = accept ? __float_as_uint(qx.w) : hit_tri, hit_inst = accept ? cur_inst : hit_inst;
        }
        if (do1) {
          if (NCACHE)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0) ; RT_V8_WAIT_PAIR" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qr));
          else
            asm volatile("s_waitcnt vmcnt(0) ; RT_V8_WAIT_PAIR" : "+v"(qx), "+v"(qy), "+v"(qz), "+v"(qr));
          const float4 px = make_float4(qx.x, qx.y, qx.z, qx.w), py = make_float4(qy.x, qy.y, qy.z, qy.w),
                       pz = make_float4(qz.x, qz.y, qz.z, qz.w), pr = make_float4(qr.x, qr.y, 0.f, 0.f);
          float dist1, dist2;
          if (fastv) {
            slab_pair_fast<MODE>(co, cd, cinv, px, py, pz, closest, dist1, dist2);
          } else {
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
          const bool hit1 = dist1 != kMiss, hit2 = dist2 != kMiss;
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
          float4 a = q[0], bb = q[1], c = q[2], e = q[3];
          V3 c0 = v3(a.x, a.y, a.z), c1 = v3(bb.x, bb.y, bb.z), c2 = v3(c.x, c.y, c.z), c3 = v3(e.x, e.y, e.z);
          co = mul_cols(c0, c1, c2, c3, wo, 1.f), cd = mul_cols(c0, c1, c2, c3, wd, 0.f);
          bool f2;
          prep_space<MODE>(co, cd, scene_ok, cinv, f2);
          fastv = f2 ? 1u : 0u;
          st.push(RT_REF_SENTINEL);
          ref = __float_as_uint(a.w);
        }
      }
      if (do_pop) {
        if (st.sp == 0) {
          state = WT_FINISHED;
        } else {
          ref = st.pop();
          if (ref == RT_REF_SENTINEL) { // back to world space
            bool f2;
            co = wo, cd = wd, prep_space<MODE>(co, cd, scene_ok, cinv, f2);
            fastv = f2 ? 1u : 0u;
            if (st.sp == 0)
              state = WT_FINISHED;
            else
              ref = st.pop();
          }
        }
      }
    }
  }
  if (PERSIST)
    wp_shade<(WANG & 3)>(sc, cam, pc, b, ch, wp, wp_nt);
  } while (PERSIST);
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_wf_finish: the last paths of a chunk, each seen through to its end by one lane — trace, shade, trace, ... in one
// kernel, as k_render_v0 does per pixel.  The last rounds of a chunk hold a few thousand paths and last as long as
// their longest ray each (0.3 ms, plus a shade pass and a host round trip); here the remaining bounces of a path cost
// their own rays only.  Walks the live-slot list.  A lane whose path ends starts the next path of its slot's shade
// workgroup, if that still has ids (the same strided share as wf_regenerate, one id per atomic), so the chunk is
// complete when the kernel ends whatever was left.  No slot may be suspended (the host checks).
// ---------------------------------------------------------------------------
template <bool WANG>
__global__ void __launch_bounds__(RT_BLOCK)
    k_wf_finish(rt_dev_scene sc, rt_camera_uniforms cam, rt_render_params pc, rt_wf_buffers b, WfChunk ch, uint32_t set,
                uint32_t shade_blocks, unsigned long long *ray_counter) {
  __shared__ uint32_t lds_stack[RT_LDS_STACK * RT_BLOCK];
  RT_STACK_DECL(RT_BLOCK, RT_LDS_STACK, st, lds_stack);
  // entry -> (stripe, position): the stripes' lengths differ, so a workgroup walks one stripe (blockIdx.y)
  const uint32_t stripe = blockIdx.y, at = blockIdx.x * RT_BLOCK + threadIdx.x;
  uint32_t traced = 0;
  if (at < b.ctrl[wf_alive_set(set) + stripe * WF_PAD]) {
    const uint32_t slot = b.live[(size_t)stripe * ch.list_span + at];
    const float4 rd = b.ray_d[slot], ro = b.ray_o[slot];
    uint32_t depth = __float_as_uint(rd.w), id = __float_as_uint(ro.w);
    const bool has_rad = (depth & WF_HAS_RAD) != 0u, has_att = (depth & WF_HAS_ATT) != 0u;
    depth &= ~(WF_HAS_RAD | WF_HAS_ATT);
    const float4 at4 = has_att ? b.att[slot] : make_float4(1.f, 1.f, 1.f, 0.f);
    const float4 ra = has_rad ? b.rad[slot] : make_float4(0.f, 0.f, 0.f, 0.f);
    Path p;
    p.o = v3(ro.x, ro.y, ro.z), p.d = v3(rd.x, rd.y, rd.z);
    p.attenuation = v3(at4.x, at4.y, at4.z), p.sample_radiance = v3(ra.x, ra.y, ra.z);
    const uint32_t spp = pc.sqrt_spp * pc.sqrt_spp;
    const float recip_sqrt_spp = 1.f / (float)pc.sqrt_spp; // Renderer.cpp:628
    const uint32_t blk = slot / WF_SHADE_BLOCK; // the shade workgroup that owns the slot, and its share of the ids
    // Wang: the pixel's RNG state and the frame's radiance so far travel with the slot
    uint32_t seed = 0, s_idx = 0;
    V3 sum = v3(0.f, 0.f, 0.f);
    if (WANG) {
      seed = b.seed[slot];
      if (spp > 1u) {
        const float4 fs = b.fsum[slot];
        sum = v3(fs.x, fs.y, fs.z), s_idx = __float_as_uint(fs.w);
      }
    }
    while (true) {
      uint32_t x, y, entry;
      wf_pixel_of(pc, ch, id, x, y, entry);
      Rng rng = make_rng(pc, x, y);
      if (WANG)
        rng.seed = seed;
      else
        rng.sample = pc.frame_index * spp + entry;
      while (true) { // (keyed PCG: one sample; Wang: what is left of the frame's samples, one after the other)
        for (; depth < pc.max_depth; ++depth) { // (a path in a slot has depth < max_depth)
          rng.set_slot(depth + 1u);
          Hit hit;
          const bool any = trace_mode(sc, pc.aabb_mode, p.o, p.d, 0.0001f, 1000.f, hit, st);
          ++traced;
          if (!shade(sc, pc.flags, any, hit, rng, p))
            break;
        }
        if (!WANG)
          break;
        sum = sum + p.sample_radiance; // RayTracing.slang:170
        if (++s_idx >= spp)
          break;
        camera_ray(cam, recip_sqrt_spp, x, y, s_idx % pc.sqrt_spp, s_idx / pc.sqrt_spp, rng, p);
        depth = 0;
      }
      if (WANG)
        b.sr[id] = make_float4(sum.x, sum.y, sum.z, 0.f);
      else
        b.sr[id] = make_float4(p.sample_radiance.x, p.sample_radiance.y, p.sample_radiance.z, 0.f);
      // the next path of the workgroup's share, if any (ids on tile padding are skipped)
      bool started = false;
      while (true) {
        const uint32_t q = atomicAdd(b.block_pos + blk, 1u);
        const uint64_t id64 = wf_share_id(blk, shade_blocks, q);
        if (id64 >= ch.total)
          break;
        id = (uint32_t)id64;
        wf_pixel_of(pc, ch, id, x, y, entry);
        if (x < pc.image_width && y < pc.row_end) {
          rng = make_rng(pc, x, y);
          if (WANG) {
            rng.seed = wf_wang_seed(pc, x, y, entry);
            camera_ray(cam, recip_sqrt_spp, x, y, 0u, 0u, rng, p);
            seed = rng.seed, s_idx = 0, sum = v3(0.f, 0.f, 0.f);
          } else {
            const uint32_t sample = entry % spp;
            rng.sample = pc.frame_index * spp + entry;
            rng.set_slot(0);
            camera_ray(cam, recip_sqrt_spp, x, y, sample % pc.sqrt_spp, sample / pc.sqrt_spp, rng, p);
          }
          depth = 0;
          started = true;
          break;
        }
      }
      if (!started)
        break;
    }
    b.ray_d[slot] = make_float4(0.f, 0.f, 0.f, __uint_as_float(WF_DEAD));
  }
  count_rays(ray_counter, traced);
}

// ---------------------------------------------------------------------------
// k_wf_resolve: one lane per pixel; radiance = sum of the chunk's sample radiances in sample order on top of the
// chunks before (RayTracing.slang:170), the last chunk folds it into the running mean (:174-179).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(RT_BLOCK)
    k_wf_resolve(rt_render_params pc, rt_wf_buffers b, WfChunk ch, uint32_t tile_slots, float4 *__restrict__ accum,
                 uint32_t first, uint32_t last) {
  const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; // tile * 64 + pixel in tile
  if (i >= tile_slots)
    return;
  const uint32_t tile = i >> 6, in = i & 63u;
  const uint32_t x = (tile % ch.tiles_x) * 8u + (in & 7u);
  const uint32_t y = rt_row(pc, (tile / ch.tiles_x) * 8u + (in >> 3));
  if (!(x < pc.image_width && y < pc.row_end))
    return;
  V3 radiance = v3(0.f, 0.f, 0.f); // of the frame that is open at the chunk's first entry
  if (!first) {
    const float4 part = b.partial[i];
    radiance = v3(part.x, part.y, part.z);
  }
  const float4 *sr = b.sr + ((size_t)tile * ch.cs) * 64u + in;
  float4 *px = accum + ((size_t)y * pc.image_width + x);
  const float pixel_sample_scale = 1.f / (float)(pc.sqrt_spp * pc.sqrt_spp);
  V3 acc = v3(0.f, 0.f, 0.f);
  bool have_acc = false; // the pixel's running mean is read when its first frame of this chunk closes
  uint32_t in_frame = ch.s0 % ch.per_frame, frame = ch.s0 / ch.per_frame;
  for (uint32_t s = 0; s < ch.cs; ++s) {
    const float4 v = sr[(size_t)s * 64u];
    radiance = radiance + v3(v.x, v.y, v.z);
    if (++in_frame == ch.per_frame) { // the frame is complete: RayTracing.slang:174-179 with frame_index + frame
      if (!have_acc) {
        const float4 prev = *px;
        acc = v3(prev.x, prev.y, prev.z), have_acc = true;
      }
      const uint32_t count = pc.accum_count + frame;
      const float fi = (float)count, fi1 = (float)(count + 1u);
      acc = (acc * fi + radiance * pixel_sample_scale) / fi1;
      radiance = v3(0.f, 0.f, 0.f);
      in_frame = 0, ++frame;
    }
  }
  if (have_acc)
    *px = make_float4(acc.x, acc.y, acc.z, 1.f);
  if (!last)
    b.partial[i] = make_float4(radiance.x, radiance.y, radiance.z, 0.f);
}

} // namespace

// ---------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------
#ifndef RT_WF_MINW
#define RT_WF_MINW 6 // resident waves per SIMD the trace kernel is built for
#endif
#define RT_WF_NLDS 16 // stack entries per lane in LDS: 16 KiB per 256 lanes

uint32_t rt_wf_pool_records(uint32_t cus) { // one record per lane of every trace wave that can be resident
  return cus * RT_WF_MINW * 4u * 64u;
}

uint64_t rt_wf_persistent_slots(uint32_t cus) { // a window for every trace wave that can be resident
  return (uint64_t)cus * RT_WF_MINW * 4u * WP_WINDOW;
}

size_t rt_wf_state_bytes(uint32_t slots, uint64_t tile_slots, uint32_t pool_records) {
  return (size_t)slots * (6 * sizeof(float4) + sizeof(float) + sizeof(uint32_t)) + RT_WF_CTRL_WORDS * sizeof(uint32_t) + 2 * 256 +
         ((size_t)slots / WF_SHADE_BLOCK + 1) * sizeof(uint32_t) + ((size_t)slots + (size_t)(WF_STRIPES + 1) * WF_SHADE_BLOCK) * sizeof(uint32_t) +
         (size_t)2 * pool_records * WF_REC_WORDS * sizeof(uint32_t) +
         (size_t)tile_slots * sizeof(float4) + 16 * 256;
}

void rt_wf_carve(void *state_mem, uint32_t slots, uint64_t tile_slots, uint32_t pool_records, rt_wf_buffers &b) {
  static_assert(WF_CTRL_WORDS <= RT_WF_CTRL_WORDS, "control block");
  char *p = static_cast<char *>(state_mem);
  auto take = [&](size_t bytes) {
    char *r = p;
    p += (bytes + 255) & ~(size_t)255;
    return r;
  };
  b.ctrl = reinterpret_cast<uint32_t *>(take(RT_WF_CTRL_WORDS * sizeof(uint32_t)));
  b.ray_o = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.ray_d = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.att = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.rad = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.hit = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.hit_v = reinterpret_cast<float *>(take((size_t)slots * sizeof(float)));
  b.seed = reinterpret_cast<uint32_t *>(take((size_t)slots * sizeof(uint32_t)));
  b.fsum = reinterpret_cast<float4 *>(take((size_t)slots * sizeof(float4)));
  b.partial = reinterpret_cast<float4 *>(take((size_t)tile_slots * sizeof(float4)));
  b.block_pos = reinterpret_cast<uint32_t *>(take(((size_t)slots / WF_SHADE_BLOCK + 1) * sizeof(uint32_t)));
  b.live = reinterpret_cast<uint32_t *>(take(((size_t)slots + (size_t)(WF_STRIPES + 1) * WF_SHADE_BLOCK) * sizeof(uint32_t)));
  b.pool = reinterpret_cast<uint32_t *>(take((size_t)2 * pool_records * WF_REC_WORDS * sizeof(uint32_t)));
  b.pool_records = pool_records;
  b.slots = slots;
}

// the cooperative-fetch build: 12 KiB of stacks + 16.25 KiB of staging per workgroup, 5 workgroups per CU
#ifndef RT_WF_COOP_MINW
#define RT_WF_COOP_MINW 5
#endif
#ifndef RT_WF_COOP_NLDS
#define RT_WF_COOP_NLDS 12
#endif

uint64_t rt_wf_entries(const rt_render_params &pc, uint32_t frame_count) {
  return pc.rng_mode == RT_RNG_WANG ? (uint64_t)frame_count : (uint64_t)frame_count * pc.sqrt_spp * pc.sqrt_spp;
}

// Workgroup shape of the trace kernel (per-lane fetch): lanes, stack entries per lane in LDS, records of the LDS copy
// of the scene's breadth-first top.  Measured on C3 (256 spp, ms per launch; experiments/r03_lds_cache/ has all nine
// shapes and the counters): 256 / 16 / none 242.0 (rounds 2 and 3), 256 / 16 / 160 237.5, 512 / 16 / 320 236.8,
// 768 / 16 / 448 236.7, 768 / 12 / 640 236.5; larger workgroups or 12 stack entries without the copy: 242.0.  The copy
// takes 28 % of the L1 accesses and 11 % of the L2 requests away and buys 2 %: the step is bound by its slowest lane's
// miss, not by the access rate.  Three workgroups of 512 lanes per CU: 3 x (32 + 20) KiB of the 160 KiB of LDS.
struct WfTraceShape {
  uint32_t block, nlds, ncache;
};
#define WF_TRACE_BLOCK 512
#define WF_TRACE_CACHE 320
template <int MODE, int BLOCK, int NLDS, int NCACHE, typename... Args>
static hipError_t wf_launch_trace_shape(uint32_t blocks, hipStream_t stream, Args... args) {
  const size_t lds_bytes = (size_t)NLDS * BLOCK * 4 + (size_t)NCACHE * 64;
  static_assert((size_t)NLDS * BLOCK * 4 + (size_t)NCACHE * 64 <= 65536, "more needs hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  hipLaunchKernelGGL((k_wf_trace<MODE, RT_WF_MINW, BLOCK, NLDS, false, NCACHE, false, false>), dim3(blocks), dim3(BLOCK), lds_bytes, stream, args...);
  return hipSuccess;
}
// The persistent mode's shapes.  The kernel's WANG parameter: 0 = keyed PCG, 1 = the reference's RNG at ONE sample per
// frame (the reference's own sampling, Renderer.cpp:21), 2 = at more (the slot carries the frame's sample sum and the
// shade phase may start the pixel's next camera ray); WP_KIND_BANDS added to kind 0 or 1 = the groups of path ids dealt
// to the XCD classes in bands (wp_draw_group).  Kinds 0 and 1 take the trace pass's shape (6 waves per SIMD,
// 512 lanes, 320 records in LDS); kind 2, whose shade part needs 86 registers, 5 waves per SIMD in workgroups of 256
// lanes (5 x (16 + 10) KiB of LDS per CU).  C3, 256 frames of 1 spp: one kernel for both Wang kinds at 6 waves (28 spills in
// the shade phase) 298 ms, at 5 waves 236 ms, at 4 waves 266 ms, the rounds 240 ms; kind 1 on its own at 6 waves
// (the next-sample code compiled out: 14 spills) 222.5 ms = 5430 Mrays/s.  The keyed PCG RNG in the 5-wave shape: 236
// against 228 ms.
#ifndef WP_WANG_MINW
#define WP_WANG_MINW 5
#endif
#define WP_WANG_SHAPE5 2 // the kernel's WANG kind that takes the 5-wave shape: the reference's RNG at spp > 1 per frame
#define WP_WANG_BLOCK 256
#define WP_WANG_CACHE 160
template <int MODE, int WANG, typename... Args>
static hipError_t wf_launch_persist(uint32_t blocks, hipStream_t stream, Args... args) {
  constexpr int BLOCK = WANG == WP_WANG_SHAPE5 ? WP_WANG_BLOCK : WF_TRACE_BLOCK, CACHE = WANG == WP_WANG_SHAPE5 ? WP_WANG_CACHE : WF_TRACE_CACHE;
  const size_t lds_bytes = (size_t)RT_WF_NLDS * BLOCK * 4 + (size_t)CACHE * 64;
  hipLaunchKernelGGL((k_wf_trace<MODE, WANG == WP_WANG_SHAPE5 ? WP_WANG_MINW : RT_WF_MINW, BLOCK, RT_WF_NLDS, false, CACHE, true, WANG>), dim3(blocks),
                     dim3(BLOCK), lds_bytes, stream, args...);
  return hipSuccess;
}

template <int MODE, typename... Args>
static hipError_t wf_launch_trace(const WfTraceShape &s, uint32_t blocks, hipStream_t stream, Args... args) {
  if (s.ncache)
    return wf_launch_trace_shape<MODE, WF_TRACE_BLOCK, RT_WF_NLDS, WF_TRACE_CACHE>(blocks, stream, args...);
  return wf_launch_trace_shape<MODE, RT_BLOCK, RT_WF_NLDS, 0>(blocks, stream, args...);
}

hipError_t rt_launch_render_wavefront(const rt_dev_scene &sc, const rt_wf_geom &geom, const rt_camera_uniforms &cam,
                                      const rt_render_params &pc, uint32_t frame_count, float *d_accum, unsigned long long *d_counter,
                                      const rt_wf_buffers &b, uint32_t *h_ctrl, uint32_t knobs, const rt_wf_tuning &tune,
                                      rt_wf_counts *counts, uint32_t cus, hipStream_t stream) {
  const uint32_t rows = rt_virtual_rows(pc);
  if (rows == 0 || pc.image_width == 0 || frame_count == 0)
    return hipSuccess;
  const bool wang = pc.rng_mode == RT_RNG_WANG;
  const uint64_t entries = rt_wf_entries(pc, frame_count);
  if (cus == 0 || b.slots < 64 || pc.rng_mode > RT_RNG_PCG || pc.max_depth >= WF_SUSPENDED || entries > 0xFFFFFFFFull)
    return hipErrorInvalidValue;
  const uint32_t tiles_x = (pc.image_width + 7u) >> 3;
  const uint64_t tile_slots = (uint64_t)tiles_x * ((rows + 7u) >> 3) * 64u;
  const uint32_t spp = (uint32_t)entries; // entries of the launch: the loop below hands them out in chunks
  uint64_t chunk = b.sr_paths / tile_slots; // entries whose results fit the result buffer
  if (chunk > spp)
    chunk = spp;
  if (chunk != 0 && tile_slots * chunk > 0xF0000000ull) // path ids are 32-bit
    chunk = 0xF0000000ull / tile_slots;
  if (chunk == 0)
    return hipErrorInvalidValue;
  float4 *accum = reinterpret_cast<float4 *>(d_accum);
  // knobs: bits 8-15 regeneration threshold of the trace kernel, 16-27 vote weights, 28-30 join code, 31 no fast slab
  // default 24 idle lanes (C3, 256 spp: 4 -> 308 ms, 8 -> 301, 16 -> 289, 24 -> 282, 32 -> 292, 48 -> 327 per launch)
  const uint32_t regen_idle = ((knobs >> 8) & 0x7Fu) ? ((knobs >> 8) & 0x7Fu) : 24u;
  // scheduling of the rounds (rt_wf_tuning; 0xFFFFFFFF = the measured defaults, rt_wavefront.h)
  const uint32_t susp_lanes = tune.suspend_lanes == 0xFFFFFFFFu ? 32u : tune.suspend_lanes > 64u ? 64u : tune.suspend_lanes;
  const uint32_t list_below = tune.list_percent == 0xFFFFFFFFu ? 75u : tune.list_percent;
  // (half a path per lane the trace kernel can have resident: C3 shard of an eighth, ms per launch, finisher from
  // 0 / 50 K / 200 K / 500 K / 1 M / 2 M / 4 M paths on: 39.9 / 39.7 / 39.0 / 39.4 / 40.1 / 41.6 / 44.8; full frame 255.3 ... 254.3)
  const uint64_t finish_paths = tune.finish_paths == 0xFFFFFFFFu ? (uint64_t)cus * RT_WF_MINW * 4u * 32u : tune.finish_paths;
  const uint32_t full_share = tune.share == 0xFFFFFFFFu ? WF_GRAB : tune.share < 64u ? 64u : tune.share > 4096u ? 4096u : tune.share / 64u * 64u;
  const bool coop = (knobs & 0x8000u) != 0u; // knob bit 15: quad-cooperative fetch instead of per-lane gathers
  const uint32_t sched = ((knobs >> 16) & 0xFFFu) | ((knobs >> 31) ? 0x8000u : 0u) | (((knobs >> 28) & 7u) << 12);
  // workgroup shape of the trace kernel: lanes, stack entries in LDS, records of the LDS copy of the scene's top
  const WfTraceShape shape = coop ? WfTraceShape{RT_BLOCK, RT_WF_COOP_NLDS, 0}
                           : tune.record_cache != 0u ? WfTraceShape{WF_TRACE_BLOCK, RT_WF_NLDS, WF_TRACE_CACHE} : WfTraceShape{RT_BLOCK, RT_WF_NLDS, 0};
  const uint32_t tblock = shape.block;
  const uint64_t resident = (uint64_t)cus * ((coop ? RT_WF_COOP_MINW : RT_WF_MINW) * 4u / (tblock / 64u));
#define RT_WF_TRACE(MODE)                                                                                              \
  do {                                                                                                                 \
    if (coop) {                                                                                                        \
      const size_t lds_bytes = (size_t)RT_WF_COOP_NLDS * RT_BLOCK * 4 + (RT_BLOCK / 64u) * WF_COOP_BYTES;              \
      hipLaunchKernelGGL((k_wf_trace<MODE, RT_WF_COOP_MINW, RT_BLOCK, RT_WF_COOP_NLDS, true, 0, false, false>), dim3(blocks), dim3(RT_BLOCK), \
                         lds_bytes, stream, sc, geom, b, ch, d_counter, regen_idle, sched, susp, cam, pc);             \
    } else if ((e = wf_launch_trace<MODE>(shape, blocks, stream, sc, geom, b, ch, d_counter, regen_idle, sched, susp, cam, pc)) != hipSuccess) \
      return e;                                                                                                        \
  } while (0)
  auto span64 = [](uint64_t total) { // per-band share, whole windows of the shade pass (so that a share is one window)
    const uint64_t per = (total + WF_STRIPES - 1) / WF_STRIPES;
    return (uint32_t)((per + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK * WF_SHADE_BLOCK);
  };
  // in-window compaction of the shade pass (k_wf_shade); not for the reference's RNG at spp > 1, whose per-slot sample
  // sum would have to move with the path
  const uint32_t compact_bit = tune.compact != 0u && !(wang && pc.sqrt_spp > 1u) ? 0x40000000u : 0u;
  hipError_t e;
  // measurement mode: an event before and after every dispatch, read back when the launch has drained
  struct Stamp {
    hipEvent_t a, b;
    double *into;
  };
  std::vector<Stamp> stamps;
  const bool timing = counts && counts->timing;
  auto stamp_begin = [&](double *into) {
    if (!timing)
      return;
    Stamp st{nullptr, nullptr, into};
    if (hipEventCreate(&st.a) == hipSuccess && hipEventCreate(&st.b) == hipSuccess && hipEventRecord(st.a, stream) == hipSuccess)
      stamps.push_back(st);
  };
  auto stamp_end = [&]() {
    if (timing && !stamps.empty())
      (void)hipEventRecord(stamps.back().b, stream);
  };
  struct StampsDone { // runs on every return path
    std::vector<Stamp> &v;
    hipStream_t stream;
    rt_wf_counts *counts;
    ~StampsDone() {
      if (v.empty())
        return;
      (void)hipStreamSynchronize(stream);
      for (Stamp &st : v) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, st.a, st.b) == hipSuccess)
          *st.into += ms;
        (void)hipEventDestroy(st.a);
        (void)hipEventDestroy(st.b);
      }
      ++counts->timed_launches;
    }
  } stamps_done{stamps, stream, counts};
  for (uint32_t s0 = 0; s0 < spp; s0 += (uint32_t)chunk) {
    WfChunk ch;
    ch.s0 = s0;
    ch.cs = (uint32_t)(spp - s0 < chunk ? spp - s0 : chunk);
    ch.tiles_x = tiles_x;
    ch.total = (uint32_t)(tile_slots * ch.cs);
    ch.slots = ch.total < b.slots ? (ch.total + 63u) / 64u * 64u : b.slots / 64u * 64u;
    if (ch.slots > b.slots)
      ch.slots = b.slots / 64u * 64u;
    ch.slot_span = span64(ch.slots);
    ch.list_span = ((ch.slots + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK + WF_STRIPES - 1) / WF_STRIPES * WF_SHADE_BLOCK;
    ch.per_frame = wang ? 1u : pc.sqrt_spp * pc.sqrt_spp;
    const bool first = s0 == 0, last = s0 + ch.cs == spp;
    if (pc.max_depth == 0) { // no bounce is traced: every sample contributes nothing (RayTracing.slang:92)
      if ((e = hipMemsetAsync(b.sr, 0, (size_t)ch.total * sizeof(float4), stream)) != hipSuccess)
        return e;
    } else {
      if ((e = hipMemsetAsync(b.ctrl, 0, RT_WF_CTRL_WORDS * sizeof(uint32_t), stream)) != hipSuccess)
        return e;
      const uint32_t shade_blocks = (ch.slots + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK;
      if ((e = hipMemsetAsync(b.block_pos, 0, (size_t)shade_blocks * sizeof(uint32_t), stream)) != hipSuccess)
        return e;
      // (a budget below one window: the rounds)
      if (tune.persistent != 0u && !coop && b.slots >= WP_WINDOW) {
        // No rounds: one launch of the trace kernel in its PERSIST mode renders the chunk — every wave fills, traces
        // and shades its own window of slots until the chunk's path ids are used up.  Nothing is read back, the host
        // does not wait.
        const int wkind = !wang ? 0 : pc.sqrt_spp > 1u ? 2 : 1; // the kernel's WANG parameter
        const bool shape5 = wkind == WP_WANG_SHAPE5;
        const uint32_t pblock = shape5 ? WP_WANG_BLOCK : WF_TRACE_BLOCK;
        const uint64_t waves_max = (uint64_t)cus * (shape5 ? WP_WANG_MINW : RT_WF_MINW) * 4u;
        uint64_t waves = ((uint64_t)ch.total + WP_WINDOW - 1) / WP_WINDOW;
        waves = waves < waves_max ? waves : waves_max;
        if (waves * WP_WINDOW > b.slots)
          waves = b.slots / WP_WINDOW;
        const uint32_t pblocks = (uint32_t)((waves + pblock / 64 - 1) / (pblock / 64));
        if (waves == 0 || waves > 0xFFFFu || (uint64_t)pblocks * pblock > b.pool_records)
          return hipErrorInvalidValue;
        const uint32_t susp = susp_lanes | ((uint32_t)waves << 16); // (one pool, no list; bits 16-31: windows)
        const bool big_scene = geom.bytes > WP_NT_SCENE_BYTES;
        const uint32_t psched = sched | (big_scene ? 0x10000u : 0u);
        const bool bands = wkind != 2 && (tune.bands == 0xFFFFFFFFu ? big_scene && ch.total >= WP_BANDS_MIN_PATHS
                                                                  : tune.bands != 0u); // (wp_draw_group)
        stamp_begin(counts ? &counts->trace_ms : nullptr);
#define RT_WF_PERSIST_KIND(MODE, KIND)                                                                                 \
  wf_launch_persist<MODE, KIND>(pblocks, stream, sc, geom, b, ch, d_counter, regen_idle, psched, susp, cam, pc)
#define RT_WF_PERSIST(MODE)                                                                                            \
  (wkind == 2   ? RT_WF_PERSIST_KIND(MODE, 2)                                                                          \
   : wkind == 1 ? (bands ? RT_WF_PERSIST_KIND(MODE, 1 | WP_KIND_BANDS) : RT_WF_PERSIST_KIND(MODE, 1))                  \
                : (bands ? RT_WF_PERSIST_KIND(MODE, 0 | WP_KIND_BANDS) : RT_WF_PERSIST_KIND(MODE, 0)))
        e = pc.aabb_mode == RT_AABB_RCP ? RT_WF_PERSIST(RT_AABB_RCP) : RT_WF_PERSIST(RT_AABB_DIV);
#undef RT_WF_PERSIST_KIND
#undef RT_WF_PERSIST
        stamp_end();
        if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess)
          return e;
        if (counts)
          ++counts->rounds;
      } else {
        // first fill: every slot starts a path (or is marked dead when the chunk has fewer paths than slots)
        stamp_begin(counts ? &counts->fill_ms : nullptr);
        if (wang)
          hipLaunchKernelGGL((k_wf_shade<true, true>), dim3(shade_blocks), dim3(WF_SHADE_BLOCK), 0, stream, sc, cam, pc, b, ch, compact_bit);
        else
          hipLaunchKernelGGL((k_wf_shade<true, false>), dim3(shade_blocks), dim3(WF_SHADE_BLOCK), 0, stream, sc, cam, pc, b, ch, compact_bit);
        stamp_end();
        bool have_list = false; // the last shade pass wrote the live-slot list
        bool maybe_suspended = false; // the last trace pass was allowed to suspend rays
        for (uint32_t round = 0;; ++round) { // trace pass `round` follows shade pass `round` (the first fill is pass 0)
          // live paths after the last shade: the host decides whether another round is needed
          if ((e = hipMemcpyAsync(h_ctrl, b.ctrl + wf_alive_set(round), WF_STRIPES * WF_PAD * sizeof(uint32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess)
            return e;
          if ((e = hipStreamSynchronize(stream)) != hipSuccess)
            return e;
          uint64_t alive = 0;
          for (uint32_t k = 0; k < WF_STRIPES; ++k)
            alive += h_ctrl[k * WF_PAD];
          if (alive == 0)
            break;
          // the next shade pass writes the live-slot list once this chunk has run out of new paths (until then every
          // slot is refilled and the list would be all of them)
          const bool make_list = list_below && alive < ch.slots;
          if (have_list && !maybe_suspended && alive <= finish_paths) { // the last few: one kernel sees them through
            uint32_t longest = 0;
            for (uint32_t k = 0; k < WF_STRIPES; ++k)
              longest = h_ctrl[k * WF_PAD] > longest ? h_ctrl[k * WF_PAD] : longest;
            stamp_begin(counts ? &counts->finish_ms : nullptr);
            if (wang)
              hipLaunchKernelGGL(k_wf_finish<true>, dim3((longest + RT_BLOCK - 1) / RT_BLOCK, WF_STRIPES), dim3(RT_BLOCK), 0, stream,
                                 sc, cam, pc, b, ch, round, shade_blocks, d_counter);
            else
              hipLaunchKernelGGL(k_wf_finish<false>, dim3((longest + RT_BLOCK - 1) / RT_BLOCK, WF_STRIPES), dim3(RT_BLOCK), 0, stream,
                                 sc, cam, pc, b, ch, round, shade_blocks, d_counter);
            stamp_end();
            if ((e = hipGetLastError()) != hipSuccess)
              return e;
            if (counts)
              counts->finished_paths += alive;
            break;
          }
          // reset the trace kernel's queue heads and the counters of the NEXT shade pass (adjacent in the control block;
          // this pass's counters stay: they are the lengths of the live-slot list)
          if ((e = hipMemsetAsync(b.ctrl + ((round + 1u) & 1u ? WF_HEAD : WF_ALIVE0), 0, 2 * WF_STRIPES * WF_PAD * sizeof(uint32_t),
                                  stream)) != hipSuccess)
            return e;
          // While most slots hold a path the trace pass walks the slots themselves (coalesced, no indirection; a dead
          // slot costs a 16-byte read); once they thin out it walks the list of live slots the shade pass left, and
          // the grid is sized by that list.
          const bool by_list = have_list && alive * 100u < (uint64_t)ch.slots * list_below;
          // (a short list goes out in smaller shares, down to one ray per lane: the last rounds are a matter of latency)
          uint32_t share = full_share;
          if (by_list && alive / (resident * (tblock / 64u)) < share)
            share = (uint32_t)(alive / (resident * (tblock / 64u))) / 64u * 64u;
          share = share < 64u ? 64u : share;
          const uint64_t need = ((by_list ? alive : (uint64_t)ch.slots) + share - 1) / share; // one wave per share at least
          const uint32_t blocks = (uint32_t)(need / (tblock / 64) + 1 < resident ? need / (tblock / 64) + 1 : resident);
          // rays may be suspended while the rounds are full (the next round comes anyway and carries them along); once
          // the paths thin out every ray is seen through, or the stragglers would add rounds of their own
          const uint32_t susp = (susp_lanes && !coop && alive * 2u >= ch.slots && (uint64_t)blocks * tblock <= b.pool_records
                                     ? susp_lanes : 0u) | ((round & 1u) << 8) | (by_list ? 0x200u : 0u) | ((round & 1u) << 10) | (share << 16);
          stamp_begin(counts ? &counts->trace_ms : nullptr);
          if (pc.aabb_mode == RT_AABB_RCP)
            RT_WF_TRACE(RT_AABB_RCP);
          else
            RT_WF_TRACE(RT_AABB_DIV);
          stamp_end();
          stamp_begin(counts ? &counts->shade_ms : nullptr);
          if (wang)
            hipLaunchKernelGGL((k_wf_shade<false, true>), dim3(shade_blocks), dim3(WF_SHADE_BLOCK), 0, stream, sc, cam, pc, b, ch,
                               ((round + 1u) & 1u) | compact_bit | (make_list ? 0x80000000u : 0u));
          else
            hipLaunchKernelGGL((k_wf_shade<false, false>), dim3(shade_blocks), dim3(WF_SHADE_BLOCK), 0, stream, sc, cam, pc, b, ch,
                               ((round + 1u) & 1u) | compact_bit | (make_list ? 0x80000000u : 0u));
          stamp_end();
          have_list = make_list;
          if ((e = hipGetLastError()) != hipSuccess)
            return e;
          maybe_suspended = (susp & 0xFFu) != 0u;
          if (counts) {
            ++counts->rounds;
            counts->list_rounds += by_list ? 1u : 0u;
          }
        }
      }
    }
    stamp_begin(counts ? &counts->resolve_ms : nullptr);
    hipLaunchKernelGGL(k_wf_resolve, dim3((uint32_t)((tile_slots + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, stream, pc,
                       b, ch, (uint32_t)tile_slots, accum, first ? 1u : 0u, last ? 1u : 0u);
    stamp_end();
    if ((e = hipGetLastError()) != hipSuccess)
      return e;
  }
  return hipSuccess;
}
