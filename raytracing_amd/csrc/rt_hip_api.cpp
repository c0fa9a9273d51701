// rt_hip_api.cpp — the C ABI of include/rt_hip.h: context, HBM residency of the
// flattened scene, launches, statistics.  No CPU fallback: without a HIP device
// rt_create fails.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>

#include <math.h>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_flatten.h"
#include "rt_kernels.h"
#include "rt_wavefront.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,            \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

template <typename T> struct DevBuf {
  T *ptr = nullptr;
  size_t count = 0; // capacity in elements
  hipError_t reserve(size_t n) {
    if (n <= count && ptr)
      return hipSuccess;
    if (ptr)
      (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
    // + 64: kernels may fetch a whole 64-byte line at the last record (k_render_v2 SCHED 3 reads 64 bytes at a 48-byte triangle)
    size_t bytes = (n ? n : 1) * sizeof(T) + 64;
    hipError_t e = hipMalloc((void **)&ptr, bytes);
    if (e == hipSuccess)
      count = n ? n : 1;
    return e;
  }
  void release() {
    if (ptr)
      (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
};

struct EventPair {
  hipEvent_t a, b;
  bool watch = false; // a launch of the automatically chosen kernel: its time is checked for drift (rt_ctx::Tune)
};

} // namespace

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;     // the stream every launch and copy is queued on
  hipStream_t own_stream = nullptr; // created by rt_create; `stream` unless the caller set one (rt_set_stream)
  // scene in HBM
  DevBuf<rt_dev_pair> pairs;
  DevBuf<rt_dev_tri> tris;
  DevBuf<rt_dev_inst> inst_trace;
  DevBuf<rt_dev_inst_shade> inst_shade; // matrices + resolved material per instance
  DevBuf<rt_triangle_shading> shading;
  DevBuf<uint8_t> texels;
  DevBuf<unsigned long long> counter;
  DevBuf<rt_ray> trace_rays; // staging of rt_trace_closest (host arrays in and out)
  DevBuf<rt_hit> trace_hits;
  DevBuf<uint32_t> queue; // pixel queue head of the persistent kernel
  uint32_t cus = 0;       // compute units of the device
  // kernel 9 (rt_wavefront.hip): path state, per-path results and the pinned read-back of the list lengths
  DevBuf<uint8_t> wf_state, wf_sr;
  DevBuf<uint8_t> wf_geom;    // [pair records | triangles] in ONE allocation: 32-bit record offsets from one base
  bool wf_geom_dirty = true;  // scene or instances changed since wf_geom was filled
  rt_wf_geom wf_geom_desc{};
  rt_wf_buffers wf{};
  uint64_t wf_tile_slots = 0;
  uint32_t wf_slots = 1u << 26;          // most paths in flight (rt_set_wavefront_budget): 64 Mi slots = 5.6 GB of state
  uint32_t wf_capacity = 0;              // slots the state allocation holds
  uint64_t wf_sr_bytes = 40ull << 30;    // result buffer: 16 B per path of a sample chunk; allocated as needed (a 1080p
                                         // frame of 1024 spp in one chunk: 984 ms against 1020 ms in four chunks of 10 GiB)
  uint32_t *wf_h_ctrl = nullptr;
  rt_wf_tuning wf_tuning;                // rt_debug_wavefront_tuning
  rt_wf_counts wf_counts;                // since rt_reset_stats
  rt_dev_scene dev{};
  bool have_scene = false;
  // host-side remnants needed by rt_update_instances
  rt_flat_scene flat;
  uint32_t blas_count = 0;
  rt_scene_desc material_counts{}; // counts only (pointers cleared)
  std::vector<rt_lambert> h_lamberts;
  std::vector<rt_texture_desc> h_textures;
  std::vector<rt_metal> h_metals;
  std::vector<rt_dielectric> h_dielectrics;
  std::vector<rt_emissive> h_emissives;
  std::vector<uint8_t> h_texels; // 1x1 textures are decoded into the per-instance records
  uint64_t texel_bytes = 0;
  // image
  DevBuf<float> accum_own;
  float *accum = nullptr;
  uint32_t width = 0, height = 0;
  bool accum_bound = false;
  rt_camera_uniforms camera{};
  bool have_camera = false;
  uint32_t variant = 8; // persistent, regenerating, vote-scheduled traversal with joined pair/leaf steps (k_render_v8)
  bool variant_auto = true; // rt_render chooses between variants 8 and 0 (RT_KERNEL_VARIANT_AUTO)
  uint32_t regen_idle = 0;  // persistent kernels: regenerate once this many lanes of a wave are idle; 0 = by scene size
  uint32_t last_kernel = 0; // kernel of the most recent rt_render
  // Automatic choice between kernels 8 and 9 for keyed-PCG launches (rt_render): both are run once on a launch shape
  // and the faster one keeps it.
  struct Tune {
    uint32_t key[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t phase = 0; // 0 nothing measured, 1 kernel 9 measured, 2 both measured: `choice` holds
    bool warmed = false; // kernel 9 has run once on this key's buffers (its first launch on fresh allocations is up to
                         // 1.3x slower than its second: the timed trial is the second)
    float ms9 = 0.f, ms8 = 0.f;
    uint32_t choice = 9;
    uint32_t updates = 0; // rt_update_instances calls since the trial
    float watched_ms = 0.f; // time of the most recent finished launch of the chosen kernel on this key (0 = none)
  } tune;
  // A measured choice belongs to an uploaded scene (`scene_epoch` counts rt_upload_scene) of a certain size (the bit
  // length of its pair count) and to a launch shape.  rt_update_instances — the reference's per-frame path when objects
  // move (Renderer.cpp:901-916, 994-1015) — does NOT restart the trial (every frame of an animated scene would then be
  // a blocking trial launch); it is repeated after RT_TUNE_MAX_UPDATES updates, since the moved instances may have
  // changed what the rays meet.
  uint32_t scene_epoch = 0;
  uint32_t fail_allocs = 0; // rt_debug_fail_wavefront_allocs: kernel 9's next device allocations fail (test hook)
  uint32_t pair_count = 0;  // interior records of the uploaded scene (TLAS + every BLAS)
  // timing
  std::vector<EventPair> pending, free_events;
  double total_ms = 0.0, last_ms = 0.0;
  uint64_t launches = 0;
};

namespace {

int drain_events(rt_ctx *ctx) {
  for (EventPair &p : ctx->pending) {
    HIP_TRY(hipEventSynchronize(p.b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
    ctx->last_ms = ms;
    ctx->total_ms += ms;
    if (p.watch)
      ctx->tune.watched_ms = ms;
    p.watch = false;
    ctx->free_events.push_back(p);
  }
  ctx->pending.clear();
  return RT_OK;
}

int begin_timed(rt_ctx *ctx, EventPair &ev) {
  if (ctx->pending.size() >= 1024) {
    int rc = drain_events(ctx);
    if (rc != RT_OK)
      return rc;
  }
  if (!ctx->free_events.empty()) {
    ev = ctx->free_events.back();
    ctx->free_events.pop_back();
  } else {
    HIP_TRY(hipEventCreate(&ev.a));
    HIP_TRY(hipEventCreate(&ev.b));
  }
  HIP_TRY(hipEventRecord(ev.a, ctx->stream));
  return RT_OK;
}

int end_timed(rt_ctx *ctx, const EventPair &ev) {
  HIP_TRY(hipEventRecord(ev.b, ctx->stream));
  ctx->pending.push_back(ev);
  ++ctx->launches;
  return RT_OK;
}

template <typename T>
int upload(rt_ctx *ctx, DevBuf<T> &buf, const void *src, size_t n) {
  HIP_TRY(buf.reserve(n));
  if (n)
    HIP_TRY(hipMemcpyAsync(buf.ptr, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return RT_OK;
}

// [blas pairs | tlas pairs] with TLAS-internal references rebased.
int upload_pairs_and_instances(rt_ctx *ctx, const rt_blas_instance *instances, uint32_t n_inst) {
  rt_flat_scene &f = ctx->flat;
  const uint32_t base = (uint32_t)f.blas_pairs.size();
  std::vector<rt_dev_pair> all(f.blas_pairs);
  all.reserve(all.size() + f.tlas_pairs.size());
  for (rt_dev_pair p : f.tlas_pairs) {
    if (!(p.ref0 & RT_REF_LEAF))
      p.ref0 += base;
    if (!(p.ref1 & RT_REF_LEAF))
      p.ref1 += base;
    all.push_back(p);
  }
  if (all.size() >= RT_REF_INST)
    return fail(RT_ERR_UNSUPPORTED, "too many interior nodes for a 30-bit reference");
  if (f.stack_needed > RT_MAX_STACK)
    return fail(RT_ERR_UNSUPPORTED, "scene needs a traversal stack of " + std::to_string(f.stack_needed) +
                                        " entries; the kernels hold " + std::to_string(RT_MAX_STACK));
  uint32_t root = f.root_ref;
  if (root != RT_REF_NONE && !(root & RT_REF_LEAF))
    root += base;
  // Renumber so that the top of the scene comes first: breadth-first from the
  // TLAS root, through instance leaves into each BLAS once, for the first
  // RT_CACHE_PAIRS records (every ray walks through them: one dense 96-KiB region
  // instead of records scattered over the arrays of several BLASes); the rest keep
  // their depth-first order behind them.
  std::vector<rt_dev_inst> insts(f.insts);
  uint32_t cached = 0;
  {
    const uint32_t n = (uint32_t)all.size();
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<uint8_t> seen(n, 0);
    if (root != RT_REF_NONE) {
      std::vector<uint32_t> fifo;
      auto visit = [&](uint32_t ref) {
        if (ref & RT_REF_LEAF) {
          if (!(ref & RT_REF_INST))
            return;
          ref = insts[ref & 0x3FFFFFFFu].root_ref; // into the BLAS
          if (ref == RT_REF_NONE || (ref & RT_REF_LEAF))
            return;
        }
        if (!seen[ref]) {
          seen[ref] = 1;
          fifo.push_back(ref);
        }
      };
      visit(root);
      for (size_t head = 0; head < fifo.size() && order.size() < RT_CACHE_PAIRS; ++head) {
        const uint32_t i = fifo[head];
        order.push_back(i);
        visit(all[i].ref0);
        visit(all[i].ref1);
      }
    }
    cached = (uint32_t)order.size();
    std::vector<uint8_t> placed(n, 0);
    for (uint32_t i : order)
      placed[i] = 1;
    for (uint32_t i = 0; i < n; ++i)
      if (!placed[i])
        order.push_back(i);
    std::vector<uint32_t> new_index(n);
    for (uint32_t k = 0; k < n; ++k)
      new_index[order[k]] = k;
    auto remap = [&](uint32_t ref) {
      return (ref == RT_REF_NONE || (ref & RT_REF_LEAF)) ? ref : new_index[ref];
    };
    std::vector<rt_dev_pair> sorted(n);
    for (uint32_t k = 0; k < n; ++k) {
      rt_dev_pair p = all[order[k]];
      p.ref0 = remap(p.ref0);
      p.ref1 = remap(p.ref1);
      sorted[k] = p;
    }
    all.swap(sorted);
    for (rt_dev_inst &d : insts)
      d.root_ref = remap(d.root_ref);
    root = remap(root);
  }
  bool moderate = true; // RT_SCENE_BOXES_MODERATE
  for (const rt_dev_pair &p : all) {
    const float *axes[3] = {p.x, p.y, p.z};
    for (const float *a : axes)
      for (int k = 0; k < 4; ++k) {
        const float m = fabsf(a[k]);
        if (!(m == 0.0f || (m >= 0x1p-66f && m <= 0x1p59f))) // NaN and inf fail too
          moderate = false;
      }
  }
  // an error return below must not leave copies from these (or the caller's) host arrays queued
  auto drained = [&](int code) {
    (void)hipStreamSynchronize(ctx->stream);
    return code;
  };
  int rc;
  if ((rc = upload(ctx, ctx->pairs, all.data(), all.size())) != RT_OK)
    return drained(rc);
  ctx->dev.pairs = ctx->pairs.ptr; // kept current after every reallocation
  if ((rc = upload(ctx, ctx->inst_trace, insts.data(), insts.size())) != RT_OK)
    return drained(rc);
  ctx->dev.inst_trace = ctx->inst_trace.ptr;
  // per-instance shading records: matrices + the material resolved against the host copies of the
  // material tables (validated by rt_validate_materials for every instance a TLAS leaf names)
  std::vector<rt_dev_inst_shade> shade(n_inst);
  for (uint32_t i = 0; i < n_inst; ++i) {
    rt_dev_inst_shade &d = shade[i];
    memset(&d, 0, sizeof d);
    memcpy(d.transform, instances[i].transform, sizeof d.transform);
    memcpy(d.inv_transform, instances[i].inv_transform, sizeof d.inv_transform);
    d.mtype = RT_MATERIAL_NONE;
    if (i >= f.inst_live.size() || !f.inst_live[i])
      continue;
    const rt_material_handle &m = instances[i].material;
    d.mtype = m.type;
    if (m.type == RT_MATERIAL_LAMBERT) {
      const rt_texture_desc &t = ctx->h_textures[ctx->h_lamberts[m.index].index];
      d.tex_offset = t.offset, d.tex_w = t.width, d.tex_h = t.height;
      if (t.width == 1 && t.height == 1) { // sample_albedo of a 1x1 texture: the texel, decoded as RGBA8 UNORM
        const uint8_t *px = ctx->h_texels.data() + t.offset;
        d.param[0] = (float)px[0] / 255.0f, d.param[1] = (float)px[1] / 255.0f, d.param[2] = (float)px[2] / 255.0f;
      }
    } else if (m.type == RT_MATERIAL_METAL) {
      memcpy(d.param, ctx->h_metals[m.index].albedo_fuzz, sizeof d.param);
    } else if (m.type == RT_MATERIAL_DIELECTRIC) {
      d.param[0] = ctx->h_dielectrics[m.index].refraction_index;
    } else if (m.type == RT_MATERIAL_EMISSIVE) {
      memcpy(d.param, ctx->h_emissives[m.index].intensity, sizeof d.param);
    }
  }
  if ((rc = upload(ctx, ctx->inst_shade, shade.data(), shade.size())) != RT_OK)
    return drained(rc);
  // the staging vector dies at return: finish the copies first
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->dev.pairs = ctx->pairs.ptr;
  ctx->dev.inst_trace = ctx->inst_trace.ptr;
  ctx->dev.inst_shade = ctx->inst_shade.ptr;
  ctx->dev.root_ref = root;
  ctx->dev.cache_pairs = cached;
  ctx->pair_count = (uint32_t)all.size();
  ctx->wf_geom_dirty = true;
  ctx->dev.flags = moderate ? RT_SCENE_BOXES_MODERATE : 0u;
  return RT_OK;
}

// Kernel 9's HBM state for this launch shape: result buffer, path state, the contiguous geometry copy.  Kept for the
// next launch.  RT_ERR_OUT_OF_MEMORY when the device cannot give it (the caller decides what that means).
#define RT_TUNE_MAX_UPDATES 256u
hipError_t wf_reserve(rt_ctx *ctx, DevBuf<uint8_t> &buf, size_t bytes) {
  if (ctx->fail_allocs && !(buf.ptr && bytes <= buf.count)) { // test hook: this allocation "fails"
    --ctx->fail_allocs;
    buf.release();
    return hipErrorOutOfMemory;
  }
  return buf.reserve(bytes);
}

void release_wavefront(rt_ctx *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  ctx->wf_state.release();
  ctx->wf_sr.release();
  ctx->wf_geom.release();
  ctx->wf = rt_wf_buffers{};
  ctx->wf_tile_slots = 0;
  ctx->wf_capacity = 0;
  ctx->wf_geom_dirty = true;
}

int prepare_wavefront(rt_ctx *ctx, const rt_render_params &pc, uint32_t frame_count) {
  if (pc.max_depth >= (1u << 28))
    return fail(RT_ERR_UNSUPPORTED, "rt_render: kernel 9 keeps flags in the top bits of the bounce counter (max_depth < 2^28)");
  // HBM state of the pipeline, sized for this launch shape and kept for the next one
  const uint32_t rows = rt_virtual_rows(pc);
  const uint64_t tile_slots = (uint64_t)((pc.image_width + 7u) >> 3) * ((rows + 7u) >> 3) * 64u;
  const uint64_t spp = rt_wf_entries(pc, frame_count); // entries of work per pixel (rt_wavefront.hip)
  if (spp > 0xFFFFFFFFull)
    return fail(RT_ERR_UNSUPPORTED, "rt_render_frames: more than 2^32 samples per pixel in one launch");
  // paths in flight: a sixth of the chunk's paths (shards of 6.6e7 / 1.3e8 paths, ms per launch with 1/8, 1/7, 1/6,
  // 1/5, 1/4, 1/3: 38.8 / 38.3 / 37.6 / 37.6 / 38.1 / 37.9 and 69.5 / 68.6 / 68.5 / 69.1 / 70.4 / 71.4; earlier, on C3 at 1080p: 64 spp = 1.3e8 paths: 16 Mi slots
  // 3960, 32 Mi 3940, 64 Mi 3730 Mrays/s; 256 spp = 5.3e8 paths: 32 Mi 4510, 64 Mi 4580), at most the budget
  // The result buffer first: everything in one chunk if the budget allows, else chunks of whole samples.  If the
  // device cannot give that much (other tenants), the budget is halved until it can, and stays there.
  uint64_t chunk_paths;
  while (true) {
    chunk_paths = tile_slots * spp;
    if (chunk_paths * 16 > ctx->wf_sr_bytes)
      chunk_paths = ctx->wf_sr_bytes / 16 / tile_slots * tile_slots;
    if (chunk_paths < tile_slots)
      chunk_paths = tile_slots; // at least one sample per chunk
    if (ctx->wf_sr.ptr && ctx->wf.sr_paths >= chunk_paths)
      break;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const hipError_t ae = wf_reserve(ctx, ctx->wf_sr, (size_t)chunk_paths * 16);
    ctx->wf.sr = reinterpret_cast<float4 *>(ctx->wf_sr.ptr);
    ctx->wf.sr_paths = ae == hipSuccess ? chunk_paths : 0;
    if (ae == hipSuccess)
      break;
    (void)hipGetLastError(); // (the failed allocation's sticky error)
    if (chunk_paths == tile_slots)
      return fail(RT_ERR_OUT_OF_MEMORY, "rt_render: no device memory for kernel 9's result buffer (one sample per pixel)");
    ctx->wf_sr_bytes = chunk_paths * 16 / 2;
  }
  uint64_t want_slots = (chunk_paths / 6 + 255) / 256 * 256; // (a 6.6e7-path launch: 1/2 ... 1/8 level, 1/16 -20 %)
  if (want_slots < (1u << 20))
    want_slots = chunk_paths < (1u << 20) ? chunk_paths : (1u << 20);
  if (ctx->wf_tuning.persistent != 0u) { // no rounds: a window of slots per resident trace wave, whatever the launch's size
    const uint64_t windows = rt_wf_persistent_slots(ctx->cus);
    want_slots = chunk_paths < windows ? (chunk_paths + 511) / 512 * 512 : windows;
  }
  if (want_slots > ctx->wf_slots)
    want_slots = ctx->wf_slots;
  if (want_slots < 256)
    want_slots = 256;
  if (!ctx->wf_h_ctrl)
    HIP_TRY(hipHostMalloc((void **)&ctx->wf_h_ctrl, RT_WF_CTRL_WORDS * sizeof(uint32_t), hipHostMallocDefault));
  if (!ctx->wf_state.ptr || ctx->wf_capacity < want_slots || ctx->wf_tile_slots < tile_slots) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const uint64_t ts = tile_slots > ctx->wf_tile_slots ? tile_slots : ctx->wf_tile_slots;
    const uint32_t sl = (uint32_t)(want_slots > ctx->wf_capacity ? want_slots : ctx->wf_capacity);
    const uint32_t pool = rt_wf_pool_records(ctx->cus);
    ctx->wf_capacity = 0, ctx->wf_tile_slots = 0; // (a failed reallocation leaves nothing behind)
    HIP_TRY(wf_reserve(ctx, ctx->wf_state, rt_wf_state_bytes(sl, ts, pool)));
    rt_wf_carve(ctx->wf_state.ptr, sl, ts, pool, ctx->wf);
    ctx->wf_tile_slots = ts;
    ctx->wf_capacity = sl;
  }
  ctx->wf.slots = (uint32_t)want_slots; // this launch's share of the allocation
  if (ctx->wf_geom_dirty || !ctx->wf_geom.ptr) {
    // the trace kernel's cooperative fetch addresses pair records and triangles as 32-bit offsets from one base
    const size_t pair_bytes = ((size_t)ctx->pair_count * sizeof(rt_dev_pair) + 255) & ~(size_t)255;
    const size_t tri_bytes = ctx->tris.count * sizeof(rt_dev_tri);
    if (pair_bytes + tri_bytes + 256 >= (1ull << 32))
      return fail(RT_ERR_UNSUPPORTED, "rt_render: kernel 9 addresses node records and triangles with 32-bit offsets "
                                      "(scene too large)");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(wf_reserve(ctx, ctx->wf_geom, pair_bytes + tri_bytes + 64));
    if (ctx->pair_count)
      HIP_TRY(hipMemcpyAsync(ctx->wf_geom.ptr, ctx->pairs.ptr, (size_t)ctx->pair_count * sizeof(rt_dev_pair),
                             hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->wf_geom.ptr + pair_bytes, ctx->tris.ptr, tri_bytes, hipMemcpyDeviceToDevice, ctx->stream));
    ctx->wf_geom_desc.base = ctx->wf_geom.ptr;
    ctx->wf_geom_desc.pairs_off = 0;
    ctx->wf_geom_desc.tris_off = (uint32_t)pair_bytes;
    ctx->wf_geom_desc.bytes = (uint32_t)(pair_bytes + tri_bytes);
    ctx->wf_geom_dirty = false;
  }
  return RT_OK;
}

} // namespace

extern "C" {

const char *rt_last_error_string(void) { return g_last_error.c_str(); }
// (rt_multi.cpp reports through the same channel)
void rt_internal_set_last_error(const char *msg) { g_last_error = msg ? msg : ""; }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int rt_create(int device_id, rt_ctx **out_ctx) {
  if (!out_ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_create: out_ctx is NULL");
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(RT_ERR_NO_DEVICE, "rt_create: no HIP device (this library has no CPU path)");
  if (device_id < 0 || device_id >= n)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_create: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  rt_ctx *ctx = new rt_ctx();
  ctx->device = device_id;
  hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  ctx->stream = ctx->own_stream;
  if (e == hipSuccess)
    e = ctx->counter.reserve(2); // rays traced; rays suspended (kernel 9)
  if (e == hipSuccess)
    e = ctx->queue.reserve(64);
  if (e == hipSuccess)
    e = hipMemsetAsync(ctx->counter.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream);
  int cus = 0;
  if (e == hipSuccess)
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id);
  ctx->cus = (uint32_t)(cus > 0 ? cus : 0);
  if (e == hipSuccess && ctx->cus == 0)
    e = hipErrorInvalidDevice;
  if (e == hipSuccess) // this device's address of the static scratch array (rt_build_tlas_cluster_noalloc)
    e = rt_tlas_scratch_init(device_id);
  if (e != hipSuccess) {
    ctx->counter.release();
    ctx->queue.release();
    if (ctx->own_stream)
      (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return fail(RT_ERR_HIP, std::string("rt_create: ") + hipGetErrorString(e));
  }
  ctx->dev.root_ref = RT_REF_NONE;
  *out_ctx = ctx;
  return RT_OK;
}

int rt_destroy(rt_ctx *ctx) {
  if (!ctx)
    return RT_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (EventPair &p : ctx->pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (EventPair &p : ctx->free_events) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  ctx->pairs.release();
  ctx->tris.release();
  ctx->inst_trace.release();
  ctx->inst_shade.release();
  ctx->shading.release();
  ctx->texels.release();
  ctx->counter.release();
  ctx->trace_rays.release();
  ctx->trace_hits.release();
  ctx->queue.release();
  ctx->accum_own.release();
  ctx->wf_state.release();
  ctx->wf_sr.release();
  ctx->wf_geom.release();
  if (ctx->wf_h_ctrl)
    (void)hipHostFree(ctx->wf_h_ctrl);
  if (ctx->own_stream)
    (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return RT_OK;
}

int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *sc) {
  if (!ctx || !sc)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_upload_scene: NULL argument");
  if ((sc->tri_capacity && (!sc->tri_geom || !sc->tri_shading || !sc->tri_ids)) ||
      (sc->bvh_node_count && !sc->bvh_nodes) || (sc->blas_count && !sc->blas) ||
      (sc->instance_count && !sc->instances) || (sc->tlas_node_count && !sc->tlas_nodes) ||
      (sc->lambert_count && !sc->lamberts) || (sc->metal_count && !sc->metals) ||
      (sc->dielectric_count && !sc->dielectrics) || (sc->emissive_count && !sc->emissives) ||
      (sc->texture_count && !sc->textures) || (sc->texel_bytes && !sc->texels))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_upload_scene: a non-empty array has a NULL pointer");
  HIP_TRY(hipSetDevice(ctx->device));
  ctx->have_scene = false;
  std::string err;
  if (!rt_flatten_blas(sc, ctx->flat, err))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_upload_scene: " + err);
  if (!rt_flatten_tlas(sc->instances, sc->instance_count, sc->tlas_nodes, sc->tlas_node_count,
                       sc->blas_count, ctx->flat, err))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_upload_scene: " + err);
  if (!rt_validate_materials(sc, sc->instances, sc->instance_count, ctx->flat.inst_live, err))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_upload_scene: " + err);
  ctx->blas_count = sc->blas_count;
  // an error return must not leave copies from the caller's arrays queued (the caller may free them at once)
  auto drained = [&](int code) {
    (void)hipStreamSynchronize(ctx->stream);
    return code;
  };
  int rc;
  if ((rc = upload(ctx, ctx->tris, ctx->flat.tris.data(), ctx->flat.tris.size())) != RT_OK)
    return drained(rc);
  if ((rc = upload(ctx, ctx->shading, sc->tri_shading, sc->tri_capacity)) != RT_OK)
    return drained(rc);
  if ((rc = upload(ctx, ctx->texels, sc->texels, (size_t)sc->texel_bytes)) != RT_OK)
    return drained(rc);
  ctx->dev.tris = ctx->tris.ptr;
  ctx->dev.shading = ctx->shading.ptr;
  ctx->dev.texels = ctx->texels.ptr;
  // host copies of the material tables: the per-instance shading records are resolved against them
  // (now and in rt_update_instances), and rt_update_instances re-validates against their counts
  ctx->material_counts = *sc;
  ctx->h_lamberts.assign(sc->lamberts, sc->lamberts + sc->lambert_count);
  ctx->h_textures.assign(sc->textures, sc->textures + sc->texture_count);
  ctx->h_metals.assign(sc->metals, sc->metals + sc->metal_count);
  ctx->h_dielectrics.assign(sc->dielectrics, sc->dielectrics + sc->dielectric_count);
  ctx->h_emissives.assign(sc->emissives, sc->emissives + sc->emissive_count);
  ctx->h_texels.assign(sc->texels, sc->texels + sc->texel_bytes);
  ctx->material_counts.lamberts = ctx->h_lamberts.data();
  ctx->material_counts.textures = ctx->h_textures.data();
  ctx->texel_bytes = sc->texel_bytes;
  if ((rc = upload_pairs_and_instances(ctx, sc->instances, sc->instance_count)) != RT_OK)
    return drained(rc);
  ctx->flat.tris.clear();
  ctx->flat.tris.shrink_to_fit(); // resident in HBM now
  ctx->have_scene = true;
  ++ctx->scene_epoch; // a measured kernel choice (rt_render) does not carry over to another scene
  return RT_OK;
}

int rt_update_instances(rt_ctx *ctx, const rt_blas_instance *instances, uint32_t instance_count,
                        const rt_tlas_node *tlas_nodes, uint32_t tlas_node_count) {
  if (!ctx || (instance_count && !instances) || (tlas_node_count && !tlas_nodes))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_update_instances: NULL argument");
  if (!ctx->have_scene)
    return fail(RT_ERR_NO_SCENE, "rt_update_instances: no scene uploaded");
  HIP_TRY(hipSetDevice(ctx->device));
  // Validation works on the side: a rejected update leaves the scene that is on the device, and the host
  // description of it (ctx->flat: stack depth, instance table), exactly as they were.
  rt_flat_scene &f = ctx->flat;
  std::vector<rt_dev_pair> old_pairs = std::move(f.tlas_pairs);
  std::vector<rt_dev_inst> old_insts = std::move(f.insts);
  std::vector<uint8_t> old_live = std::move(f.inst_live);
  const uint32_t old_root = f.root_ref, old_depth = f.tlas_depth, old_stack = f.stack_needed;
  std::string err;
  bool ok = rt_flatten_tlas(instances, instance_count, tlas_nodes, tlas_node_count, ctx->blas_count, f, err) &&
            rt_validate_materials(&ctx->material_counts, instances, instance_count, f.inst_live, err);
  if (ok && f.stack_needed > RT_MAX_STACK) {
    ok = false;
    err = "scene needs a traversal stack of " + std::to_string(f.stack_needed) + " entries; the kernels hold " +
          std::to_string(RT_MAX_STACK);
  }
  if (!ok) {
    f.tlas_pairs = std::move(old_pairs), f.insts = std::move(old_insts), f.inst_live = std::move(old_live);
    f.root_ref = old_root, f.tlas_depth = old_depth, f.stack_needed = old_stack;
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_update_instances: " + err);
  }
  // kernels in flight still read the old arrays
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // From here on device buffers may be reallocated: until the upload has finished there is no scene to render
  // (a failure below — out of memory — leaves have_scene false, and rt_render refuses instead of reading freed memory).
  ctx->have_scene = false;
  const int rc = upload_pairs_and_instances(ctx, instances, instance_count);
  if (rc != RT_OK)
    return rc;
  ctx->have_scene = true;
  ++ctx->tune.updates;
  return RT_OK;
}

int rt_set_stream(rt_ctx *ctx, void *hip_stream) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_stream: NULL context");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int rc = drain_events(ctx); // the timing events were recorded on the old stream
  if (rc != RT_OK)
    return rc;
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  return RT_OK;
}

int rt_set_camera(rt_ctx *ctx, const rt_camera_uniforms *camera) {
  if (!ctx || !camera)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_camera: NULL argument");
  ctx->camera = *camera;
  ctx->have_camera = true;
  return RT_OK;
}

int rt_resize(rt_ctx *ctx, uint32_t width, uint32_t height) {
  if (!ctx || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 31))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_resize: bad extent");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx->accum_own.reserve((size_t)width * height * 4));
  ctx->accum = ctx->accum_own.ptr;
  ctx->accum_bound = false;
  ctx->width = width;
  ctx->height = height;
  return rt_clear_accum(ctx);
}

int rt_bind_accum(rt_ctx *ctx, void *device_rgba32f, uint32_t width, uint32_t height) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_bind_accum: NULL context");
  if (!device_rgba32f) {
    ctx->accum = ctx->accum_own.ptr;
    ctx->accum_bound = false;
    if (!ctx->accum)
      ctx->width = ctx->height = 0;
    return RT_OK;
  }
  if (width == 0 || height == 0 || ((uintptr_t)device_rgba32f & 15u))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_bind_accum: bad extent or pointer not 16-byte aligned");
  ctx->accum = static_cast<float *>(device_rgba32f);
  ctx->accum_bound = true;
  ctx->width = width;
  ctx->height = height;
  return RT_OK;
}

int rt_clear_accum(rt_ctx *ctx) {
  if (!ctx || !ctx->accum)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_clear_accum: no image");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemsetAsync(ctx->accum, 0, (size_t)ctx->width * ctx->height * 16, ctx->stream));
  return RT_OK;
}

// Regeneration threshold of the persistent kernels when the caller set none.  Measured
// (tools/bench_lowspp.py): small scenes, whose rays finish within a few steps of each other, want
// rare, well-filled regeneration phases (48 idle lanes: Cornell 1280x720 1 spp 1.02 -> 0.58 ms,
// 64 spp 38.6 -> 36.5 ms); deep trees, whose rays finish far apart, want early ones (16: C3
// 75 ms vs 84 ms at 48).
// With many samples per launch the small scenes do better at 32 (tools/sweep_idle.sh, 64 spp, depth 8: C1 7360 ->
// 7685, C4 2579 -> 2635 Mrays/s).
static uint32_t effective_regen_idle(const rt_ctx *ctx, const rt_render_params &pc) {
  if (ctx->regen_idle)
    return ctx->regen_idle;
  if (ctx->pair_count > 16384u)
    return 16u;
  return pc.sqrt_spp * pc.sqrt_spp <= 16u ? 48u : 32u;
}

int rt_render(rt_ctx *ctx, const rt_render_params *params) { return rt_render_frames(ctx, params, 1u); }

int rt_render_frames(rt_ctx *ctx, const rt_render_params *params, uint32_t frame_count) {
  if (!ctx || !params)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: NULL argument");
  if (frame_count == 0 || frame_count > (1u << 24) || params->frame_index + frame_count < params->frame_index ||
      params->accum_count + frame_count < params->accum_count)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_frames: 1 .. 2^24 frames, frame and accumulation counters must not wrap");
  if (!ctx->have_scene)
    return fail(RT_ERR_NO_SCENE, "rt_render: no scene uploaded");
  if (!ctx->have_camera)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: no camera set");
  rt_render_params pc = *params;
  if (pc.image_width == 0 || pc.image_height == 0 || pc.sqrt_spp == 0 || pc.sqrt_spp > 4096 ||
      pc.rng_mode > RT_RNG_PCG || pc.aabb_mode > RT_AABB_RCP || (pc.flags & ~RT_RENDER_FLAGS_ALL))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: bad parameters");
  if (!ctx->accum || ctx->width != pc.image_width || ctx->height != pc.image_height) {
    if (ctx->accum_bound)
      return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: extent differs from the bound image");
    int rc = rt_resize(ctx, pc.image_width, pc.image_height);
    if (rc != RT_OK)
      return rc;
  }
  if (pc.row_begin == 0 && pc.row_end == 0)
    pc.row_end = pc.image_height;
  if (pc.row_end > pc.image_height)
    pc.row_end = pc.image_height;
  if (pc.row_begin >= pc.row_end)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: empty row range");
  // interleaved launches (rt_types.h): the stride counts 8-row groups from row_begin on; 8 * stride must not wrap
  // (rt_virtual_rows divides by it), and no image has more than 2^16 groups of its own
  if (pc.row_group_stride > (1u << 16))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_render: row_group_stride exceeds 65536 groups");
  HIP_TRY(hipSetDevice(ctx->device));
  // Kernel choice when the caller set none (tools/bench_lowspp.py): the vote-scheduled persistent kernel, except
  // for a tiny scene rendered with a few short paths per pixel — the reference's live configuration, built-in
  // Cornell box at 1 sample and depth 3 per frame — where every lane runs the same few steps and the plain
  // one-lane-per-pixel kernel is 1.3x faster (0.40 vs 0.53 ms at 1280x720).  All kernels give the same bits.
  // dev.cache_pairs = node pairs reachable from the TLAS root (counted up to RT_CACHE_PAIRS at upload).
  // ... and for the keyed PCG RNG the wavefront pipeline (kernel 9) where it can be faster (tools/bench_wf.py, C3 scene,
  // 1080p, kernel 8 / kernel 9 in Grays/s: 4 spp 2.1 / 1.3, 16 spp 3.4 / 2.8, 36 spp 3.6 / 3.6, 64 spp 3.6 / 4.0,
  // 256 spp 3.65 / 4.6; 3840x2160 at 64 spp 4.55 / 4.6): launches of at least 2e7 paths.  The threshold is low because
  // of the multi-GPU shards (tools/bench_shard.py): a rank's eighth of a 1080p frame at 256 spp is 6.6e7 paths on only
  // 2.6e5 pixels — about one pixel per lane of kernel 8, whose lanes run a pixel's samples one after the other, so it
  // takes as long as the longest pixel (186 ms, against 43 ms for kernel 9, which spreads the samples).
  // Which of the two wins depends on the scene, not only on the launch shape — closed Cornell box (every path runs to
  // full depth, short traversals): kernel 8 9.2 against 6.6 Grays/s; teapot under the sky 6.0 against 8.3; 2025
  // instances 3.2 against 4.05 (tools/bench_configs.py, 1080p) — so an eligible shape is MEASURED: its first launch
  // runs kernel 9, its second kernel 8, and from the third on the faster of the two.  The images are the same bits
  // either way.  The measurement is repeated when the scene, the instances or the launch shape change.
  // End of round 3: kernel 9 without rounds (rt_wavefront.hip, PERSIST) has no per-round costs left, and wins from ONE
  // sample per pixel on (tools/bench_spp_sweep.py, 1080p, kernel 8 / kernel 9 in Mrays/s — C3: 1 spp 1531 / 1667, 4 spp
  // 2111 / 3275, 16 spp 3397 / 4499, 64 spp 3586 / 5173; Cornell box 3971 / 4985, 4652 / 8282, 8021 / 9593, 8981 / 9979;
  // teapot 3015 / 3257, 4376 / 7508, 5301 / 10928, 5818 / 13139): the threshold is now 2e6 paths, the choice still measured.
  uint32_t variant = ctx->variant;
  uint32_t tuning = 0; // 9 / 8: this launch is the timed trial of that kernel
  bool paths_eligible = false; // this launch's kernel came from the measured choice
  if (ctx->variant_auto) {
    const uint64_t px = (uint64_t)pc.image_width * (pc.row_end - pc.row_begin) / (pc.row_group_stride > 1u ? pc.row_group_stride : 1u);
    // independent units of work the pipeline can keep in flight: samples (keyed PCG) or frames (Wang) per pixel
    const uint64_t paths = px * rt_wf_entries(pc, frame_count);
    // the kernel that renders frame by frame: the megakernel, or the plain one for the reference's live configuration
    const uint32_t per_frame = (ctx->dev.cache_pairs <= 256u && pc.sqrt_spp <= 2u && pc.max_depth <= 3u) ? 0u : 8u;
    if (per_frame == 0u && frame_count == 1u)
      variant = 0u;
    else if (pc.max_depth > 0u && pc.max_depth < (1u << 28) && paths >= 2000000ull) {
      uint32_t size_class = 0; // bit length of the pair count: instance edits keep it, another scene rarely does
      for (uint32_t n = ctx->pair_count; n; n >>= 1)
        ++size_class;
      const uint32_t key[10] = {ctx->scene_epoch, size_class,  pc.image_width, pc.image_height,      pc.sqrt_spp | (frame_count << 8),
                                pc.max_depth,     pc.row_begin, pc.row_end,    pc.row_group_stride, pc.aabb_mode | (pc.rng_mode << 4) | (pc.flags << 8)};
      if (memcmp(key, ctx->tune.key, sizeof key) != 0 || ctx->tune.updates >= RT_TUNE_MAX_UPDATES) {
        memcpy(ctx->tune.key, key, sizeof key);
        ctx->tune.phase = 0;
        ctx->tune.updates = 0;
        ctx->tune.watched_ms = 0.f;
        ctx->tune.warmed = false;
      }
      // Drift: the camera, a moved instance or another tenant can change what the launch costs without
      // changing its key.  If the launch before this one ran the chosen kernel on this key and its time — known without
      // waiting: its events have completed — is off the trial's by more than 1.3x, the trial is repeated.
      if (ctx->tune.phase == 2) {
        if (!ctx->pending.empty()) { // finished launches give up their times; a running one is not waited for
          if (hipEventQuery(ctx->pending.back().b) == hipSuccess)
            (void)drain_events(ctx);
          else
            (void)hipGetLastError(); // (hipErrorNotReady)
        }
        const float was = ctx->tune.choice == 9u ? ctx->tune.ms9 : ctx->tune.ms8, now = ctx->tune.watched_ms;
        if (was > 0.f && now > 0.f && (now > 1.3f * was || was > 1.3f * now))
          ctx->tune.phase = 0;
        ctx->tune.watched_ms = 0.f;
      }
      if (ctx->tune.phase == 0)
        variant = 9u, tuning = ctx->tune.warmed ? 9u : 1u; // (1: kernel 9's untimed first launch)
      else if (ctx->tune.phase == 1)
        variant = per_frame, tuning = 8u;
      else
        variant = ctx->tune.choice == 9u ? 9u : per_frame, paths_eligible = true;
    } else
      variant = per_frame;
  }
  bool wavefront = (variant & 0xFFu) == 9u;
  if (wavefront) {
    const int prc = prepare_wavefront(ctx, pc, frame_count);
    if (prc != RT_OK) {
      // Automatic choice: a kernel 9 that cannot have its memory (other tenants on the GPU, a tiny budget) is simply
      // not eligible — kernel 8 renders the same image with none of it.  A caller that NAMED kernel 9 gets the error.
      if (!ctx->variant_auto || prc != RT_ERR_OUT_OF_MEMORY)
        return prc;
      (void)hipGetLastError();
      release_wavefront(ctx);
      ctx->tune.phase = 2, ctx->tune.choice = 8u, ctx->tune.ms9 = ctx->tune.ms8 = 0.f, ctx->tune.warmed = false;
      variant = (ctx->dev.cache_pairs <= 256u && pc.sqrt_spp <= 2u && pc.max_depth <= 3u) ? 0u : 8u, tuning = 0u, wavefront = false;
    }
  }
  EventPair ev;
  int rc = begin_timed(ctx, ev);
  if (rc != RT_OK)
    return rc;
  hipError_t le = hipSuccess;
  if (wavefront) { // bits 8-15: kernel 9's own knobs
    le = rt_launch_render_wavefront(ctx->dev, ctx->wf_geom_desc, ctx->camera, pc, frame_count, ctx->accum, ctx->counter.ptr,
                                    ctx->wf, ctx->wf_h_ctrl, variant | (ctx->variant_auto ? 0u : ctx->regen_idle << 8),
                                    ctx->wf_tuning, &ctx->wf_counts, ctx->cus, ctx->stream);
  } else { // frame by frame, as the reference's loop does (PathTracer.cpp:396-411)
    rt_render_params pf = pc;
    for (uint32_t k = 0; k < frame_count && le == hipSuccess; ++k) {
      pf.frame_index = pc.frame_index + k, pf.accum_count = pc.accum_count + k;
      le = rt_launch_render(variant, ctx->dev, ctx->camera, pf, ctx->accum, ctx->counter.ptr, ctx->queue.ptr,
                            effective_regen_idle(ctx, pf), ctx->cus, ctx->stream);
    }
  }
  if (le != hipSuccess) { // the event pair goes back to the pool, `pending` stays consistent
    (void)hipStreamSynchronize(ctx->stream); // (frames of a batch that were launched before the failure)
    ctx->free_events.push_back(ev);
    return fail(RT_ERR_HIP, std::string("rt_render: launch failed: ") + hipGetErrorString(le));
  }
  ctx->last_kernel = variant & 0xFFu;
  rc = end_timed(ctx, ev);
  if (rc == RT_OK && ctx->variant_auto && !tuning && ctx->tune.phase == 2 && (wavefront || (variant & 0xFFu) == 8u || (variant & 0xFFu) == 0u))
    ctx->pending.back().watch = paths_eligible; // (only launches that went through the measured choice)
  if (rc == RT_OK && tuning) { // a trial launch: wait for it and keep its time
    if ((rc = drain_events(ctx)) != RT_OK)
      return rc;
    if (tuning == 1u) {
      ctx->tune.warmed = true;
    } else if (tuning == 9u) {
      ctx->tune.ms9 = (float)ctx->last_ms;
      ctx->tune.phase = 1;
    } else {
      ctx->tune.ms8 = (float)ctx->last_ms;
      ctx->tune.choice = ctx->tune.ms9 <= ctx->tune.ms8 ? 9u : 8u;
      ctx->tune.phase = 2;
      if (ctx->tune.choice == 8u) { // kernel 9 lost: its gigabytes of path state go back to the device
        release_wavefront(ctx);
        ctx->tune.warmed = false;
      }
    }
  }
  return rc;
}

int rt_trace_closest_device(rt_ctx *ctx, const void *device_rays, uint64_t n, void *device_hits,
                            uint32_t aabb_mode) {
  if (!ctx || (n && (!device_rays || !device_hits)) || aabb_mode > RT_AABB_RCP)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest_device: bad argument");
  if (!ctx->have_scene)
    return fail(RT_ERR_NO_SCENE, "rt_trace_closest: no scene uploaded");
  if (((uintptr_t)device_rays & 15u) || ((uintptr_t)device_hits & 15u))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest_device: buffers must be 16-byte aligned");
  HIP_TRY(hipSetDevice(ctx->device));
  EventPair ev;
  int rc = begin_timed(ctx, ev);
  if (rc != RT_OK)
    return rc;
  const hipError_t le = rt_launch_trace(ctx->dev, static_cast<const rt_ray *>(device_rays), n,
                                        static_cast<rt_hit *>(device_hits), aabb_mode, ctx->counter.ptr, ctx->stream);
  if (le != hipSuccess) {
    ctx->free_events.push_back(ev);
    return fail(RT_ERR_HIP, std::string("rt_trace_closest_device: launch failed: ") + hipGetErrorString(le));
  }
  return end_timed(ctx, ev);
}

int rt_trace_closest(rt_ctx *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits, uint32_t aabb_mode) {
  if (!ctx || (n && (!rays || !hits)))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_trace_closest: NULL argument");
  if (n == 0)
    return RT_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  // staging buffers of the context: they grow to the largest batch seen and stay (no allocation per call)
  HIP_TRY(hipStreamSynchronize(ctx->stream)); // a growing buffer is freed: nothing in flight may still use it
  if (ctx->trace_rays.reserve(n) != hipSuccess || ctx->trace_hits.reserve(n) != hipSuccess)
    return fail(RT_ERR_OUT_OF_MEMORY, "rt_trace_closest: hipMalloc failed");
  rt_ray *d_rays = ctx->trace_rays.ptr;
  rt_hit *d_hits = ctx->trace_hits.ptr;
  hipError_t e;
  int rc = RT_OK;
  e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    rc = rt_trace_closest_device(ctx, d_rays, n, d_hits, aabb_mode);
  if (e == hipSuccess && rc == RT_OK)
    e = hipMemcpyAsync(hits, d_hits, n * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream);
  hipError_t es = hipStreamSynchronize(ctx->stream);
  if (rc != RT_OK)
    return rc;
  if (e != hipSuccess || es != hipSuccess)
    return fail(RT_ERR_HIP, std::string("rt_trace_closest: ") + hipGetErrorString(e != hipSuccess ? e : es));
  return RT_OK;
}

int rt_read_accum(rt_ctx *ctx, float *host_rgba32f) {
  if (!ctx || !host_rgba32f || !ctx->accum)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_read_accum: no image or NULL destination");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipMemcpyAsync(host_rgba32f, ctx->accum, (size_t)ctx->width * ctx->height * 16,
                         hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_accum_device_ptr(rt_ctx *ctx, void **out) {
  if (!ctx || !out)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_accum_device_ptr: NULL argument");
  *out = ctx->accum;
  return RT_OK;
}

int rt_sync(rt_ctx *ctx) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_sync: NULL context");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_get_stats(rt_ctx *ctx, rt_stats *out) {
  if (!ctx || !out)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_get_stats: NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int rc = drain_events(ctx);
  if (rc != RT_OK)
    return rc;
  unsigned long long rays = 0;
  HIP_TRY(hipMemcpy(&rays, ctx->counter.ptr, sizeof rays, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->rays_traced = rays;
  out->last_kernel_ms = ctx->last_ms;
  out->total_kernel_ms = ctx->total_ms;
  out->launches = ctx->launches;
  out->stack_depth_needed = ctx->flat.stack_needed;
  out->last_kernel = ctx->last_kernel;
  out->device_bytes = ctx->pairs.count * sizeof(rt_dev_pair) + ctx->tris.count * sizeof(rt_dev_tri) +
                      ctx->inst_trace.count * sizeof(rt_dev_inst) +
                      ctx->inst_shade.count * sizeof(rt_dev_inst_shade) +
                      ctx->shading.count * sizeof(rt_triangle_shading) + ctx->texels.count;
  return RT_OK;
}

int rt_reset_stats(rt_ctx *ctx) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_reset_stats: NULL context");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int rc = drain_events(ctx);
  if (rc != RT_OK)
    return rc;
  HIP_TRY(hipMemsetAsync(ctx->counter.ptr, 0, 2 * sizeof(unsigned long long), ctx->stream));
  ctx->total_ms = ctx->last_ms = 0.0;
  ctx->launches = 0;
  const bool timing = ctx->wf_counts.timing;
  ctx->wf_counts = rt_wf_counts{};
  ctx->wf_counts.timing = timing;
  return RT_OK;
}

int rt_debug_probe_walk(rt_ctx *ctx, uint32_t blocks, uint32_t steps, uint32_t loads, double *out_ms,
                        uint64_t *out_steps) {
  if (!ctx || !out_ms || !out_steps || blocks == 0 || (loads & 0xFFu) < 2 || (loads & 0xFFu) > 5 || (loads >> 8) > 64)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_probe_walk: bad argument");
  if (!ctx->have_scene)
    return fail(RT_ERR_NO_SCENE, "rt_debug_probe_walk: no scene uploaded");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int rc = drain_events(ctx);
  if (rc != RT_OK)
    return rc;
  unsigned long long before = 0, after = 0;
  HIP_TRY(hipMemcpy(&before, ctx->counter.ptr, sizeof before, hipMemcpyDeviceToHost));
  EventPair ev;
  if ((rc = begin_timed(ctx, ev)) != RT_OK)
    return rc;
  const hipError_t le = rt_launch_probe_walk(ctx->dev, blocks, steps, loads, ctx->counter.ptr, ctx->stream);
  if (le != hipSuccess) {
    ctx->free_events.push_back(ev);
    return fail(RT_ERR_HIP, std::string("rt_debug_probe_walk: launch failed: ") + hipGetErrorString(le));
  }
  if ((rc = end_timed(ctx, ev)) != RT_OK)
    return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if ((rc = drain_events(ctx)) != RT_OK)
    return rc;
  HIP_TRY(hipMemcpy(&after, ctx->counter.ptr, sizeof after, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ctx->counter.ptr, &before, sizeof before, hipMemcpyHostToDevice)); // not rays
  *out_ms = ctx->last_ms;
  *out_steps = after - before;
  return RT_OK;
}

int rt_debug_wavefront_tuning(rt_ctx *ctx, uint32_t suspend_lanes, uint32_t list_percent, uint32_t share,
                              uint32_t finish_paths) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_tuning: NULL context");
  ctx->wf_tuning.suspend_lanes = suspend_lanes;
  ctx->wf_tuning.list_percent = list_percent;
  ctx->wf_tuning.share = share;
  ctx->wf_tuning.finish_paths = finish_paths;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old settings
  return RT_OK;
}

int rt_debug_wavefront_compaction(rt_ctx *ctx, uint32_t enable) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_compaction: NULL context");
  ctx->wf_tuning.compact = enable ? 0xFFFFFFFFu : 0u;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old setting
  return RT_OK;
}

int rt_debug_wavefront_persistent(rt_ctx *ctx, uint32_t enable) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_persistent: NULL context");
  ctx->wf_tuning.persistent = enable ? 0xFFFFFFFFu : 0u;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old setting
  return RT_OK;
}

int rt_debug_wavefront_record_cache(rt_ctx *ctx, uint32_t enable) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_record_cache: NULL context");
  ctx->wf_tuning.record_cache = enable ? 0xFFFFFFFFu : 0u;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old setting
  return RT_OK;
}

int rt_debug_wavefront_bands(rt_ctx *ctx, uint32_t mode) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_bands: NULL context");
  if (mode > 2u)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_bands: mode is 0 (never), 1 (always) or 2 (by launch and scene size)");
  ctx->wf_tuning.bands = mode == 2u ? 0xFFFFFFFFu : mode;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old setting
  return RT_OK;
}

int rt_debug_wavefront_counts(rt_ctx *ctx, uint64_t *rounds, uint64_t *list_rounds, uint64_t *suspended_rays,
                              uint64_t *finished_paths) {
  if (!ctx || !rounds || !list_rounds || !suspended_rays || !finished_paths)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_counts: NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  unsigned long long susp = 0;
  HIP_TRY(hipMemcpy(&susp, ctx->counter.ptr + 1, sizeof susp, hipMemcpyDeviceToHost));
  *rounds = ctx->wf_counts.rounds;
  *list_rounds = ctx->wf_counts.list_rounds;
  *suspended_rays = susp;
  *finished_paths = ctx->wf_counts.finished_paths;
  return RT_OK;
}

int rt_debug_fast_div(rt_ctx *ctx, const float *n, const float *d, uint64_t count, float *q_fast,
                      float *q_ieee) {
  if (!ctx || !n || !d || !q_fast || !q_ieee || count > (1ull << 28))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_fast_div: bad argument");
  if (count == 0)
    return RT_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  struct Scoped : DevBuf<float> {
    ~Scoped() { release(); }
  } bn, bd, bf, bi;
  HIP_TRY(bn.reserve(count));
  HIP_TRY(bd.reserve(count));
  HIP_TRY(bf.reserve(count));
  HIP_TRY(bi.reserve(count));
  HIP_TRY(hipMemcpyAsync(bn.ptr, n, count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(bd.ptr, d, count * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(rt_launch_debug_fast_div(bn.ptr, bd.ptr, count, bf.ptr, bi.ptr, ctx->stream));
  HIP_TRY(hipMemcpyAsync(q_fast, bf.ptr, count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(q_ieee, bi.ptr, count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_build_tlas_cluster(rt_ctx *ctx, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                          uint32_t *out_node_count, float *kernel_ms) {
  if (!ctx || !leaves || !out_nodes || !out_node_count || n == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster: bad argument");
  if (n > RT_TLAS_GPU_MAX_LEAVES)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster: more than RT_TLAS_GPU_MAX_LEAVES leaves");
  for (uint32_t k = 0; k < n; ++k) {
    if (leaves[k].left_right != 0)
      return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster: input node is not a leaf");
    for (int c = 0; c < 3; ++c) // the kernel's scan relies on finite boxes (rt_tlas_build.hip)
      if (!std::isfinite(leaves[k].aabb_min[c]) || !std::isfinite(leaves[k].aabb_max[c]))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster: instance boxes are not finite");
  }
  HIP_TRY(hipSetDevice(ctx->device));
  struct ScopedN : DevBuf<rt_tlas_node> {
    ~ScopedN() { release(); }
  } d_leaves, d_out;
  struct ScopedC : DevBuf<uint32_t> {
    ~ScopedC() { release(); }
  } d_count;
  HIP_TRY(d_leaves.reserve(n));
  HIP_TRY(d_out.reserve(2 * (size_t)n));
  HIP_TRY(d_count.reserve(1));
  HIP_TRY(hipMemcpyAsync(d_leaves.ptr, leaves, (size_t)n * sizeof(rt_tlas_node), hipMemcpyHostToDevice, ctx->stream));
  struct ScopedEvents {
    hipEvent_t a = nullptr, b = nullptr;
    ~ScopedEvents() {
      if (a)
        (void)hipEventDestroy(a);
      if (b)
        (void)hipEventDestroy(b);
    }
  } ev;
  HIP_TRY(hipEventCreate(&ev.a));
  HIP_TRY(hipEventCreate(&ev.b));
  HIP_TRY(hipEventRecord(ev.a, ctx->stream));
  HIP_TRY(rt_launch_tlas_cluster(d_leaves.ptr, n, d_out.ptr, d_count.ptr, ctx->stream));
  HIP_TRY(hipEventRecord(ev.b, ctx->stream));
  uint32_t count = 0;
  HIP_TRY(hipMemcpyAsync(&count, d_count.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(out_nodes, d_out.ptr, 2 * (size_t)n * sizeof(rt_tlas_node), hipMemcpyDeviceToHost,
                         ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (kernel_ms)
    HIP_TRY(hipEventElapsedTime(kernel_ms, ev.a, ev.b));
  if (count == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster: instance boxes are not finite");
  *out_node_count = count;
  return RT_OK;
}

int rt_set_wavefront_budget(rt_ctx *ctx, uint32_t paths_in_flight, uint64_t result_bytes) {
  if (!ctx || paths_in_flight < 256 || paths_in_flight > (1u << 30) || result_bytes < (1ull << 20))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_wavefront_budget: 256 .. 2^30 paths, at least 1 MiB of results");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  ctx->wf_slots = paths_in_flight;
  ctx->wf_sr_bytes = result_bytes;
  ctx->wf_state.release(); // re-sized by the next launch of kernel 9
  ctx->wf_sr.release();
  ctx->wf = rt_wf_buffers{};
  ctx->wf_tile_slots = 0;
  ctx->wf_capacity = 0;
  ctx->tune = rt_ctx::Tune{}; // the measured kernel choice was made with the old budget
  return RT_OK;
}

int rt_build_tlas_cluster_noalloc(rt_ctx *ctx, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                                  uint32_t *out_node_count) {
  if (!ctx || !leaves || !out_nodes || !out_node_count || n == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster_noalloc: bad argument");
  if (n > RT_TLAS_GPU_MAX_LEAVES)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster_noalloc: " + std::to_string(n) + " leaves exceed the " +
                                             std::to_string(RT_TLAS_GPU_MAX_LEAVES) + " (RT_TLAS_GPU_MAX_LEAVES) the static "
                                             "scratch buffer holds");
  for (uint32_t k = 0; k < n; ++k) {
    if (leaves[k].left_right != 0)
      return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster_noalloc: input node is not a leaf");
    for (int c = 0; c < 3; ++c) // the kernel's scan relies on finite boxes (rt_tlas_build.hip)
      if (!std::isfinite(leaves[k].aabb_min[c]) || !std::isfinite(leaves[k].aabb_max[c]))
        return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster_noalloc: instance boxes are not finite");
  }
  HIP_TRY(hipSetDevice(ctx->device));
  uint32_t count = 0;
  HIP_TRY(rt_run_tlas_cluster_noalloc(ctx->device, leaves, n, out_nodes, &count, ctx->stream));
  if (count == 0)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_build_tlas_cluster_noalloc: instance boxes are not finite");
  *out_node_count = count;
  return RT_OK;
}

int rt_debug_wavefront_timing(rt_ctx *ctx, int enable, double *out_ms5, uint64_t *out_launches) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_wavefront_timing: NULL context");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (out_ms5) {
    out_ms5[0] = ctx->wf_counts.trace_ms, out_ms5[1] = ctx->wf_counts.shade_ms, out_ms5[2] = ctx->wf_counts.finish_ms;
    out_ms5[3] = ctx->wf_counts.resolve_ms, out_ms5[4] = ctx->wf_counts.fill_ms;
  }
  if (out_launches)
    *out_launches = ctx->wf_counts.timed_launches;
  ctx->wf_counts.timing = enable != 0;
  return RT_OK;
}

int rt_debug_fail_wavefront_allocs(rt_ctx *ctx, uint32_t count) {
  if (!ctx)
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_debug_fail_wavefront_allocs: NULL context");
  ctx->fail_allocs = count;
  return RT_OK;
}

int rt_set_kernel_variant(rt_ctx *ctx, uint32_t variant) {
  if (ctx && variant == RT_KERNEL_VARIANT_AUTO) {
    ctx->variant_auto = true, ctx->variant = 8, ctx->regen_idle = 0;
    return RT_OK;
  }
  const uint32_t kernel = variant & 0xFFu;
  if (!ctx || (kernel != 0 && kernel != 8 && kernel != 9))
    return fail(RT_ERR_INVALID_ARGUMENT, "rt_set_kernel_variant: the library holds kernels 0, 8 and 9");
  ctx->variant_auto = false;
  // bits 0-7 kernel, 8-15 regen_idle (0 = automatic); higher bits: schedule knobs of kernel 8 (rt_launch_render)
  ctx->variant = variant & 0xFFFF00FFu;
  ctx->regen_idle = (variant >> 8) & 0xFFu; // 0: chosen by scene size (effective_regen_idle)
  return RT_OK;
}

} // extern "C"
