// rt_multi.cpp — the multi-GPU driver of include/rt_hip.h (rt_mgpu_*): one rt_ctx and one host thread per GPU, the
// scene replicated, every frame cut into groups of 8 rows dealt round-robin to the ranks (rt_render_params.row_begin /
// row_group_stride: rank r renders groups r, r + n, ...), and after every launch ONE exchange of the image — a direct
// ncclAllGather of the packed row groups over RCCL / xGMI — so that every rank holds the complete image, bit-identical
// to the one-GPU image (nothing is re-associated).  SURVEY.md §7 step 8, §8(e); the reference is single-GPU and has no
// counterpart.
//
// RCCL is loaded at run time (dlopen) the first time an exchange needs it: single-GPU users of the library never load
// it, and a process that already holds a copy (PyTorch bundles one) shares that copy instead of mapping a second.
// Two ranks on the same device cannot form an RCCL communicator; for them (a rehearsal of the multi-rank path on one
// GPU, as the tests do) and when RCCL is not available the exchange is peer copies between the ranks' images ordered
// by host barriers.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_hip.h"

extern "C" void rt_internal_set_last_error(const char *msg); // rt_hip_api.cpp

namespace {

thread_local std::string g_mgpu_error;

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string &err) {
    if (AllGather)
      return true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) // a copy the process already holds
      if (!lib)
        lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
      if (!lib)
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      err = std::string("RCCL not found: ") + dlerror();
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !GetErrorString) {
      err = "RCCL library lacks ncclCommInitAll / ncclAllGather";
      AllGather = nullptr;
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// One worker per rank: runs the closures the driver hands it, in order.
struct Worker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, done = true, quit = false;
  int rc = RT_OK;
  std::string err;
  void loop() {
    std::unique_lock<std::mutex> lk(m);
    while (true) {
      cv.wait(lk, [&] { return has_job || quit; });
      if (quit)
        return;
      std::function<int()> j = std::move(job);
      has_job = false;
      lk.unlock();
      const int r = j();
      std::string e = r == RT_OK ? std::string() : (g_mgpu_error.empty() ? std::string(rt_last_error_string()) : g_mgpu_error);
      g_mgpu_error.clear();
      lk.lock();
      rc = r, err = std::move(e), done = true;
      cv.notify_all();
    }
  }
};

// Reusable host barrier for the copy exchange.
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  uint32_t n = 0, waiting = 0, phase = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const uint32_t p = phase;
    if (++waiting == n) {
      waiting = 0, ++phase;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return phase != p; });
    }
  }
};

int mfail(int code, const std::string &msg) {
  g_mgpu_error = msg;
  return code;
}

#define M_HIP(expr)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return mfail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,            \
                   std::string(#expr) + ": " + hipGetErrorString(e_));                       \
  } while (0)

} // namespace

struct rt_mgpu {
  uint32_t n = 0;
  std::vector<int> device;
  std::vector<rt_ctx *> ctx;
  std::vector<hipStream_t> stream;
  std::vector<Worker *> worker;
  // exchange
  bool use_rccl = false;
  std::vector<ncclComm_t> comm;
  std::vector<float *> send, recv; // per rank: [per_rank_groups][8 * W * 4], [n][per_rank_groups][8 * W * 4]
  size_t exchange_floats = 0;      // capacity of `send` in floats
  uint32_t width = 0, height = 0;
  Barrier barrier;
  std::string last_error;
  uint64_t exchanges = 0;
};

extern "C" {

int rt_mgpu_partition(uint32_t height, uint32_t n, uint32_t rank, rt_mgpu_share *out) {
  if (!out || n == 0 || rank >= n || height == 0)
    return RT_ERR_INVALID_ARGUMENT;
  const uint32_t groups = (height + 7u) / 8u;
  memset(out, 0, sizeof *out);
  out->groups_total = groups;
  out->groups_per_rank = (groups + n - 1u) / n; // the all-gather's equal share (the last ranks' tail may be padding)
  out->groups = rank < groups ? (groups - rank + n - 1u) / n : 0u;
  out->row_begin = 8u * rank;
  out->row_end = height;
  out->row_group_stride = n;
  uint32_t rows = 0;
  for (uint32_t g = rank; g < groups; g += n)
    rows += (g * 8u + 8u <= height) ? 8u : height - g * 8u;
  out->rows = rows;
  return RT_OK;
}

} // extern "C"

namespace {

// Runs fn(rank) on every worker and waits; the first failure is reported.
int run_all(rt_mgpu *m, const std::function<int(uint32_t)> &fn) {
  for (uint32_t r = 0; r < m->n; ++r) {
    Worker *w = m->worker[r];
    std::lock_guard<std::mutex> lk(w->m);
    w->job = [fn, r] { return fn(r); };
    w->has_job = true, w->done = false;
    w->cv.notify_all();
  }
  int rc = RT_OK;
  for (uint32_t r = 0; r < m->n; ++r) {
    Worker *w = m->worker[r];
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [&] { return w->done; });
    if (w->rc != RT_OK && rc == RT_OK) {
      rc = w->rc;
      m->last_error = "rank " + std::to_string(r) + " (device " + std::to_string(m->device[r]) + "): " + w->err;
    }
  }
  if (rc != RT_OK)
    rt_internal_set_last_error(m->last_error.c_str());
  return rc;
}

// The exchange after a launch, on rank r's thread.  Image rows of a group are contiguous: 8 * W * 16 bytes.
int exchange(rt_mgpu *m, uint32_t r) {
  if (m->n == 1 && !m->use_rccl)
    return RT_OK;
  const uint32_t W = m->width, H = m->height, n = m->n;
  const uint32_t groups = (H + 7u) / 8u, per = (groups + n - 1u) / n;
  const size_t group_floats = (size_t)8 * W * 4;
  auto group_bytes = [&](uint32_t g) { return (size_t)((g * 8u + 8u <= H) ? 8u : H - g * 8u) * W * 16; };
  void *img_v = nullptr;
  (void)rt_accum_device_ptr(m->ctx[r], &img_v);
  float *img = static_cast<float *>(img_v);
  hipStream_t s = m->stream[r];
  if (m->use_rccl) {
    if (!img)
      return mfail(RT_ERR_INVALID_ARGUMENT, "exchange: rank has no image");
    // pack my groups, all-gather, unpack the others' (everything on my stream, behind my launch)
    for (uint32_t k = 0, g = r; g < groups; ++k, g += n)
      M_HIP(hipMemcpyAsync(m->send[r] + k * group_floats, img + (size_t)g * group_floats, group_bytes(g),
                           hipMemcpyDeviceToDevice, s));
    const ncclResult_t nr = g_rccl.AllGather(m->send[r], m->recv[r], per * group_floats, ncclFloat32, m->comm[r], s);
    if (nr != ncclSuccess)
      return mfail(RT_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr));
    for (uint32_t q = 0; q < n; ++q) {
      if (q == r)
        continue;
      for (uint32_t k = 0, g = q; g < groups; ++k, g += n)
        M_HIP(hipMemcpyAsync(img + (size_t)g * group_floats, m->recv[r] + ((size_t)q * per + k) * group_floats, group_bytes(g),
                             hipMemcpyDeviceToDevice, s));
    }
    M_HIP(hipStreamSynchronize(s));
    return RT_OK;
  }
  // peer copies: every rank's launch has finished (barrier), then I pull the other ranks' groups out of their images
  const hipError_t e0 = hipStreamSynchronize(s); // (no early return before the barriers: the other ranks wait in them)
  m->barrier.wait();
  int rc = e0 == hipSuccess && img ? RT_OK : mfail(RT_ERR_HIP, "exchange: launch failed or rank has no image");
  for (uint32_t q = 0; q < n && rc == RT_OK; ++q) {
    if (q == r)
      continue;
    void *src_v = nullptr;
    if (rt_accum_device_ptr(m->ctx[q], &src_v) != RT_OK || !src_v) {
      rc = mfail(RT_ERR_INVALID_ARGUMENT, "exchange: peer has no image");
      break;
    }
    const float *src = static_cast<const float *>(src_v);
    for (uint32_t g = q; g < groups; g += n) {
      const hipError_t e = m->device[q] == m->device[r]
                               ? hipMemcpyAsync(img + (size_t)g * group_floats, src + (size_t)g * group_floats, group_bytes(g),
                                                hipMemcpyDeviceToDevice, s)
                               : hipMemcpyPeerAsync(img + (size_t)g * group_floats, m->device[r], src + (size_t)g * group_floats,
                                                    m->device[q], group_bytes(g), s);
      if (e != hipSuccess) {
        rc = mfail(RT_ERR_HIP, std::string("exchange copy: ") + hipGetErrorString(e));
        break;
      }
    }
  }
  const hipError_t es = hipStreamSynchronize(s);
  m->barrier.wait(); // nobody renders into rows I am still reading
  if (rc == RT_OK && es != hipSuccess)
    rc = mfail(RT_ERR_HIP, std::string("exchange: ") + hipGetErrorString(es));
  return rc;
}

int ensure_exchange_buffers(rt_mgpu *m, uint32_t r, uint32_t W, uint32_t H) {
  if (!m->use_rccl)
    return RT_OK;
  const uint32_t groups = (H + 7u) / 8u, per = (groups + m->n - 1u) / m->n;
  const size_t need = (size_t)per * 8 * W * 4;
  if (m->send[r] && m->exchange_floats >= need)
    return RT_OK;
  if (m->send[r])
    (void)hipFree(m->send[r]);
  if (m->recv[r])
    (void)hipFree(m->recv[r]);
  m->send[r] = m->recv[r] = nullptr;
  M_HIP(hipMalloc((void **)&m->send[r], need * sizeof(float)));
  M_HIP(hipMalloc((void **)&m->recv[r], need * sizeof(float) * m->n));
  return RT_OK;
}

} // namespace

extern "C" {

int rt_mgpu_create(const int *device_ids, uint32_t n, uint32_t exchange_mode, rt_mgpu **out) {
  if (!device_ids || !out || n == 0 || n > 64 || exchange_mode > RT_MGPU_EXCHANGE_COPY)
    return RT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  rt_mgpu *m = new rt_mgpu();
  m->n = n;
  m->device.assign(device_ids, device_ids + n);
  m->ctx.assign(n, nullptr), m->stream.assign(n, nullptr), m->send.assign(n, nullptr), m->recv.assign(n, nullptr);
  m->comm.assign(n, nullptr);
  m->barrier.n = n;
  bool distinct = true;
  for (uint32_t a = 0; a < n; ++a)
    for (uint32_t b = a + 1; b < n; ++b)
      distinct = distinct && device_ids[a] != device_ids[b];
  for (uint32_t r = 0; r < n; ++r) {
    Worker *w = new Worker();
    w->th = std::thread([w] { w->loop(); });
    m->worker.push_back(w);
  }
  int rc = run_all(m, [m](uint32_t r) {
    int rc2 = rt_create(m->device[r], &m->ctx[r]);
    if (rc2 != RT_OK)
      return rc2;
    M_HIP(hipSetDevice(m->device[r]));
    M_HIP(hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking));
    return rt_set_stream(m->ctx[r], m->stream[r]);
  });
  if (rc == RT_OK && exchange_mode != RT_MGPU_EXCHANGE_COPY && (n > 1 || exchange_mode == RT_MGPU_EXCHANGE_RCCL)) {
    // one communicator per rank, all in this process (ncclCommInitAll); ranks on one device cannot form one
    std::string err;
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (!distinct)
      err = "two ranks share a device: RCCL needs one GPU per rank";
    else if (g_rccl.load(err)) {
      const ncclResult_t nr = g_rccl.CommInitAll(m->comm.data(), (int)n, m->device.data());
      if (nr != ncclSuccess)
        err = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(nr);
      else
        m->use_rccl = true;
    }
    if (!m->use_rccl && exchange_mode == RT_MGPU_EXCHANGE_RCCL) {
      m->last_error = "rt_mgpu_create: " + err;
      rc = RT_ERR_UNSUPPORTED;
    }
  }
  if (rc != RT_OK) {
    const std::string keep = m->last_error;
    rt_mgpu_destroy(m);
    rt_internal_set_last_error(keep.c_str());
    return rc;
  }
  *out = m;
  return RT_OK;
}

int rt_mgpu_destroy(rt_mgpu *m) {
  if (!m)
    return RT_OK;
  (void)run_all(m, [m](uint32_t r) {
    (void)hipSetDevice(m->device[r]);
    if (m->ctx[r]) {
      (void)rt_sync(m->ctx[r]);
      (void)rt_set_stream(m->ctx[r], nullptr);
    }
    if (m->comm[r])
      (void)g_rccl.CommDestroy(m->comm[r]);
    if (m->send[r])
      (void)hipFree(m->send[r]);
    if (m->recv[r])
      (void)hipFree(m->recv[r]);
    if (m->ctx[r])
      (void)rt_destroy(m->ctx[r]);
    if (m->stream[r])
      (void)hipStreamDestroy(m->stream[r]);
    return (int)RT_OK;
  });
  for (Worker *w : m->worker) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
      w->cv.notify_all();
    }
    w->th.join();
    delete w;
  }
  delete m;
  return RT_OK;
}

uint32_t rt_mgpu_ranks(rt_mgpu *m) { return m ? m->n : 0u; }
int rt_mgpu_uses_rccl(rt_mgpu *m) { return m && m->use_rccl ? 1 : 0; }
rt_ctx *rt_mgpu_context(rt_mgpu *m, uint32_t rank) { return m && rank < m->n ? m->ctx[rank] : nullptr; }

int rt_mgpu_upload_scene(rt_mgpu *m, const rt_scene_desc *scene) {
  if (!m || !scene)
    return RT_ERR_INVALID_ARGUMENT;
  return run_all(m, [m, scene](uint32_t r) { return rt_upload_scene(m->ctx[r], scene); });
}

int rt_mgpu_update_instances(rt_mgpu *m, const rt_blas_instance *instances, uint32_t instance_count,
                             const rt_tlas_node *tlas_nodes, uint32_t tlas_node_count) {
  if (!m)
    return RT_ERR_INVALID_ARGUMENT;
  return run_all(m, [=](uint32_t r) { return rt_update_instances(m->ctx[r], instances, instance_count, tlas_nodes, tlas_node_count); });
}

int rt_mgpu_set_camera(rt_mgpu *m, const rt_camera_uniforms *camera) {
  if (!m || !camera)
    return RT_ERR_INVALID_ARGUMENT;
  return run_all(m, [m, camera](uint32_t r) { return rt_set_camera(m->ctx[r], camera); });
}

int rt_mgpu_set_kernel_variant(rt_mgpu *m, uint32_t variant) {
  if (!m)
    return RT_ERR_INVALID_ARGUMENT;
  return run_all(m, [m, variant](uint32_t r) { return rt_set_kernel_variant(m->ctx[r], variant); });
}

int rt_mgpu_clear_accum(rt_mgpu *m) {
  if (!m)
    return RT_ERR_INVALID_ARGUMENT;
  return run_all(m, [m](uint32_t r) { return rt_clear_accum(m->ctx[r]); });
}

int rt_mgpu_resize(rt_mgpu *m, uint32_t width, uint32_t height) {
  if (!m || width == 0 || height == 0)
    return RT_ERR_INVALID_ARGUMENT;
  if ((height + 7u) / 8u < m->n) {
    rt_internal_set_last_error("rt_mgpu_resize: fewer 8-row groups than ranks");
    return RT_ERR_INVALID_ARGUMENT;
  }
  const int rc = run_all(m, [m, width, height](uint32_t r) {
    int rc2 = rt_resize(m->ctx[r], width, height);
    if (rc2 != RT_OK)
      return rc2;
    M_HIP(hipSetDevice(m->device[r]));
    return ensure_exchange_buffers(m, r, width, height);
  });
  if (rc == RT_OK) {
    const uint32_t groups = (height + 7u) / 8u, per = (groups + m->n - 1u) / m->n;
    m->exchange_floats = (size_t)per * 8 * width * 4;
    m->width = width, m->height = height;
  }
  return rc;
}

int rt_mgpu_render_frames(rt_mgpu *m, const rt_render_params *params, uint32_t frame_count) {
  if (!m || !params)
    return RT_ERR_INVALID_ARGUMENT;
  if (params->image_width != m->width || params->image_height != m->height) {
    const int rc = rt_mgpu_resize(m, params->image_width, params->image_height);
    if (rc != RT_OK)
      return rc;
  }
  const rt_render_params base = *params;
  const int rc = run_all(m, [m, base, frame_count](uint32_t r) {
    M_HIP(hipSetDevice(m->device[r]));
    rt_mgpu_share sh;
    (void)rt_mgpu_partition(base.image_height, m->n, r, &sh);
    rt_render_params p = base;
    p.row_begin = sh.row_begin, p.row_end = sh.row_end, p.row_group_stride = sh.row_group_stride;
    const int rc2 = rt_render_frames(m->ctx[r], &p, frame_count);
    std::string err2 = rc2 != RT_OK ? std::string(rt_last_error_string()) : std::string();
    // a rank whose launch failed still takes part in the exchange (barriers, collective), or the others wait for ever
    const int rc3 = exchange(m, r);
    if (rc2 != RT_OK)
      g_mgpu_error = err2;
    return rc2 != RT_OK ? rc2 : rc3;
  });
  if (rc == RT_OK)
    ++m->exchanges;
  return rc;
}

int rt_mgpu_read_accum(rt_mgpu *m, uint32_t rank, float *host_rgba32f) {
  if (!m || rank >= m->n || !host_rgba32f)
    return RT_ERR_INVALID_ARGUMENT;
  int rc = RT_OK;
  Worker *w = m->worker[rank];
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->job = [m, rank, host_rgba32f] { return rt_read_accum(m->ctx[rank], host_rgba32f); };
    w->has_job = true, w->done = false;
    w->cv.notify_all();
  }
  std::unique_lock<std::mutex> lk(w->m);
  w->cv.wait(lk, [&] { return w->done; });
  if ((rc = w->rc) != RT_OK)
    rt_internal_set_last_error(("rank " + std::to_string(rank) + ": " + w->err).c_str());
  return rc;
}

int rt_mgpu_get_stats(rt_mgpu *m, uint32_t rank, rt_stats *out) {
  if (!m || rank >= m->n || !out)
    return RT_ERR_INVALID_ARGUMENT;
  Worker *w = m->worker[rank];
  {
    std::lock_guard<std::mutex> lk(w->m);
    w->job = [m, rank, out] { return rt_get_stats(m->ctx[rank], out); };
    w->has_job = true, w->done = false;
    w->cv.notify_all();
  }
  std::unique_lock<std::mutex> lk(w->m);
  w->cv.wait(lk, [&] { return w->done; });
  return w->rc;
}

} // extern "C"
