/* rt_hip.h — C ABI of librt_hip.so: the MI355X (gfx950) render path.
 *
 * This is the drop-in boundary.  It replaces everything the reference does
 * from `vkCmdDispatch` downwards (src/Renderer.cpp:615-651 and the Slang kernel
 * Assets/Shaders/RayTracing.slang:62-181) plus the device-buffer staging of
 * src/Renderer.cpp:790-853, 867-868, 1007-1011.  Inputs are the reference's own
 * host arrays (rt_types.h); the library flattens them into its device layout.
 *
 * Conventions: every function returns 0 on success or a negative rt_status;
 * the message of the last failure on the calling thread is available from
 * rt_last_error_string().  A context belongs to one GPU and one host thread.
 * rt_render / rt_trace_closest_device are asynchronous on the context's HIP
 * stream; rt_sync, rt_read_accum, rt_trace_closest and rt_get_stats wait.
 * The library fails (RT_ERR_NO_DEVICE) when no HIP device is present — there
 * is no CPU fallback.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = -1,
  RT_ERR_NO_DEVICE = -2,
  RT_ERR_HIP = -3,
  RT_ERR_OUT_OF_MEMORY = -4,
  RT_ERR_NO_SCENE = -5,
  RT_ERR_UNSUPPORTED = -6 /* e.g. tree deeper than the traversal stack */
};

typedef struct rt_stats {
  uint64_t rays_traced;   /* closest-hit queries since rt_reset_stats (primary + bounces) */
  double last_kernel_ms;  /* HIP-event time of the most recent kernel launch */
  double total_kernel_ms; /* sum over launches since rt_reset_stats */
  uint64_t launches;
  uint32_t stack_depth_needed; /* TLAS depth + 1 + deepest BLAS of the scene */
  uint32_t last_kernel; /* kernel the most recent rt_render launched (0, 8 or 9; what RT_KERNEL_VARIANT_AUTO resolved to) */
  uint64_t device_bytes; /* scene bytes resident in HBM */
} rt_stats;

/* replaces Renderer::init's device/pipeline set-up, src/Renderer.cpp:215-410 */
int rt_create(int device_id, rt_ctx **out_ctx);
/* replaces Renderer::shutdown, src/Renderer.cpp:412-460 */
int rt_destroy(rt_ctx *ctx);

/* replaces the staging of the 11 UniformData arrays, src/Renderer.cpp:790-853
 * (add_blas), :867-868 (add_blas_instance), :1007-1011 (build_tlas) and
 * src/Material.cpp:196-201.  Copies; the caller may free its arrays after. */
int rt_upload_scene(rt_ctx *ctx, const rt_scene_desc *scene);
/* replaces the per-frame instance/TLAS restaging after
 * set_blas_instance_transform, src/Renderer.cpp:901-916, 994-1015. */
int rt_update_instances(rt_ctx *ctx, const rt_blas_instance *instances,
                        uint32_t instance_count, const rt_tlas_node *tlas_nodes,
                        uint32_t tlas_node_count);

/* replaces `device->get_current_cmd_buffer()`, src/Renderer.cpp:563: the reference records its dispatch into the
 * caller's per-frame command buffer; here the caller may hand over the HIP stream (a hipStream_t, e.g. a torch
 * stream's handle) that every later launch, clear and copy of this context is queued on, so that its own work —
 * a collective that reads the image — can be ordered against the render with stream events instead of host waits.
 * NULL reverts to the context's own stream.  Waits for the work already queued on the old stream. */
int rt_set_stream(rt_ctx *ctx, void *hip_stream);

/* replaces the uniform-buffer memcpy, src/Renderer.cpp:532-561 */
int rt_set_camera(rt_ctx *ctx, const rt_camera_uniforms *camera);

/* replaces create_output_image / resize, src/Renderer.cpp:462-486, 654-684.
 * Allocates (or re-uses) a W*H rgba32f accumulation image and zeroes it. */
int rt_resize(rt_ctx *ctx, uint32_t width, uint32_t height);
/* Renders into caller-owned device memory (W*H*16 bytes, e.g. a tensor that a
 * collective will read) instead of the internal image; NULL reverts. */
int rt_bind_accum(rt_ctx *ctx, void *device_rgba32f, uint32_t width,
                  uint32_t height);
/* replaces vkCmdClearColorImage on frame 0, src/Renderer.cpp:584-606 */
int rt_clear_accum(rt_ctx *ctx);

/* replaces push constants + vkCmdDispatch, src/Renderer.cpp:619-648:
 * one launch of the path-tracing megakernel (compute_main). */
int rt_render(rt_ctx *ctx, const rt_render_params *params);
/* replaces `frame_count` iterations of the reference's frame loop with an unchanged scene and camera —
 * PathTracer::run calls Renderer::render once per displayed frame (src/PathTracer.cpp:396-411), each a dispatch of
 * sqrt_spp^2 (= 1, Renderer.cpp:21, 625) samples per pixel re-seeded by frame_index (RayTracing.slang:71-72) and folded
 * into the running mean (:176-179; ++frame_index, Renderer.cpp:651).  Frames params->frame_index + k are accumulated
 * as params->accum_count + k, k = 0 .. frame_count - 1: the image is that of frame_count successive rt_render calls,
 * bit for bit.  Handing the frames over together lets the library run them side by side: (pixel, frame) pairs are
 * independent also under the reference's sequential Wang RNG, so batches of at least 2e6 of them may go through the
 * wavefront pipeline (kernel 9) as independent paths; within one frame of the reference's RNG only the pixels are independent. */
int rt_render_frames(rt_ctx *ctx, const rt_render_params *params, uint32_t frame_count);

/* Test hook: intersect_tlas (TLAS.slang:14-67) over an explicit ray batch.
 * Host-pointer form copies in and out and waits; device form is async. */
int rt_trace_closest(rt_ctx *ctx, const rt_ray *rays, uint64_t n, rt_hit *hits,
                     uint32_t aabb_mode);
int rt_trace_closest_device(rt_ctx *ctx, const void *device_rays, uint64_t n,
                            void *device_hits, uint32_t aabb_mode);

/* The reference has no read-back (the image is only sampled by its blit
 * shader, src/PathTracer.cpp:491); these serve the CLI, tests and bench. */
int rt_read_accum(rt_ctx *ctx, float *host_rgba32f);
int rt_accum_device_ptr(rt_ctx *ctx, void **out_device_ptr);
int rt_sync(rt_ctx *ctx);
int rt_get_stats(rt_ctx *ctx, rt_stats *out);
int rt_reset_stats(rt_ctx *ctx);
/* Low byte: 0 = reference-structured kernel (one lane per pixel, nested sample / bounce / traversal loops);
 * 8 = persistent wave64s with path regeneration, a resumable per-lane traversal, vote scheduling and joined
 * pair / leaf steps; 9 = sample-parallel wavefront pipeline (rt_wavefront.hip: traversal and shading as separate
 * phases over path slots in device memory, a resolve kernel): its units of work are the samples of a pixel under RT_RNG_PCG
 * and the FRAMES of a pixel (rt_render_frames) under the reference's sequential Wang RNG, whose samples within a frame
 * run one after the other in a slot.  In its default, persistent mode (rt_debug_wavefront_persistent) a kernel-9
 * launch is asynchronous like the others; in the rounds mode rt_render blocks the host for its length.  Higher bits are schedule knobs (rt_kernels.hip,
 * rt_launch_render).  All kernels give the same results bit for bit.  RT_KERNEL_VARIANT_AUTO (the state of a new
 * context) lets rt_render choose: kernel 8, except kernel 0 for a tiny scene rendered with at most 4 samples and
 * depth 3 per launch (the reference's live configuration), and except launches of at least 2e6 independent units of
 * work — pixels x frames x samples under RT_RNG_PCG (1920x1080 from one sample per pixel on), pixels x frames under
 * RT_RNG_WANG: which of kernels 8 and 9 is faster there depends on the scene and the shape, so the first such
 * launches of a (scene, launch shape) run kernel 9 twice (the first, on freshly allocated buffers, only warms up; the
 * second is timed) and kernel 8 once (timed; rt_render waits for the three), and the following ones the faster of the
 * two.  If a later launch of the chosen kernel takes more than 1.3x longer or shorter than its trial did — the camera
 * turned, another tenant arrived — the two timed trials are repeated.  The choice belongs to the uploaded scene and the launch shape:
 * rt_update_instances keeps it (an animated scene is not re-measured every frame; after 256 updates it is), another
 * rt_upload_scene, launch shape, wavefront budget or tuning restarts it.  If kernel 9's device memory cannot be
 * allocated, the automatic choice settles on kernel 8 instead of failing, and when kernel 8 wins the trial kernel 9's
 * buffers are released.  rt_stats.last_kernel reports the choice.  The kernels
 * that round 1 measured and dropped (numbers 1-7, 10) are in experiments/r01_variants/, outside the library. */
#define RT_KERNEL_VARIANT_AUTO 0xFFFFFFFFu
int rt_set_kernel_variant(rt_ctx *ctx, uint32_t variant);

/* Sizing of kernel 9's HBM state (no counterpart in the reference): at most `paths_in_flight` path slots of 88 B
 * (default 2^26; up to 2^30 — the persistent mode uses a window of 512 slots per resident trace wave, 3.1 Mi slots =
 * 277 MB on an MI355X, fewer waves if the budget is smaller, and falls back to the rounds below 512; the rounds mode
 * uses a sixth of the launch's paths, at least 2^20, at most this: 64 Mi slots = 5.9 GB)
 * and `result_bytes` for the per-path sample radiances of one sample chunk (16 B per path, default 40 GiB,
 * allocated as far as a launch needs it: a 1920x1080 launch of 1024 spp in one chunk; larger launches run in chunks of
 * whole samples, each with its own thinning rounds).
 * Takes effect at the next launch. */
int rt_set_wavefront_budget(rt_ctx *ctx, uint32_t paths_in_flight, uint64_t result_bytes);

/* Test and measurement aids for kernel 9's round scheduling (no counterpart in the reference).
 * _tuning: a trace wave that is out of fresh slots gives up its rays once `suspend_lanes` or fewer are left (their
 * traversal state is saved; they resume next round; 0 = never; default 32); the trace pass walks the list of live
 * slots once fewer than `list_percent` % of the slots hold a path (0 = never; default 75); `share` = slots a wave
 * takes per atomic (multiple of 64, default 256); once `finish_paths` paths or fewer are alive one kernel sees them
 * through to their ends (0 = never; default 768 per CU).  0xFFFFFFFF = default.  The image does not depend on any of them.
 * _counts: since rt_reset_stats — trace + shade rounds, rounds whose trace pass walked the list, rays suspended, paths handed to the
 * finishing kernel.
 * _compaction: the shade pass keeps the paths of its 256 slots sorted — paths that go on first, oldest first, the new
 * paths' camera rays behind them, dead slots last — so that a trace wave holds coherent rays (default on; 0 = every
 * path stays in the slot of its lane, the round-2 layout).  The image does not depend on it either.
 * _record_cache: the trace kernel copies the first 320 pair records — the breadth-first top of the scene, which every
 * ray walks through — into LDS once per workgroup (512 lanes) and lanes at such a record read it there instead of
 * gathering it through the L1 (default on; 0 = every record through the L1, workgroups of 256 lanes, the shape of
 * rounds 2 and 3).  The image does not depend on it.
 * _persistent: kernel 9 without rounds (default on) — one launch of the trace kernel in which every wave owns a window
 * of 512 path slots and fills it with new paths, traces it and shades it in a loop until the launch's paths are used
 * up; the host reads nothing back, so rt_render returns at once as it does for kernels 0 and 8.  0 = the rounds of
 * rounds 2 and 3: a trace and a shade kernel per round over all slots, the host reading the live-path count between
 * them (rt_render then blocks until the image is complete); _tuning's list_percent, share and finish_paths and the
 * counts of _counts belong to that mode.  The image does not depend on it.
 * _bands: how the persistent mode deals the launch's paths to the GPU's eight XCDs.  0 = interleaved, every XCD works
 * on the same tiles at the same time; 1 = each XCD (each class of workgroups that share one) works through its own
 * eighth of the frame, so that the eight L2s hold eight different parts of the scene; 2 = the default: bands for
 * launches of at least 1e8 paths over a scene of more than 32 MB of records, where they gain about 1 % (they cost small
 * launches up to 30 % and scenes that fit the L2s 4 %).  The reference's RNG at more than one sample per frame is
 * always interleaved.  The image does not depend on it. */
int rt_debug_wavefront_tuning(rt_ctx *ctx, uint32_t suspend_lanes, uint32_t list_percent, uint32_t share,
                              uint32_t finish_paths);
int rt_debug_wavefront_compaction(rt_ctx *ctx, uint32_t enable);
int rt_debug_wavefront_record_cache(rt_ctx *ctx, uint32_t enable);
int rt_debug_wavefront_persistent(rt_ctx *ctx, uint32_t enable);
int rt_debug_wavefront_bands(rt_ctx *ctx, uint32_t mode);
int rt_debug_wavefront_counts(rt_ctx *ctx, uint64_t *rounds, uint64_t *list_rounds, uint64_t *suspended_rays,
                              uint64_t *finished_paths);

/* Measurement aid (no counterpart in the reference): while enabled, every dispatch of a kernel-9 launch is bracketed by
 * HIP events on the launch stream and the launch waits for them at its end.  Returns (either pointer may be NULL) the
 * times summed since rt_reset_stats — out_ms5 = k_wf_trace, k_wf_shade, k_wf_finish, k_wf_resolve, first fill, in ms — and
 * the number of launches they cover, then sets the mode.  bench.py's per-kernel roofline durations come from here. */
int rt_debug_wavefront_timing(rt_ctx *ctx, int enable, double *out_ms5, uint64_t *out_launches);

/* Test hook (no counterpart in the reference): the next `count` device allocations for kernel 9's buffers fail as if
 * the device were out of memory (0 = none). */
int rt_debug_fail_wavefront_allocs(rt_ctx *ctx, uint32_t count);

/* Measurement aid (no counterpart in the reference): every lane of `blocks`
 * 256-thread workgroups walks the uploaded node records for `steps` dependent,
 * divergent gathers of `loads` x 16 bytes: the memory-path ceiling of the
 * traversal kernels.  Returns kernel time and the number of steps walked. */
int rt_debug_probe_walk(rt_ctx *ctx, uint32_t blocks, uint32_t steps, uint32_t loads, double *out_ms,
                        uint64_t *out_steps);

/* Test hook (no counterpart in the reference): the exact fast division the slab
 * test uses for operands of moderate magnitude (rt_device_inl.h, slab_pair_fast)
 * beside the IEEE n/d, element by element, for host arrays of `count` floats. */
int rt_debug_fast_div(rt_ctx *ctx, const float *n, const float *d, uint64_t count, float *q_fast,
                      float *q_ieee);

/* replaces the clustering loop of TLAS::build, src/TLAS.cpp:38-58 with find_best_match :61-80 (called from
 * Renderer::build_tlas, src/Renderer.cpp:994-1015, whenever an instance is added, moved or removed).
 * `leaves` are the n leaf nodes of TLAS.cpp:15-36 (world boxes, blas_instance_idx, left_right = 0) in list order;
 * `out_nodes` receives 2n nodes laid out as the reference lays them out: root copy at 0, leaves at 1..n, merged
 * nodes behind them; `*out_node_count` = 2n - 1 (the value TLAS::build returns).  The loop is sequential, so it
 * runs in one 512-lane workgroup with the live list in registers and LDS; n is limited to RT_TLAS_GPU_MAX_LEAVES.  The result
 * is bit-equal to the host builder (rth_build_tlas).  `kernel_ms` (optional) receives the kernel time. */
#define RT_TLAS_GPU_MAX_LEAVES 4096u
int rt_build_tlas_cluster(rt_ctx *ctx, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                          uint32_t *out_node_count, float *kernel_ms);

/* No counterpart in the reference.  rt_build_tlas_cluster for callers that cannot allocate device memory — a render
 * loop that rebuilds the TLAS every frame under a no-allocation rule, serving loops, harnesses that forbid hipMalloc:
 * the same kernel and the same results, with the staged leaves, the output nodes and the node count held in a static
 * device array of the library (384 KiB + 256 B per GPU, resident until the process ends) instead of per-call hipMalloc
 * / hipFree, and no events (no kernel time is reported).  Calls nothing that allocates.  It is NOT capturable into a
 * HIP graph: it copies from and to the caller's pageable host arrays and waits for the stream before it returns.
 * n > RT_TLAS_GPU_MAX_LEAVES is refused with RT_ERR_INVALID_ARGUMENT.  Every context on a GPU shares that GPU's array:
 * calls that overlap on one GPU race (calls of one context are ordered by its stream and the entry point waits before
 * it returns).  The array's per-device address is looked up by rt_create (hipGetSymbolAddress); contexts created
 * concurrently from several threads may repeat that lookup — harmless, it yields the same address. */
int rt_build_tlas_cluster_noalloc(rt_ctx *ctx, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                                  uint32_t *out_node_count);

const char *rt_last_error_string(void);
/* number of HIP devices visible (0 on a CPU-only machine; never fails) */
int rt_device_count(void);

/* ---- several GPUs of one node (no counterpart in the reference, which is single-GPU) ----
 * One rt_ctx and one host thread per rank, the scene replicated on every GPU.  A frame is cut into groups of 8 rows
 * dealt round-robin: rank r of n renders groups r, r + n, ... (rt_render_params.row_begin = 8 r, row_group_stride = n;
 * rt_mgpu_partition is that bookkeeping, callable without a GPU).  After every rt_mgpu_render_frames the row groups are
 * exchanged ONCE — a direct ncclAllGather of the packed groups over RCCL (xGMI), each rank packing and unpacking on its
 * own stream behind its launch — so every rank's image is complete and equals the one-GPU image bit for bit (no
 * floating-point re-association).  RCCL is dlopen'ed on first use.  exchange_mode: AUTO = RCCL when every rank has a
 * GPU of its own and RCCL loads, else peer copies between the ranks' images (hipMemcpyPeerAsync, host barriers: the
 * rehearsal of n ranks on fewer GPUs); RCCL = refuse to fall back; COPY = peer copies.
 * Calls are made from ONE controlling thread; the per-rank work runs on the driver's own threads.  Failures return a
 * negative rt_status and leave the message in rt_last_error_string() of the calling thread. */
typedef struct rt_mgpu rt_mgpu;
typedef struct rt_mgpu_share {
  uint32_t groups_total;     /* 8-row groups of the image */
  uint32_t groups;           /* groups this rank renders */
  uint32_t groups_per_rank;  /* ceil(groups_total / n): the all-gather's equal share (tail = padding) */
  uint32_t rows;             /* image rows this rank renders */
  uint32_t row_begin, row_end, row_group_stride; /* the rank's rt_render_params fields */
} rt_mgpu_share;
enum { RT_MGPU_EXCHANGE_AUTO = 0, RT_MGPU_EXCHANGE_RCCL = 1, RT_MGPU_EXCHANGE_COPY = 2 };
int rt_mgpu_partition(uint32_t image_height, uint32_t ranks, uint32_t rank, rt_mgpu_share *out);
int rt_mgpu_create(const int *device_ids, uint32_t ranks, uint32_t exchange_mode, rt_mgpu **out);
int rt_mgpu_destroy(rt_mgpu *m);
uint32_t rt_mgpu_ranks(rt_mgpu *m);
int rt_mgpu_uses_rccl(rt_mgpu *m);
rt_ctx *rt_mgpu_context(rt_mgpu *m, uint32_t rank); /* e.g. for rt_set_wavefront_budget; not for launches */
int rt_mgpu_upload_scene(rt_mgpu *m, const rt_scene_desc *scene);             /* rt_upload_scene on every rank */
int rt_mgpu_update_instances(rt_mgpu *m, const rt_blas_instance *instances, uint32_t instance_count,
                             const rt_tlas_node *tlas_nodes, uint32_t tlas_node_count);
int rt_mgpu_set_camera(rt_mgpu *m, const rt_camera_uniforms *camera);
int rt_mgpu_set_kernel_variant(rt_mgpu *m, uint32_t variant);
int rt_mgpu_resize(rt_mgpu *m, uint32_t width, uint32_t height);
int rt_mgpu_clear_accum(rt_mgpu *m);
/* rt_render_frames on every rank's share (params' row fields are set per rank), then the exchange; returns when every
 * rank holds the complete image */
int rt_mgpu_render_frames(rt_mgpu *m, const rt_render_params *params, uint32_t frame_count);
int rt_mgpu_read_accum(rt_mgpu *m, uint32_t rank, float *host_rgba32f);
int rt_mgpu_get_stats(rt_mgpu *m, uint32_t rank, rt_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
