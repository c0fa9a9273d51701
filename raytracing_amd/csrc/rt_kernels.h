// rt_kernels.h — kernel argument block and launch entry points shared by
// rt_kernels.hip (device code) and rt_hip_api.cpp (the C ABI).
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_types.h"
#include "rt_device_layout.h"

// Rows of one launch.  A launch renders the rows y in [row_begin, row_end) that lie in the 8-row groups
// row_begin/8 + k * row_group_stride (k = 0, 1, ...): stride 1 is a contiguous band, stride N with row_begin = 8 * r
// is rank r's share of an image interleaved over N GPUs.  Kernels enumerate "virtual" rows 0 .. rt_virtual_rows - 1.
static inline __host__ __device__ uint32_t rt_row_stride(const rt_render_params &pc) {
  return pc.row_group_stride ? pc.row_group_stride : 1u;
}
static inline __host__ __device__ uint32_t rt_virtual_rows(const rt_render_params &pc) {
  const uint32_t span = pc.row_end - pc.row_begin, step = 8u * rt_row_stride(pc);
  const uint32_t full = span / step, rest = span - full * step;
  return full * 8u + (rest < 8u ? rest : 8u);
}
static inline __host__ __device__ uint32_t rt_row(const rt_render_params &pc, uint32_t vrow) {
  return pc.row_begin + (vrow >> 3) * 8u * rt_row_stride(pc) + (vrow & 7u);
}

// every box coordinate is 0 or of magnitude in [2^-66, 2^59]: the precondition of the
// exact fast division of slab_pair_fast (rt_device_inl.h)
#define RT_SCENE_BOXES_MODERATE 1u

// Device pointers of the flattened scene; passed to kernels by value.
struct rt_dev_scene {
  const rt_dev_pair *pairs;          // TLAS + BLAS interior records
  const rt_dev_tri *tris;            // leaf-ordered triangles
  const rt_dev_inst *inst_trace;     // per instance: inv_transform columns + BLAS root
  const rt_dev_inst_shade *inst_shade; // per instance: matrices + resolved material, read once per shaded hit
  const rt_triangle_shading *shading; // reference layout, indexed by triangle id
  const uint8_t *texels;              // RGBA8 texels of the non-constant Lambert textures
  uint32_t root_ref; // TLAS root reference (RT_REF_NONE for an empty scene)
  uint32_t cache_pairs; // pairs[0, cache_pairs) are the breadth-first top of the scene
  uint32_t flags;       // RT_SCENE_*
  uint32_t pad;
};

hipError_t rt_launch_trace(const rt_dev_scene &sc, const rt_ray *d_rays, uint64_t n, rt_hit *d_hits,
                           uint32_t aabb_mode, unsigned long long *d_counter, hipStream_t stream);
hipError_t rt_launch_render(uint32_t variant, const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                            const rt_render_params &pc, float *d_accum,
                            unsigned long long *d_counter, uint32_t *d_queue, uint32_t regen_idle,
                            uint32_t cus, hipStream_t stream);

hipError_t rt_launch_probe_walk(const rt_dev_scene &sc, uint32_t blocks, uint32_t steps, uint32_t loads,
                                unsigned long long *d_out, hipStream_t stream);

// rt_tlas_build.hip: TLAS::build's clustering loop (src/TLAS.cpp:38-58) in one workgroup; out has 2n nodes
hipError_t rt_launch_tlas_cluster(const rt_tlas_node *d_leaves, uint32_t n, rt_tlas_node *d_out,
                                  uint32_t *d_out_count, hipStream_t stream);

// the no-allocation variant: per-device scratch in a file-scope __device__ array of rt_tlas_build.hip
hipError_t rt_tlas_scratch_init(int device); // once per device, with that device current (rt_create)
hipError_t rt_run_tlas_cluster_noalloc(int device, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                                       uint32_t *out_count, hipStream_t stream);

hipError_t rt_launch_debug_fast_div(const float *d_n, const float *d_d, uint64_t count, float *d_fast,
                                    float *d_ieee, hipStream_t stream);

#endif
