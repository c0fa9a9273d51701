// rt_wavefront.h — buffers and launcher of the sample-parallel wavefront pipeline (rt_wavefront.hip, kernel 9).
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_kernels.h"

// Path state between rounds, structure-of-arrays over path slots (84 B per slot in HBM; the depth word of ray_d is
// 0xFFFFFFFF for a slot without a path), and the per-path result buffer of one sample chunk.
#define RT_WF_CTRL_WORDS 8192u
struct rt_wf_buffers {
  float4 *ray_o;      // [slots] origin.xyz, path id within the chunk (bits)
  float4 *ray_d;      // [slots] direction.xyz, depth (bits)
  float4 *att;        // [slots] attenuation.xyz
  float4 *rad;        // [slots] sample_radiance.xyz
  float4 *hit;        // [slots] t, triangle id (bits; RT_INVALID_ID = miss), instance id (bits), u
  float *hit_v;       // [slots] v
  uint32_t *ctrl;     // [RT_WF_CTRL_WORDS] striped counters, see rt_wavefront.hip
  uint32_t *block_pos; // [slots / 256] every shade workgroup's position in its share of the chunk's path ids
  float4 *partial;    // [tile slots] radiance summed over the sample chunks done so far
  float4 *sr;         // [sr_paths] sample radiance of every path of the current chunk, in path-id order
  uint32_t slots;     // capacity: paths in flight
  uint64_t sr_paths;  // capacity of sr
};

// pair records and triangles in one allocation (copies of rt_dev_scene's arrays): the cooperative fetch of the trace
// kernel forms addresses as base + 32-bit offset
struct rt_wf_geom {
  const uint8_t *base;
  uint32_t pairs_off; // byte offset of pair record 0
  uint32_t tris_off;  // byte offset of triangle 0
};

size_t rt_wf_state_bytes(uint32_t slots, uint64_t tile_slots);
void rt_wf_carve(void *state_mem, uint32_t slots, uint64_t tile_slots, rt_wf_buffers &b);

// One rt_render launch through the pipeline.  h_ctrl: pinned host memory, RT_WF_CTRL_WORDS uint32.  Blocks the host until the
// image is complete (the host reads the list lengths back between rounds).  pc.rng_mode must be RT_RNG_PCG.
hipError_t rt_launch_render_wavefront(const rt_dev_scene &sc, const rt_wf_geom &geom, const rt_camera_uniforms &cam,
                                      const rt_render_params &pc, float *d_accum, unsigned long long *d_counter,
                                      const rt_wf_buffers &b, uint32_t *h_ctrl, uint32_t knobs, uint32_t cus,
                                      hipStream_t stream);

#endif
