// rt_wavefront.h — buffers and launcher of the wavefront pipeline (rt_wavefront.hip).
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_kernels.h"

// Path state between rounds, structure-of-arrays over linear pixel index
// p = y * width + x (116 B per pixel in HBM).
struct rt_wf_buffers {
  float4 *ray_o;      // origin.xyz
  float4 *ray_d;      // direction.xyz
  float4 *hit_a;      // t, u, v, triangle id (bits); id == RT_INVALID_ID: miss
  uint32_t *hit_inst; // instance id
  float4 *st_a;       // attenuation.xyz, sample_radiance.x
  float4 *st_b;       // sample_radiance.yz, radiance.xy
  float4 *st_c;       // radiance.z, RNG state, sample << 8 | depth, -
  uint32_t *list[2];  // live pixels of the round (ping-pong)
  uint32_t *count;    // [3] list lengths, rotating
  uint32_t *head;     // [2] queue heads of the extend kernel
};

size_t rt_wf_bytes(uint32_t width, uint32_t height);
rt_wf_buffers rt_wf_carve(void *mem, uint32_t width, uint32_t height);
hipError_t rt_launch_render_wavefront(const rt_dev_scene &sc, const rt_camera_uniforms &cam,
                                      const rt_render_params &pc, float *d_accum,
                                      unsigned long long *d_counter, const rt_wf_buffers &b,
                                      uint32_t pair_min, uint32_t refill_min, uint32_t occupancy_knob,
                                      uint32_t cus, hipStream_t stream);

#endif
