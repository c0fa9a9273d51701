// rt_wavefront.h — buffers and launcher of the sample-parallel wavefront pipeline (rt_wavefront.hip, kernel 9).
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_kernels.h"

// Path state between rounds, structure-of-arrays over path slots (84 B per slot in HBM; the depth word of ray_d is
// 0xFFFFFFFF for a slot without a path), and the per-path result buffer of one sample chunk.
#define RT_WF_CTRL_WORDS 12288u
struct rt_wf_buffers {
  float4 *ray_o;      // [slots] origin.xyz, path id within the chunk (bits)
  float4 *ray_d;      // [slots] direction.xyz, depth (bits)
  float4 *att;        // [slots] attenuation.xyz
  float4 *rad;        // [slots] sample_radiance.xyz
  float4 *hit;        // [slots] t, triangle id (bits; RT_INVALID_ID = miss), instance id (bits), u
  float *hit_v;       // [slots] v
  uint32_t *seed;     // [slots] Wang RNG: the pixel's sequential state between two bounces (Random.slang:7-14)
  float4 *fsum;       // [slots] Wang RNG, spp > 1: the frame's radiance over the samples done so far, sample index (bits)
  uint32_t *ctrl;     // [RT_WF_CTRL_WORDS] striped counters, see rt_wavefront.hip
  uint32_t *block_pos; // [slots / 256] every shade workgroup's position in its share of the chunk's path ids
  float4 *partial;    // [tile slots] radiance summed over the sample chunks done so far
  float4 *sr;         // [sr_paths] sample radiance of every path of the current chunk, in path-id order
  uint32_t *live;     // [64 stripes][list span] slots with a live path after the last shade pass (lengths in ctrl)
  uint32_t *pool;     // [2][pool_records][64 words] traversal state of suspended rays, written in one round, read in the next
  uint32_t pool_records;
  uint32_t slots;     // capacity: paths in flight
  uint64_t sr_paths;  // capacity of sr
};

// pair records and triangles in one allocation (copies of rt_dev_scene's arrays): the cooperative fetch of the trace
// kernel forms addresses as base + 32-bit offset
struct rt_wf_geom {
  const uint8_t *base;
  uint32_t pairs_off; // byte offset of pair record 0
  uint32_t tris_off;  // byte offset of triangle 0
  uint32_t bytes;     // pair records + triangles (the persistent mode's non-temporal state accesses pay from a scene
                      // larger than the L2s on)
};

uint32_t rt_wf_pool_records(uint32_t cus);
uint64_t rt_wf_persistent_slots(uint32_t cus); // path slots the persistent mode can use: a window per resident trace wave
size_t rt_wf_state_bytes(uint32_t slots, uint64_t tile_slots, uint32_t pool_records);
void rt_wf_carve(void *state_mem, uint32_t slots, uint64_t tile_slots, uint32_t pool_records, rt_wf_buffers &b);

// How the rounds are scheduled; 0xFFFFFFFF in a field = the measured default.
struct rt_wf_tuning {
  uint32_t suspend_lanes = 0xFFFFFFFFu; // a trace wave out of fresh slots gives up its rays once this many or fewer are left
                                        // (they resume next round; 0 = never; default 32)
  uint32_t list_percent = 0xFFFFFFFFu;  // the trace pass walks the live-slot list once fewer than this % of the slots are
                                        // alive (0 = never; default 75)
  uint32_t share = 0xFFFFFFFFu;         // slots a wave takes per atomic (a multiple of 64; default 256)
  uint32_t finish_paths = 0xFFFFFFFFu;  // once this many paths or fewer are alive one kernel sees them through, trace and
                                        // shade in a loop per lane (0 = never; default 768 per CU)
  uint32_t compact = 0xFFFFFFFFu;       // the shade pass moves the paths that go on to the front of their 256-slot window
                                        // and the new paths behind them (0 = every path stays in its lane's slot; default on)
  uint32_t persistent = 0xFFFFFFFFu;    // no rounds: every trace wave owns a window of 512 slots and fills, traces and shades it
                                        // in a loop (0 = rounds of trace and shade kernels with the host between them; default on)
  uint32_t record_cache = 0xFFFFFFFFu;  // the trace kernel keeps the first pair records (the breadth-first top of the scene)
                                        // in LDS, workgroups of 512 lanes (0 = every record through the L1, 256 lanes; default on)
  uint32_t bands = 0xFFFFFFFFu;         // persistent mode: each class of workgroups that share an XCD works through its own
                                        // eighth of the frame (0 = never, 1 = always; default: launches of 1e8 paths or more
                                        // over a scene of more than 32 MB, rt_wavefront.hip wp_draw_group)
};
struct rt_wf_counts { // statistics, summed over launches
  uint64_t rounds = 0;      // trace + shade rounds
  uint64_t list_rounds = 0; // of these, rounds whose trace pass walked the live-slot list
  uint64_t finished_paths = 0; // paths alive when k_wf_finish took over
  // per-kernel GPU time (HIP events on the launch stream around every dispatch), filled only while `timing` is set
  // (rt_debug_wavefront_timing: the events and the extra wait at the end of the launch are a measurement mode)
  bool timing = false;
  double trace_ms = 0, shade_ms = 0, finish_ms = 0, resolve_ms = 0, fill_ms = 0;
  uint64_t timed_launches = 0;
};

// `frame_count` consecutive frames (pc.frame_index + k accumulated as pc.accum_count + k) through the pipeline.
// d_counter: two words, rays traced and rays suspended.  h_ctrl: pinned host memory, RT_WF_CTRL_WORDS uint32.  Blocks
// the host until the image is complete (the host reads the list lengths back between rounds).
// Entries of work (rt_wavefront.hip): frame_count * spp for RT_RNG_PCG, frame_count for RT_RNG_WANG.
uint64_t rt_wf_entries(const rt_render_params &pc, uint32_t frame_count);
hipError_t rt_launch_render_wavefront(const rt_dev_scene &sc, const rt_wf_geom &geom, const rt_camera_uniforms &cam,
                                      const rt_render_params &pc, uint32_t frame_count, float *d_accum, unsigned long long *d_counter,
                                      const rt_wf_buffers &b, uint32_t *h_ctrl, uint32_t knobs, const rt_wf_tuning &tune,
                                      rt_wf_counts *counts, uint32_t cus, hipStream_t stream);

#endif
