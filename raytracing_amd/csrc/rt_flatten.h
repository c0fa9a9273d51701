// rt_flatten.h — host-side conversion of the reference's arrays
// (rt_scene_desc) into the device layout of rt_device_layout.h.  Every index
// is validated here so that no kernel can be handed an out-of-range reference.
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/rt_types.h"
#include "rt_device_layout.h"

struct rt_flat_scene {
  std::vector<rt_dev_pair> blas_pairs; // all BLASes, concatenated
  std::vector<rt_dev_pair> tlas_pairs; // indices are relative; rebased on upload
  std::vector<rt_dev_tri> tris;        // [tri_capacity]
  std::vector<rt_dev_inst> insts;      // [instance_count]
  std::vector<uint8_t> inst_live;      // instance is named by a TLAS leaf
  std::vector<uint32_t> blas_root_ref; // per BLAS slot, RT_REF_NONE when unused
  std::vector<uint32_t> blas_depth;    // leaf depth (= stack entries a traversal may need)
  uint32_t root_ref = RT_REF_NONE;     // TLAS root, pair indices relative to tlas_pairs
  uint32_t tlas_depth = 0;
  uint32_t stack_needed = 0;
};

// Flattens triangles + every live BLAS.  Returns false and sets err on bad input.
bool rt_flatten_blas(const rt_scene_desc *sc, rt_flat_scene &out, std::string &err);
// Flattens instances + TLAS against the BLAS roots already in `out`.
bool rt_flatten_tlas(const rt_blas_instance *instances, uint32_t instance_count,
                     const rt_tlas_node *tlas, uint32_t tlas_node_count, uint32_t blas_count,
                     rt_flat_scene &out, std::string &err);
// Material / texture index validation for the shading path.
bool rt_validate_materials(const rt_scene_desc *sc, const rt_blas_instance *instances,
                           uint32_t instance_count, const std::vector<uint8_t> &inst_live,
                           std::string &err);

#endif
