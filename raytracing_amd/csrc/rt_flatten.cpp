// rt_flatten.cpp — see rt_flatten.h.
#include "rt_flatten.h"

#include <string.h>

namespace {

struct Item {
  uint32_t node;  // index of the child node in its source array
  uint32_t pair;  // pair record that holds this child's box
  uint32_t side;  // 0 / 1
  uint32_t depth; // edges from the root
};

inline void set_box(rt_dev_pair &p, uint32_t side, const float *mn, const float *mx, uint32_t ref) {
  const uint32_t k = side * 2u;
  p.x[k] = mn[0], p.x[k + 1] = mx[0];
  p.y[k] = mn[1], p.y[k + 1] = mx[1];
  p.z[k] = mn[2], p.z[k + 1] = mx[2];
  if (side == 0)
    p.ref0 = ref;
  else
    p.ref1 = ref;
}

std::string at(const char *what, uint64_t i) { return std::string(what) + " " + std::to_string(i); }

// Fills tris[first, first+count) for one leaf.
bool fill_leaf(const rt_scene_desc *sc, uint32_t first, uint32_t count, rt_flat_scene &out,
               std::string &err) {
  if (count == 0 || (uint64_t)first + count > sc->tri_capacity) {
    err = "BVH leaf range outside tri_ids: " + at("first", first) + at(" count", count);
    return false;
  }
  if (first >= 0x7FFFFFFFu) {
    err = "triangle slot index too large for a leaf reference";
    return false;
  }
  for (uint32_t i = 0; i < count; ++i) {
    uint32_t id = sc->tri_ids[first + i];
    if (id >= sc->tri_capacity) {
      err = "tri_ids entry out of range at " + at("slot", first + i);
      return false;
    }
    const rt_triangle_geom &g = sc->tri_geom[id];
    rt_dev_tri &t = out.tris[first + i];
    for (int k = 0; k < 3; ++k) {
      t.v0[k] = g.v0[k];
      t.e1[k] = g.v1[k] - g.v0[k]; // Triangle.slang:7
      t.e2[k] = g.v2[k] - g.v0[k]; // Triangle.slang:8
    }
    t.id = id;
    t.last = (i + 1 == count) ? 1u : 0u;
    t.pad = 0;
  }
  return true;
}

} // namespace

bool rt_flatten_blas(const rt_scene_desc *sc, rt_flat_scene &out, std::string &err) {
  out.blas_pairs.clear();
  out.tris.assign(sc->tri_capacity, rt_dev_tri{});
  out.blas_root_ref.assign(sc->blas_count, RT_REF_NONE);
  out.blas_depth.assign(sc->blas_count, 0);
  std::vector<Item> work;
  for (uint32_t b = 0; b < sc->blas_count; ++b) {
    const rt_blas &bl = sc->blas[b];
    if (bl.nodes_count == 0)
      continue; // free slot
    if ((uint64_t)bl.bvh_nodes_offset + bl.nodes_count > sc->bvh_node_count) {
      err = "BLAS node range outside bvh_nodes: " + at("blas", b);
      return false;
    }
    const rt_bvh_node *nodes = sc->bvh_nodes + bl.bvh_nodes_offset;
    const rt_bvh_node &root = nodes[0];
    if (root.tri_count > 0) { // one-leaf BLAS
      if (!fill_leaf(sc, root.local_left_first, root.tri_count, out, err))
        return false;
      out.blas_root_ref[b] = RT_REF_LEAF | root.local_left_first;
      out.blas_depth[b] = 0;
      continue;
    }
    uint32_t pairs_made = 0, max_depth = 0;
    auto expand = [&](uint32_t node_idx, uint32_t depth, uint32_t &ref_out) -> bool {
      // node_idx is interior: allocate its pair and queue both children
      uint32_t l = nodes[node_idx].local_left_first;
      if ((uint64_t)l + 1 >= bl.nodes_count || l == 0) {
        err = "BLAS child index out of range: " + at("blas", b) + at(" node", node_idx);
        return false;
      }
      if (++pairs_made > bl.nodes_count) {
        err = "BLAS node graph is not a tree: " + at("blas", b);
        return false;
      }
      ref_out = (uint32_t)out.blas_pairs.size();
      out.blas_pairs.emplace_back();
      memset(&out.blas_pairs.back(), 0, sizeof(rt_dev_pair));
      work.push_back(Item{l + 1, ref_out, 1, depth + 1});
      work.push_back(Item{l, ref_out, 0, depth + 1}); // left subtree first
      return true;
    };
    uint32_t root_ref;
    work.clear();
    if (!expand(0, 0, root_ref))
      return false;
    while (!work.empty()) {
      Item it = work.back();
      work.pop_back();
      const rt_bvh_node &n = nodes[it.node];
      uint32_t ref;
      if (n.tri_count > 0) {
        if (!fill_leaf(sc, n.local_left_first, n.tri_count, out, err))
          return false;
        ref = RT_REF_LEAF | n.local_left_first;
        if (it.depth > max_depth)
          max_depth = it.depth;
      } else {
        if (!expand(it.node, it.depth, ref))
          return false;
      }
      set_box(out.blas_pairs[it.pair], it.side, n.aabb_min, n.aabb_max, ref);
    }
    if (out.blas_pairs.size() >= RT_REF_INST) {
      err = "too many interior nodes for a 30-bit reference";
      return false;
    }
    out.blas_root_ref[b] = root_ref;
    out.blas_depth[b] = max_depth;
  }
  return true;
}

bool rt_flatten_tlas(const rt_blas_instance *instances, uint32_t instance_count,
                     const rt_tlas_node *tlas, uint32_t tlas_node_count, uint32_t blas_count,
                     rt_flat_scene &out, std::string &err) {
  out.tlas_pairs.clear();
  out.insts.assign(instance_count, rt_dev_inst{});
  out.inst_live.assign(instance_count, 0);
  out.root_ref = RT_REF_NONE;
  out.tlas_depth = 0;
  out.stack_needed = 0;
  if (instance_count >= RT_REF_INST - 2) {
    err = "too many instances";
    return false;
  }
  for (uint32_t i = 0; i < instance_count; ++i) {
    const float *m = instances[i].inv_transform;
    rt_dev_inst &d = out.insts[i];
    memset(&d, 0, sizeof d);
    memcpy(d.c0, m + 0, 12);
    memcpy(d.c1, m + 4, 12);
    memcpy(d.c2, m + 8, 12);
    memcpy(d.c3, m + 12, 12);
    uint32_t b = instances[i].blas_id;
    d.root_ref = (b < blas_count) ? out.blas_root_ref[b] : RT_REF_NONE;
  }
  if (tlas_node_count == 0)
    return true; // empty scene: every ray misses
  uint32_t max_blas_depth = 0;
  auto leaf_ref = [&](const rt_tlas_node &n, uint32_t &ref) -> bool {
    uint32_t i = n.blas_instance_idx;
    if (i >= instance_count || out.insts[i].root_ref == RT_REF_NONE) {
      err = "TLAS leaf names an invalid instance or an instance of a freed BLAS: " + at("instance", i);
      return false;
    }
    uint32_t d = out.blas_depth[instances[i].blas_id];
    if (d > max_blas_depth)
      max_blas_depth = d;
    ref = RT_REF_LEAF | RT_REF_INST | i;
    out.inst_live[i] = 1;
    return true;
  };
  const rt_tlas_node &root = tlas[0];
  if (root.left_right == 0) {
    if (!leaf_ref(root, out.root_ref))
      return false;
    out.stack_needed = 1 + max_blas_depth;
    return true;
  }
  std::vector<Item> work;
  uint32_t pairs_made = 0, max_depth = 0;
  auto expand = [&](uint32_t node_idx, uint32_t depth, uint32_t &ref_out) -> bool {
    uint32_t lr = tlas[node_idx].left_right;
    uint32_t a = lr & 0xFFFFu, b = lr >> 16; // TLAS.slang:36-37
    if (a >= tlas_node_count || b >= tlas_node_count) {
      err = "TLAS child index out of range at " + at("node", node_idx);
      return false;
    }
    if (++pairs_made > tlas_node_count) {
      err = "TLAS node graph is not a tree";
      return false;
    }
    ref_out = (uint32_t)out.tlas_pairs.size();
    out.tlas_pairs.emplace_back();
    memset(&out.tlas_pairs.back(), 0, sizeof(rt_dev_pair));
    work.push_back(Item{b, ref_out, 1, depth + 1});
    work.push_back(Item{a, ref_out, 0, depth + 1});
    return true;
  };
  if (!expand(0, 0, out.root_ref))
    return false;
  while (!work.empty()) {
    Item it = work.back();
    work.pop_back();
    const rt_tlas_node &n = tlas[it.node];
    uint32_t ref;
    if (n.left_right == 0) {
      if (!leaf_ref(n, ref))
        return false;
      if (it.depth > max_depth)
        max_depth = it.depth;
    } else {
      if (!expand(it.node, it.depth, ref))
        return false;
    }
    set_box(out.tlas_pairs[it.pair], it.side, n.aabb_min, n.aabb_max, ref);
  }
  out.tlas_depth = max_depth;
  out.stack_needed = max_depth + 1 + max_blas_depth;
  return true;
}

bool rt_validate_materials(const rt_scene_desc *sc, const rt_blas_instance *instances,
                           uint32_t instance_count, const std::vector<uint8_t> &inst_live,
                           std::string &err) {
  for (uint32_t i = 0; i < sc->texture_count; ++i) {
    const rt_texture_desc &t = sc->textures[i];
    if (t.width == 0 || t.height == 0 ||
        (uint64_t)t.offset + 4ull * t.width * t.height > sc->texel_bytes || (t.offset & 3u)) {
      err = "texture outside the texel blob: " + at("texture", i);
      return false;
    }
  }
  for (uint32_t i = 0; i < sc->lambert_count; ++i)
    if (sc->lamberts[i].index >= sc->texture_count) {
      err = "lambert material names a missing texture: " + at("lambert", i);
      return false;
    }
  for (uint32_t i = 0; i < instance_count; ++i) {
    if (!inst_live[i])
      continue;
    const rt_material_handle &m = instances[i].material;
    uint32_t limit = m.type == RT_MATERIAL_LAMBERT      ? sc->lambert_count
                     : m.type == RT_MATERIAL_METAL      ? sc->metal_count
                     : m.type == RT_MATERIAL_DIELECTRIC ? sc->dielectric_count
                     : m.type == RT_MATERIAL_EMISSIVE   ? sc->emissive_count
                                                        : 0xFFFFFFFFu; // NONE: never read
    if (m.index >= limit && m.type <= RT_MATERIAL_EMISSIVE) {
      err = "instance material index out of range: " + at("instance", i);
      return false;
    }
  }
  return true;
}
