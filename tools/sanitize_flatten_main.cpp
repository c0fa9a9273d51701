// Fuzz driver for the scene validation + flattening of librt_hip.so (rt_flatten.cpp), built with ASan/UBSan by
// tools/sanitize_host.sh: valid scenes whose indices, counts and offsets are overwritten at random must be
// rejected (or accepted) without touching memory out of bounds.  CPU only.
#include <stdio.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "../raytracing_amd/csrc/host/renderer.h"
#include "../raytracing_amd/csrc/host/scene_file.h"
#include "../raytracing_amd/csrc/rt_flatten.h"

int main(int argc, char **argv) {
  int accepted = 0, rejected = 0;
  std::mt19937 rng(7);
  for (int i = 1; i < argc; ++i) {
    rta::SceneFile sf;
    std::string err;
    rta::Renderer r;
    r.init(-1, 8, 8);
    if (!rta::load_scene_file(argv[i], sf, err) || rta::instantiate_scene(sf, r, err) < 0) {
      fprintf(stderr, "cannot load %s: %s\n", argv[i], err.c_str());
      return 1;
    }
    const rt_scene_desc base = r.scene();
    for (int round = 0; round < 300; ++round) {
      rt_scene_desc d = base;
      std::vector<rt_bvh_node> nodes(base.bvh_nodes, base.bvh_nodes + base.bvh_node_count);
      std::vector<rt_blas> blas(base.blas, base.blas + base.blas_count);
      std::vector<rt_blas_instance> inst(base.instances, base.instances + base.instance_count);
      std::vector<rt_tlas_node> tlas(base.tlas_nodes, base.tlas_nodes + base.tlas_node_count);
      std::vector<uint32_t> ids(base.tri_ids, base.tri_ids + base.tri_capacity);
      std::vector<rt_lambert> lam(base.lamberts, base.lamberts + base.lambert_count);
      std::vector<rt_texture_desc> tex(base.textures, base.textures + base.texture_count);
      auto value = [&]() -> uint32_t {
        switch (rng() % 5) {
        case 0: return 0;
        case 1: return 0xFFFFFFFFu;
        case 2: return rng();
        case 3: return rng() % 100000;
        default: return 0x80000000u | (rng() % 1000);
        }
      };
      const int edits = 1 + (int)(rng() % 3);
      for (int e = 0; e < edits && round > 0; ++e) {
        switch (rng() % 12) {
        case 0: if (!nodes.empty()) nodes[rng() % nodes.size()].local_left_first = value(); break;
        case 1: if (!nodes.empty()) nodes[rng() % nodes.size()].tri_count = value(); break;
        case 2: if (!blas.empty()) blas[rng() % blas.size()].bvh_nodes_offset = value(); break;
        case 3: if (!blas.empty()) blas[rng() % blas.size()].nodes_count = value(); break;
        case 4: if (!inst.empty()) inst[rng() % inst.size()].blas_id = value(); break;
        case 5: if (!inst.empty()) inst[rng() % inst.size()].material.index = value(); break;
        case 6: if (!inst.empty()) inst[rng() % inst.size()].material.type = value(); break;
        case 7: if (!tlas.empty()) tlas[rng() % tlas.size()].left_right = value(); break;
        case 8: if (!tlas.empty()) tlas[rng() % tlas.size()].blas_instance_idx = value(); break;
        case 9: if (!ids.empty()) ids[rng() % ids.size()] = value(); break;
        case 10: if (!lam.empty()) lam[rng() % lam.size()].index = value(); break;
        default: if (!tex.empty()) { tex[rng() % tex.size()].offset = value(); tex[rng() % tex.size()].width = value(); } break;
        }
      }
      d.bvh_nodes = nodes.data(), d.blas = blas.data(), d.instances = inst.data(), d.tlas_nodes = tlas.data();
      d.tri_ids = ids.data(), d.lamberts = lam.data(), d.textures = tex.data();
      rt_flat_scene f;
      bool ok = rt_flatten_blas(&d, f, err) &&
                rt_flatten_tlas(d.instances, d.instance_count, d.tlas_nodes, d.tlas_node_count, d.blas_count, f, err) &&
                rt_validate_materials(&d, d.instances, d.instance_count, f.inst_live, err);
      if (ok) { // whatever was accepted must be safe to traverse: every reference in range
        auto check = [&](uint32_t ref, size_t pairs, bool tlas) {
          if (ref == RT_REF_NONE || ref == RT_REF_SENTINEL)
            return true;
          if (!(ref & RT_REF_LEAF))
            return (size_t)ref < pairs;
          if (ref & RT_REF_INST)
            return tlas && (size_t)(ref & 0x3FFFFFFFu) < f.insts.size() && f.inst_live[ref & 0x3FFFFFFFu];
          return !tlas && (size_t)(ref & 0x7FFFFFFFu) < f.tris.size();
        };
        bool safe = check(f.root_ref, f.tlas_pairs.size(), true);
        for (const rt_dev_pair &p : f.blas_pairs)
          safe = safe && check(p.ref0, f.blas_pairs.size(), false) && check(p.ref1, f.blas_pairs.size(), false);
        for (const rt_dev_pair &p : f.tlas_pairs)
          safe = safe && check(p.ref0, f.tlas_pairs.size(), true) && check(p.ref1, f.tlas_pairs.size(), true);
        for (size_t k = 0; k < f.insts.size(); ++k)
          if (f.inst_live[k])
            safe = safe && check(f.insts[k].root_ref, f.blas_pairs.size(), false);
        if (!safe) {
          fprintf(stderr, "UNSAFE scene accepted (round %d)\n", round);
          return 2;
        }
      }
      (ok ? accepted : rejected)++;
    }
  }
  printf("flatten fuzz: accepted %d rejected %d\n", accepted, rejected);
  return accepted > 0 ? 0 : 1;
}
