// gltf_file.h — glTF 2.0 import (.gltf with external / base64 buffers, .glb) following the reference's
// load_gltf_scene, src/SceneGraph.cpp:168-407, and SceneGraph::update_transforms, :544-566.
#ifndef RTA_GLTF_FILE_H
#define RTA_GLTF_FILE_H

#include <stdint.h>

#include <string>
#include <vector>

#include "renderer.h"

namespace rta {

struct GltfImportInfo {
  uint32_t nodes = 0, meshes = 0, primitives = 0, materials = 0;
  uint64_t triangles = 0;
  std::vector<uint32_t> instance_ids; // one per primitive, in import order
};

// Adds the default scene of the asset to the renderer: one Lambert per glTF material, one BLAS and one instance per
// triangle primitive; instance transform = parent_global * (node matrices down the hierarchy).  `parent_global` is
// the world matrix of the scene-graph node the asset is attached to (identity for the scene root).
// Returns the number of triangles added, or -1 with a message in err.
int64_t import_gltf(const std::string &path, Renderer &renderer, const mat4 &parent_global, std::string &err,
                    GltfImportInfo *info = nullptr);

} // namespace rta
#endif
