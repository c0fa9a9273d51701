// meshes.h — indexed triangle meshes handed to Renderer::add_blas: the
// reference's three procedural primitives, a procedural dragon-class stand-in
// (the reference's Scenes/dragon/*.obj are missing from its snapshot), and an
// OBJ reader for the shipped teapot / Cornell boxes.
#ifndef RTA_MESHES_H
#define RTA_MESHES_H

#include <stdint.h>

#include <string>
#include <vector>

namespace rta {

struct Mesh {
  std::vector<float> positions; // xyz
  std::vector<float> normals;   // xyz
  std::vector<float> uvs;       // uv
  std::vector<uint32_t> indices;
  uint32_t triangle_count() const { return (uint32_t)(indices.size() / 3); }
};

// generate_sphere / generate_plane / generate_cube, src/Renderer.cpp:64-213
Mesh make_sphere(float radius, uint32_t segments, uint32_t rings);
Mesh make_plane(float width, float depth, uint32_t x_segments, uint32_t z_segments);
Mesh make_cube(float width, float height, float depth);

// Dragon-class stand-in for BASELINE config C3: a (2,3) torus knot swept by a
// tube whose radius and surface carry several octaves of deterministic
// displacement (ridges, scales, spikes), tessellated to 2*u_seg*v_seg triangles
// (defaults 1700 x 256 = 870,400, the size class of the Stanford dragon).
Mesh make_dragon(uint32_t u_seg, uint32_t v_seg, uint32_t seed);

// A second dragon-class mesh, with NON-UNIFORM tessellation (the tube above is a regular grid of equal triangles):
// `levels` rounds of 1:4 midpoint subdivision of an arbitrary base mesh — its triangles keep the size ratios of the
// base, for teapot.obj two orders of magnitude in area — with every vertex then moved along its normal by `amplitude`
// times a four-octave value noise of its position (seeded; a function of position, so shared vertices stay shared and
// the surface stays as closed as the base), normals recomputed area-weighted.  teapot.obj (1024 triangles) at 5 levels
// gives 1,048,576 triangles.
Mesh subdivide_displace(const Mesh &base, uint32_t levels, float amplitude, uint32_t seed);

// One OBJ file split by material: group g holds the faces that `usemtl` g.
struct ObjMaterial {
  std::string name;
  float kd[3] = {0.8f, 0.0f, 0.8f}; // loader default, src/Scene.cpp:182
  std::string map_kd;               // diffuse texture path (relative to the .mtl)
  bool has_kd = false;
};
struct ObjModel {
  std::vector<ObjMaterial> materials;
  std::vector<Mesh> groups;        // one mesh per material actually used
  std::vector<int> group_material; // index into materials, -1 = none
};
// Follows the dead loader src/Scene.cpp:38-132 (tinyobj semantics): fan
// triangulation, negative indices, v flipped to 1-v, vertices welded on
// (position, normal, uv); faces without normals get the geometric normal
// (SceneGraph.cpp:338-352).  Returns false and sets err on I/O or syntax errors.
bool load_obj(const std::string &path, ObjModel &out, std::string &err);

} // namespace rta
#endif
