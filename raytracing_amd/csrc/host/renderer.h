// renderer.h — host renderer: the reference's `hlx::Renderer`
// (src/Renderer.hpp:18-113) with the Vulkan device swapped for librt_hip's C
// ABI.  Method names, argument meaning, handle reuse (FreeIndexPool) and the
// error behaviour (warn + ignore on bad ids) follow the reference; the
// hard-coded constants of the reference (samples_per_pixel, max_depth) are
// public fields with the reference's values as defaults.
#ifndef RTA_RENDERER_H
#define RTA_RENDERER_H

#include <stdint.h>

#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/rt_hip.h"
#include "../../../include/rt_types.h"
#include "hmath.h"

namespace rta {

// hlx::Camera, src/Camera.hpp:20-25 (+ the dead loader's defocus pair, Scene.cpp:171-172)
struct Camera {
  vec3 position{0.f, 0.f, 1.f};
  vec3 look_at{0.f, 0.f, 0.f};
  vec3 v_up{0.f, 1.f, 0.f};
  vec3 u, v, w;
  float fov = 90.f; // vertical, degrees
  bool changed = false;
  float defocus_angle = 0.f; // degrees; 0 = pinhole (the reference's kernel)
  float focus_dist = 1.f;
};

// FreeIndexPool, src/Core/FreeIndexPool.cpp: a stack of free indices.
class FreeIndexPool {
public:
  void init(uint32_t capacity);
  uint32_t obtain_new(); // UINT32_MAX when exhausted
  void release(uint32_t index);
  uint32_t size = 0;
  uint32_t capacity = 0;

private:
  std::vector<uint32_t> free_indices_;
  uint32_t head_ = 0;
};

// First-fit range allocator over element indices; stands in for the TLSF pools
// that hand out tri-id and BVH-node ranges (src/Renderer.cpp:298, 310).
class RangeAllocator {
public:
  void init(uint32_t capacity);
  bool allocate(uint32_t count, uint32_t *offset); // false when no range fits
  void release(uint32_t offset, uint32_t count);
  uint32_t high_water() const { return high_water_; }

private:
  std::vector<std::pair<uint32_t, uint32_t>> free_; // sorted (offset, count)
  uint32_t high_water_ = 0;
};

class Renderer {
public:
  // device_id < 0: host-only (scene assembly and builders work, render() fails).
  int init(int device_id, uint32_t output_image_width, uint32_t output_image_height);
  void shutdown();
  int resize(uint32_t output_image_width, uint32_t output_image_height);
  int render(Camera &camera);
  int render_frames(Camera &camera, uint32_t frames); // `frames` successive render() calls in one device call

  rt_material_handle add_lambert_material(vec3 albedo);
  rt_material_handle add_lambert_material(int32_t width, int32_t height, const uint8_t *pixels);
  // image file (PNG, binary PNM) decoded to RGBA8 like stbi_load(path, .., 4), Renderer.cpp:705-716; a file that
  // cannot be decoded warns and returns a NONE handle (the reference asserts)
  rt_material_handle add_lambert_material(const std::string &file_path);
  rt_material_handle add_metal_material(vec3 albedo, float fuzz);
  rt_material_handle add_dielectric_material(float refractive_index);
  rt_material_handle add_emissive_material(vec3 intensity);
  void remove_material(const rt_material_handle &material_handle);

  uint32_t add_blas(const float *positions, const float *normals, const float *uvs,
                    const uint32_t *indices, size_t index_count);
  uint32_t add_blas_instance(uint32_t blas_index, const mat4 &transform, rt_material_handle material);
  void set_blas_instance_transform(uint32_t blas_instance_id, const mat4 &transform);
  void remove_blas(uint32_t blas_id);
  void remove_blas_instance(uint32_t blas_instance_id);

  // ---- beyond the reference's interface (it has no read-back or options) ----
  int read_image(float *rgba32f);        // waits for the GPU
  int wait();                            // drain the stream
  const rt_scene_desc &scene();          // the arrays as published to the kernel (builds the TLAS)
  rt_camera_uniforms camera_uniforms(Camera &camera) const; // Renderer.cpp:506-537
  rt_ctx *context() { return ctx_; }
  const std::string &last_error() const { return last_error_; }

public:
  uint32_t total_triangle_count = 0;
  uint32_t frame_index = 0;
  uint32_t sphere_blas_index = UINT32_MAX;
  uint32_t cube_blas_index = UINT32_MAX;
  uint32_t plane_blas_index = UINT32_MAX;
  rt_material_handle default_material{0, RT_MATERIAL_LAMBERT};
  // reference constants made configurable
  uint32_t samples_per_pixel = 3; // Renderer.cpp:21 (sqrt truncates: 1 sample per launch)
  uint32_t max_depth = 3;         // RayTracing.slang:45
  uint32_t rng_mode = RT_RNG_WANG;
  uint32_t aabb_mode = RT_AABB_DIV;
  uint32_t width = 0, height = 0;
  // frames whose index is congruent to frame_offset modulo frame_stride are
  // rendered by this renderer (multi-GPU sample sharding); defaults: all frames.
  uint32_t frame_stride = 1, frame_offset = 0;
  // rows [row_begin, row_end) are rendered (multi-GPU row sharding); 0, 0 = the whole image
  uint32_t row_begin = 0, row_end = 0;
  // > 1: only every row_group_stride-th 8-row group of that range (multi-GPU row interleaving)
  uint32_t row_group_stride = 0;
  uint32_t render_flags = 0; // rt_render_params.flags
  // TLAS rebuilds (Renderer.cpp:994-1015) with at least this many instances run their clustering loop on the GPU
  // (rt_build_tlas_cluster, up to RT_TLAS_GPU_MAX_LEAVES); 0 = always on the host.  Same nodes either way.
  uint32_t tlas_gpu_min_instances = 1536; // measured break-even against one host core (tools/bench_tlas.py)
  bool last_tlas_on_gpu = false;
  float last_tlas_kernel_ms = 0.f;

  static constexpr size_t MAX_TRIANGLE_COUNT = 4'000'000; // Renderer.cpp:24-26
  static constexpr size_t MAX_MATERIAL_COUNT = 1'000;
  static constexpr size_t MAX_BLAS_COUNT = 4'000;

private:
  void build_tlas();
  int sync_device();
  bool warn(const std::string &msg);
  struct MaterialPool {
    FreeIndexPool pool;
    std::vector<uint32_t> reference_counts;
    std::set<uint32_t> live;
    void init(uint32_t n) {
      pool.init(n);
      reference_counts.assign(n, 0);
      live.clear();
    }
    bool remove(uint32_t index); // MaterialManager::remove_material, Material.cpp:35-45
  };
  struct BlasAllocation {
    uint32_t tri_offset, tri_count, node_offset, node_count;
  };

  rt_ctx *ctx_ = nullptr;
  std::string last_error_;
  // CPU-side arrays uploaded to the GPU (Renderer.hpp:91-107)
  std::vector<rt_triangle_geom> tri_geom_;
  std::vector<rt_triangle_shading> tri_shading_;
  std::vector<float> centroids_;
  std::vector<uint32_t> tri_ids_;
  std::vector<rt_bvh_node> bvh_nodes_;
  std::vector<rt_blas> blases_;
  std::vector<rt_blas_instance> blas_instances_;
  std::vector<rt_tlas_node> tlas_nodes_;
  uint32_t tlas_node_count_ = 0;
  std::vector<uint32_t> blas_use_count_;
  RangeAllocator tri_id_allocator_, bvh_nodes_allocator_;
  std::unordered_map<uint32_t, BlasAllocation> blas_allocations_;
  FreeIndexPool blases_index_pool_, blas_inst_index_pool_;
  std::set<uint32_t> blas_instance_ids_; // ascending: the TLAS leaf order
  MaterialPool lambert_, metal_, dielectric_, emissive_;
  std::vector<rt_lambert> lambert_mats_;
  std::vector<rt_metal> metal_mats_;
  std::vector<rt_dielectric> dielectric_mats_;
  std::vector<rt_emissive> emissive_mats_;
  std::vector<rt_texture_desc> textures_; // one per lambert slot
  std::vector<std::vector<uint8_t>> texture_pixels_;
  std::vector<uint8_t> texel_blob_;
  rt_scene_desc desc_{};
  bool rebuild_tlas_ = false;
  bool geometry_dirty_ = true;  // BLAS / materials changed: full upload
  bool instances_dirty_ = true; // instances / TLAS changed
};

} // namespace rta
#endif
