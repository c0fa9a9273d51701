/* rt_host.h — C API of librt_host.so: the host renderer (`rta::Renderer`,
 * raytracing_amd/csrc/host/renderer.h), i.e. the reference's `hlx::Renderer`
 * interface (src/Renderer.hpp:18-113) over librt_hip.  One function per
 * reference method, same argument meaning; matrices are 16 floats column-major.
 * Functions return 0 / a negative rt_status unless stated otherwise;
 * rth_last_error() describes the last failure of that renderer.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_hip.h"
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_renderer rth_renderer;
typedef struct rth_mesh rth_mesh;

typedef struct rth_camera { /* hlx::Camera, src/Camera.hpp:20-25 */
  float position[3];
  float look_at[3];
  float v_up[3];
  float fov;           /* vertical, degrees */
  float defocus_angle; /* degrees; 0 = pinhole (reference) */
  float focus_dist;
  int32_t changed; /* non-zero restarts accumulation, Renderer.cpp:491-494 */
} rth_camera;

enum rth_option {
  RTH_SAMPLES_PER_PIXEL = 0, /* Renderer.cpp:21, default 3 (-> 1 per launch) */
  RTH_MAX_DEPTH = 1,         /* RayTracing.slang:45, default 3 */
  RTH_RNG_MODE = 2,          /* enum rt_rng_mode */
  RTH_AABB_MODE = 3,         /* enum rt_aabb_mode */
  RTH_FRAME_STRIDE = 4,      /* multi-GPU sample sharding: RNG frame = offset + k*stride */
  RTH_FRAME_OFFSET = 5,
  RTH_KERNEL_VARIANT = 6,
  RTH_ROW_BEGIN = 7, /* multi-GPU row sharding: render rows [begin, end) only; 0,0 = all */
  RTH_ROW_END = 8,
  RTH_TLAS_GPU_MIN_INSTANCES = 9, /* TLAS clustering on the GPU from this many instances on; 0 = host only */
  RTH_ROW_GROUP_STRIDE = 10, /* rt_render_params.row_group_stride: N = every N-th 8-row group from RTH_ROW_BEGIN on */
  RTH_RENDER_FLAGS = 11      /* rt_render_params.flags (RT_RENDER_*) */
};

/* Renderer::init / shutdown / resize / render, src/Renderer.cpp:215-652.
 * device_id < 0 creates a host-only renderer (scene assembly works, render fails). */
int rth_create(int device_id, uint32_t width, uint32_t height, rth_renderer **out);
void rth_destroy(rth_renderer *r);
int rth_resize(rth_renderer *r, uint32_t width, uint32_t height);
int rth_render(rth_renderer *r, rth_camera *camera);
/* `frames` iterations of the reference's frame loop (PathTracer.cpp:396-411) in one call: same image and frame index as
 * `frames` successive rth_render calls (rt_render_frames) */
int rth_render_frames(rth_renderer *r, rth_camera *camera, uint32_t frames);
int rth_set_option(rth_renderer *r, int option, uint32_t value);
uint32_t rth_get_frame_index(rth_renderer *r);
void rth_reset_accumulation(rth_renderer *r); /* frame_index = 0 */

/* Renderer::add_*_material / remove_material, src/Renderer.cpp:686-754 */
rt_material_handle rth_add_lambert_material(rth_renderer *r, const float *albedo_rgb);
rt_material_handle rth_add_lambert_texture(rth_renderer *r, int32_t width, int32_t height,
                                           const uint8_t *rgba8);
/* add_lambert_material(std::string_view file_path), src/Renderer.cpp:705-716: PNG or binary PNM, decoded to RGBA8
 * as stbi_load(.., 4) returns it; type RT_MATERIAL_NONE if the file cannot be decoded (see rth_last_error) */
rt_material_handle rth_add_lambert_image(rth_renderer *r, const char *file_path);
/* the decoder alone: fills *width / *height and, if rgba8 != NULL and capacity >= w*h*4, the pixels; returns 0 or
 * -1 with a message in err */
/* the command line's PNG writer (8-bit RGB, rows top to bottom); 0 or -1 */
int rth_write_png_rgb8(const char *file_path, uint32_t width, uint32_t height, const uint8_t *rgb8);
int rth_decode_image(const char *file_path, uint32_t *width, uint32_t *height, uint8_t *rgba8, uint64_t capacity,
                     char *err, size_t err_len);
rt_material_handle rth_add_metal_material(rth_renderer *r, const float *albedo_rgb, float fuzz);
rt_material_handle rth_add_dielectric_material(rth_renderer *r, float refractive_index);
rt_material_handle rth_add_emissive_material(rth_renderer *r, const float *intensity_rgb);
void rth_remove_material(rth_renderer *r, rt_material_handle m);
rt_material_handle rth_default_material(rth_renderer *r);

/* Renderer::add_blas ... remove_blas_instance, src/Renderer.cpp:756-946.
 * Ids are UINT32_MAX on failure. */
uint32_t rth_add_blas(rth_renderer *r, const float *positions, const float *normals,
                      const float *uvs, const uint32_t *indices, uint64_t index_count);
uint32_t rth_add_blas_mesh(rth_renderer *r, const rth_mesh *mesh);
uint32_t rth_add_blas_instance(rth_renderer *r, uint32_t blas_index, const float *transform16,
                               rt_material_handle material);
void rth_set_blas_instance_transform(rth_renderer *r, uint32_t instance_id,
                                     const float *transform16);
void rth_remove_blas(rth_renderer *r, uint32_t blas_id);
void rth_remove_blas_instance(rth_renderer *r, uint32_t instance_id);
/* which: 0 plane, 1 cube, 2 sphere (Renderer.hpp:65-67) */
uint32_t rth_primitive_blas(rth_renderer *r, int which);
uint32_t rth_total_triangle_count(rth_renderer *r);

/* The arrays as published to the kernel (rebuilds the TLAS when stale).  The
 * pointers stay valid until the next call that changes the scene. */
int rth_scene_desc(rth_renderer *r, rt_scene_desc *out);
int rth_camera_uniforms(rth_renderer *r, rth_camera *camera, rt_camera_uniforms *out);
int rth_read_image(rth_renderer *r, float *rgba32f);
int rth_wait(rth_renderer *r);
rt_ctx *rth_context(rth_renderer *r); /* NULL for a host-only renderer */
const char *rth_last_error(rth_renderer *r);

/* Transform::get_mat4 (src/Transform.hpp:15-18) and glm helpers for callers
 * that have no glm: out16 = translate(pos) * toMat4(angleAxis(angle, axis)) * scale(s) */
void rth_trs(const float *position, float angle_rad, const float *axis, const float *scale,
             float *out16);

/* Meshes (raytracing_amd/csrc/host/meshes.h). which: 0 plane, 1 cube, 2 sphere. */
rth_mesh *rth_make_primitive(int which);
rth_mesh *rth_make_dragon(uint32_t u_seg, uint32_t v_seg, uint32_t seed);
/* a second dragon-class mesh with non-uniform tessellation: `levels` rounds of 1:4 subdivision of `base` (any mesh,
 * e.g. a group of teapot.obj), vertices displaced along their normals by seeded noise (csrc/host/meshes.h) */
rth_mesh *rth_make_subdivided(const rth_mesh *base, uint32_t levels, float amplitude, uint32_t seed);
/* OBJ split by material; returns the number of groups (<0 on error, message in
 * err/err_len) and fills up to max_groups meshes, their Kd colours (3 floats
 * each) and has-material flags. */
int rth_load_obj(const char *path, rth_mesh **groups, float *group_kd, int32_t *group_has_material,
                 int max_groups, char *err, int err_len);
void rth_mesh_free(rth_mesh *m);
uint32_t rth_mesh_vertex_count(const rth_mesh *m);
uint32_t rth_mesh_index_count(const rth_mesh *m);
const float *rth_mesh_positions(const rth_mesh *m);
const float *rth_mesh_normals(const rth_mesh *m);
const float *rth_mesh_uvs(const rth_mesh *m);
const uint32_t *rth_mesh_indices(const rth_mesh *m);

/* Scene files: the reference's JSON scene description (README.md:53-68,
 * Scenes/TestScene.json; loader semantics of the dead src/Scene.cpp:134-233).
 * Loads every model of the file into `r` (one BLAS + instance per OBJ material
 * group) and returns the camera and the image settings of the file.  Returns the
 * number of triangles added, or a negative rt_status (message: err/err_len). */
typedef struct rth_scene_info {
  rth_camera camera;
  uint32_t screen_width, screen_height;
  uint32_t samples_per_pixel, max_depth;
  uint32_t model_count;
} rth_scene_info;
int64_t rth_load_scene_file(rth_renderer *r, const char *json_path, rth_scene_info *out, char *err,
                            int err_len);

/* Builders on their own (tests compare them with the oracle's). */
uint32_t rth_build_blas(rt_bvh_node *nodes, const rt_triangle_geom *tris, const float *centroids,
                        uint32_t *tri_ids, uint32_t tri_count, uint32_t first);
/* BLAS::refit, src/BVHNode.cpp:163-177: recomputes the boxes of an existing tree bottom-up after its triangles
 * moved (topology and leaf order kept) */
void rth_refit_blas(rt_bvh_node *nodes, uint32_t nodes_count, const rt_triangle_geom *tris, const uint32_t *tri_ids);
uint32_t rth_build_tlas(rt_tlas_node *tlas_nodes, const rt_blas_instance *instances,
                        const uint32_t *instance_ids, uint32_t n, const rt_blas *blas,
                        const rt_bvh_node *bvh_nodes);
/* the two halves of rth_build_tlas: leaf boxes (TLAS.cpp:15-36) and the clustering loop (TLAS.cpp:38-58) over
 * nodes[1..n]; the loop's GPU counterpart is rt_build_tlas_cluster (rt_hip.h) */
void rth_tlas_leaves(rt_tlas_node *leaves, const rt_blas_instance *instances, const uint32_t *instance_ids,
                     uint32_t n, const rt_blas *blas, const rt_bvh_node *bvh_nodes);
uint32_t rth_tlas_cluster(rt_tlas_node *nodes, uint32_t n);
/* glTF 2.0 import (.gltf / .glb) as the reference's load_gltf_scene + update_transforms do it
 * (src/SceneGraph.cpp:168-407, :544-566): Lamberts from the materials, one BLAS + instance per triangle primitive,
 * transforms down the node hierarchy below `parent_global16` (column-major 4x4; NULL = identity).  Returns the number
 * of triangles added or -1 (message in err); counts[4] (optional) = nodes, meshes, primitives, materials. */
int64_t rth_import_gltf(rth_renderer *r, const char *path, const float *parent_global16, uint32_t *counts,
                        char *err, size_t err_len);
/* 1 if the renderer's last TLAS build ran its clustering loop on the GPU; *kernel_ms (optional) = its kernel time */
int rth_last_tlas_build(rth_renderer *r, float *kernel_ms);
void rth_mat4_inverse(const float *m16, float *out16);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
