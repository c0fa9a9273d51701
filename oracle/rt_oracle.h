/* rt_oracle.h — C API of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the checker: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (raytracing_amd/, librt_hip.so, librt_host.so) never links, loads or
 * calls anything declared here.
 *
 * PARITY: the reference ships no tests, golden vectors or fixtures for this
 * path and its own implementation (Slang GPU code +
 * Windows/Vulkan/glm host code) cannot be built or run here.  The oracle is a
 * line-by-line CPU restatement of the reference sources cited at each function
 * in rt_oracle.cpp.  PINNED against the one output the reference holds — its
 * Screenshot3.png render of Scenes/TestScene.json (tests/golden/screenshot3.npz,
 * tests/test_reference_screenshot.py): scene loading, camera and pixel mapping
 * (silhouettes to the pixel) and Lambert / sky transport in absolute 8-bit
 * levels.  UNPINNED (restatement + hand-derived known answers only,
 * tests/test_oracle_kat.py, tests/test_appendix_c.py): hit ids, RNG sequence,
 * traversal order, metal / dielectric / emissive, instancing.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters of the restated traversal, used for the roofline's
 * algorithmic bytes. */
typedef struct orc_counters {
  uint64_t rays;          /* intersect_tlas calls */
  uint64_t tlas_interior; /* interior TLAS visits (2 child boxes each) */
  uint64_t tlas_leaf;     /* TLAS leaf visits (instance fetch + BLAS entry) */
  uint64_t blas_interior; /* interior BLAS visits (2 child boxes each) */
  uint64_t blas_leaf;     /* BLAS leaf visits */
  uint64_t tris_tested;   /* TriangleGeom::hit calls */
  uint64_t shaded_hits;   /* hits that reached the material switch */
  uint64_t max_stack;     /* deepest stack_ptr seen (TLAS or BLAS) */
} orc_counters;

/* ---- glm restatements (third-party, not in the tree: g-truc/glm, unpinned
 * submodule .gitmodules:7-9; algorithms as published in glm 1.0.x) ---- */
void orc_mat4_identity(float *m);
void orc_mat4_mul(const float *a, const float *b, float *out);
void orc_mat4_inverse(const float *m, float *out);
void orc_mat4_mul_vec4(const float *m, const float *v, float *out);
void orc_mat4_translate(const float *m, const float *v, float *out);
void orc_mat4_scale(const float *m, const float *v, float *out);
void orc_quat_angle_axis(float angle, const float *axis, float *q_wxyz);
void orc_quat_to_mat4(const float *q_wxyz, float *out);
/* Transform::get_mat4, src/Transform.hpp:15-18 */
void orc_trs(const float *pos, const float *q_wxyz, const float *scale,
             float *out);

/* ---- host-side restatements ---- */
/* Renderer::render camera block, src/Renderer.cpp:506-537 */
void orc_camera_uniforms(const float *position, const float *look_at,
                         const float *v_up, float fov_deg, uint32_t width,
                         uint32_t height, rt_camera_uniforms *out);
/* Renderer::add_blas de-indexing, src/Renderer.cpp:773-787 */
void orc_deindex(const float *positions, const float *normals, const float *uvs,
                 const uint32_t *indices, uint32_t index_count,
                 uint32_t tri_id_index, rt_triangle_geom *geom,
                 rt_triangle_shading *shading, float *centroids,
                 uint32_t *tri_ids);
/* BLAS::build, src/BVHNode.cpp:76-161.  geom/centroids/tri_ids are the GLOBAL
 * arrays; nodes is a scratch array of 2*tri_count-1.  Returns nodes_count. */
uint32_t orc_build_blas(rt_bvh_node *nodes, const rt_triangle_geom *geom,
                        const float *centroids, uint32_t *tri_ids,
                        uint32_t tri_count, uint32_t tri_id_offset);
/* BLAS::refit, src/BVHNode.cpp:163-177 */
void orc_refit_blas(rt_bvh_node *nodes, uint32_t nodes_count,
                    const rt_triangle_geom *geom, const uint32_t *tri_ids);
/* TLAS::build, src/TLAS.cpp:6-80.  tlas_nodes has room for 2*n entries.
 * Returns node_count. */
uint32_t orc_build_tlas(rt_tlas_node *tlas_nodes,
                        const rt_blas_instance *instances,
                        const uint32_t *instance_ids, uint32_t n,
                        const rt_blas *blas, const rt_bvh_node *bvh_nodes);

/* ---- shader restatements ---- */
uint32_t orc_wang_hash(uint32_t *seed);       /* Random.slang:7-14 */
float orc_rand(uint32_t *seed);               /* Random.slang:17-19 */
void orc_sincos(float a, float *s, float *c); /* deterministic sin/cos */
/* intersect_tlas, TLAS.slang:14-67, over a ray batch. */
void orc_trace_closest(const rt_scene_desc *scene, const rt_ray *rays,
                       uint64_t n, rt_hit *hits, uint32_t aabb_mode,
                       orc_counters *counters);
/* Brute force: every instance x every triangle, in (tlas-leaf, triangle)
 * index order, closest hit by strict t (validation of the BVH path only). */
void orc_trace_brute(const rt_scene_desc *scene, const rt_ray *rays, uint64_t n,
                     rt_hit *hits);
/* compute_main, RayTracing.slang:62-181, for rows [row_begin,row_end) of an
 * rgba32f image (accum is the full W*H*4 image); nthreads row-parallel. */
void orc_render(const rt_scene_desc *scene, const rt_camera_uniforms *cam,
                const rt_render_params *params, float *accum,
                orc_counters *counters, uint32_t nthreads);
/* Primary camera rays / first-bounce scattered rays of a frame, for hit
 * fixtures: writes up to cap rays, returns the count. bounce=0 camera rays,
 * bounce=k the rays entering the k-th bounce. */
uint64_t orc_gen_rays(const rt_scene_desc *scene, const rt_camera_uniforms *cam,
                      const rt_render_params *params, uint32_t bounce,
                      rt_ray *rays, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
