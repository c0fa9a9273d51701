/* rt_types.h — POD input layouts of the render path (plain C, shared by the
 * C-ABI library, the host renderer and the test oracle).
 *
 * Every struct mirrors, byte for byte, the layout the reference hands to its
 * GPU kernel, so a caller holding reference-side arrays can pass them as-is:
 *
 *   rt_triangle_geom     <- hlx::TriangleGeom     src/Triangle.hpp:9-18      (48 B)
 *   rt_triangle_shading  <- hlx::TriangleShading  src/Triangle.hpp:20-35     (80 B)
 *   rt_bvh_node          <- hlx::BVHNode          src/BVHNode.hpp:9-15       (32 B)
 *   rt_blas              <- hlx::BLAS             src/BVHNode.hpp:17-39      (16 B)
 *   rt_blas_instance     <- hlx::BLASInstance     src/BVHNode.hpp:49-59      (144 B)
 *   rt_tlas_node         <- hlx::TLASNode         src/TLAS.hpp:5-12          (32 B)
 *   rt_material_handle   <- MaterialHandle        src/Material.hpp:26-29     (8 B)
 *   rt_lambert/metal/dielectric/emissive <- src/Material.hpp:10-24           (4/16/4/16 B)
 *   rt_camera_uniforms   <- first 64 B of UniformData  src/Renderer.cpp:30-34
 *
 * Matrices are column-major (glm / `VkShaderCompilation.cpp:197`): m[c*4+r].
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_triangle_geom {
  float v0[4]; /* w = 1 (unused by the kernels) */
  float v1[4];
  float v2[4];
} rt_triangle_geom;

typedef struct rt_triangle_shading {
  float n0[4];
  float n1[4];
  float n2[4];
  float uv0[2];
  float uv1[2];
  float uv2[2];
  float padding[2];
} rt_triangle_shading;

typedef struct rt_bvh_node {
  float aabb_min[3];
  uint32_t local_left_first; /* interior: BLAS-local index of the left child
                                (right = left + 1); leaf: first index into
                                tri_ids (global) */
  float aabb_max[3];
  uint32_t tri_count; /* 0 => interior */
} rt_bvh_node;

typedef struct rt_blas {
  uint32_t bvh_nodes_offset; /* root index in the global node array */
  uint32_t nodes_count;
  uint32_t tri_count;
  uint32_t padding;
} rt_blas;

enum rt_material_type {
  RT_MATERIAL_LAMBERT = 0,
  RT_MATERIAL_METAL = 1,
  RT_MATERIAL_DIELECTRIC = 2,
  RT_MATERIAL_EMISSIVE = 3,
  RT_MATERIAL_NONE = 4
};

typedef struct rt_material_handle {
  uint32_t index;
  uint32_t type; /* enum rt_material_type */
} rt_material_handle;

typedef struct rt_blas_instance {
  float transform[16];     /* column-major object->world */
  float inv_transform[16]; /* column-major world->object */
  rt_material_handle material;
  uint32_t blas_id;
  uint32_t padding;
} rt_blas_instance;

typedef struct rt_tlas_node {
  float aabb_min[3];
  uint32_t left_right; /* child ids, 2 x 16 bit; 0 => leaf */
  float aabb_max[3];
  uint32_t blas_instance_idx;
} rt_tlas_node;

typedef struct rt_lambert {
  uint32_t index; /* albedo texture index */
} rt_lambert;

typedef struct rt_metal {
  float albedo_fuzz[4]; /* rgb, fuzz */
} rt_metal;

typedef struct rt_dielectric {
  float refraction_index;
} rt_dielectric;

typedef struct rt_emissive {
  float intensity[4];
} rt_emissive;

/* One RGBA8-UNORM albedo texture (nearest / repeat sampling,
 * src/Renderer.cpp:347-361, src/Material.cpp:163-198). */
typedef struct rt_texture_desc {
  uint32_t offset; /* byte offset into the texel blob */
  uint32_t width;
  uint32_t height;
  uint32_t padding;
} rt_texture_desc;

/* Camera block of UniformData (src/Renderer.cpp:30-34, 532-537) plus the
 * defocus disk basis (Random.slang:65-70; zero => pinhole like the reference). */
typedef struct rt_camera_uniforms {
  float pixel00_loc[4];
  float pixel_delta_u[4];
  float pixel_delta_v[4];
  float camera_center[4];
  float defocus_disk_u[4];
  float defocus_disk_v[4];
} rt_camera_uniforms;

/* A ray for the closest-hit test hook. */
typedef struct rt_ray {
  float origin[3];
  float t_min;
  float direction[3];
  float t_max;
} rt_ray;

/* Closest-hit result: (t, triangle id, instance id) are the parity triple. */
typedef struct rt_hit {
  float t;         /* t_max of the ray when nothing was hit */
  uint32_t tri_id; /* global triangle index; 0xFFFFFFFF on a miss */
  uint32_t inst_id;
  float u;
  float v;
  uint32_t pad[3];
} rt_hit;

#define RT_INVALID_ID 0xFFFFFFFFu

/* The scene exactly as the reference publishes it to its kernel: the 11
 * arrays of UniformData (src/Renderer.cpp:35-45) plus the albedo textures that
 * the reference binds as a descriptor array (src/Material.cpp:158-220).
 * All pointers are host memory owned by the caller; counts are array lengths
 * (ids index these arrays directly, holes are allowed: a BLAS slot with
 * nodes_count == 0 is unused). */
typedef struct rt_scene_desc {
  const rt_triangle_geom *tri_geom;       /* [tri_capacity] */
  const rt_triangle_shading *tri_shading; /* [tri_capacity] */
  const uint32_t *tri_ids;                /* [tri_capacity] */
  uint32_t tri_capacity;
  uint32_t bvh_node_count;
  const rt_bvh_node *bvh_nodes; /* [bvh_node_count] */
  const rt_blas *blas;          /* [blas_count] */
  uint32_t blas_count;
  uint32_t instance_count;
  const rt_blas_instance *instances; /* [instance_count] */
  const rt_tlas_node *tlas_nodes;    /* [tlas_node_count], root at 0 */
  uint32_t tlas_node_count;
  uint32_t lambert_count;
  const rt_lambert *lamberts;
  const rt_metal *metals;
  const rt_dielectric *dielectrics;
  const rt_emissive *emissives;
  uint32_t metal_count;
  uint32_t dielectric_count;
  uint32_t emissive_count;
  uint32_t texture_count;
  const rt_texture_desc *textures;
  const uint8_t *texels; /* RGBA8, [texel_bytes] */
  uint64_t texel_bytes;
} rt_scene_desc;

/* Per-launch parameters: PushConstant of the reference (src/Renderer.cpp:48-62,
 * 619-631) with its hard-coded constants turned into fields. */
typedef struct rt_render_params {
  uint32_t image_width;
  uint32_t image_height;
  uint32_t frame_index; /* RNG seed term, RayTracing.slang:71-72 */
  uint32_t accum_count; /* frames already in the running mean, :176-179
                           (the reference uses frame_index for both) */
  uint32_t sqrt_spp;    /* samples per launch = sqrt_spp^2, :76-77 */
  uint32_t max_depth;   /* reference: 3, RayTracing.slang:45 */
  uint32_t rng_mode;    /* enum rt_rng_mode */
  uint32_t aabb_mode;   /* enum rt_aabb_mode */
  uint32_t row_begin;   /* render rows [row_begin, row_end); 0,0 => all */
  uint32_t row_end;
  uint32_t row_group_stride; /* 0 or 1: every row of the range; N: only the 8-row groups row_begin/8 + k*N
                                (multi-GPU interleaving: rank r sets row_begin = 8 r, stride = N) */
  uint32_t flags; /* RT_RENDER_*; 0 = the documented reading of the reference */
} rt_render_params;

/* rt_render_params.flags.  The reference cannot be run here, so where its source admits two readings both are
 * selectable (and tested against the oracle both ways).
 * RT_RENDER_DIELECTRIC_ALWAYS_DRAWS: Material.slang:91 `cannot_refract || reflectance(..) > rand(seed)` — default
 * (flag clear): `||` short-circuits as Slang / HLSL-2021 specify for scalars, so a ray that cannot refract draws no
 * random number; flag set: both operands are evaluated (as a compiler that flattens the condition would), so every
 * dielectric hit advances the RNG once. */
#define RT_RENDER_DIELECTRIC_ALWAYS_DRAWS 1u
/* RT_RENDER_SKY_WHITE_HORIZON: the sky of the REVISION THAT RENDERED the reference's Screenshot3.png.  The shader in
 * the tree blends float3(0.7) -> (0.5, 0.7, 1.0) (RayTracing.slang:162); the screenshot's sky pixels — (224, 237, 255)
 * at the top row, (235, 244, 255) at the bottom, through the old PNG writer's 256 * clamp(sqrt(c), 0, 0.999) — are those
 * of float3(1.0) -> (0.5, 0.7, 1.0) to the level (derivation in DESIGN.md §2).  Flag set: the horizon colour is
 * float3(1.0), the arithmetic is otherwise the same.  Exists so that a render can be compared with that screenshot,
 * the only output the reference holds (tests/test_reference_screenshot.py). */
#define RT_RENDER_SKY_WHITE_HORIZON 2u
#define RT_RENDER_FLAGS_ALL (RT_RENDER_DIELECTRIC_ALWAYS_DRAWS | RT_RENDER_SKY_WHITE_HORIZON)

/* RNG selection (RayTracing.slang:71-72 is RT_RNG_WANG). */
enum rt_rng_mode {
  RT_RNG_WANG = 0, /* sequential Wang hash, state per pixel per frame */
  RT_RNG_PCG = 1   /* counter-based PCG keyed on (pixel, sample, bounce, dim) */
};

/* Slab test arithmetic (BVHNode.slang:8-26 divides: RT_AABB_DIV). */
enum rt_aabb_mode {
  RT_AABB_DIV = 0, /* (b - o) / d       — reference shader */
  RT_AABB_RCP = 1  /* (b - o) * (1 / d) — the disabled CPU path, AABB.hpp:29-48 */
};

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
static_assert(sizeof(rt_triangle_geom) == 48, "TriangleGeom");
static_assert(sizeof(rt_triangle_shading) == 80, "TriangleShading");
static_assert(sizeof(rt_bvh_node) == 32, "BVHNode");
static_assert(sizeof(rt_blas) == 16, "BLAS");
static_assert(sizeof(rt_blas_instance) == 144, "BLASInstance");
static_assert(sizeof(rt_tlas_node) == 32, "TLASNode");
static_assert(sizeof(rt_material_handle) == 8, "MaterialHandle");
static_assert(sizeof(rt_metal) == 16, "Metal");
static_assert(sizeof(rt_emissive) == 16, "Emissive");
static_assert(sizeof(rt_ray) == 32, "rt_ray");
static_assert(sizeof(rt_hit) == 32, "rt_hit");
#endif

#endif /* RT_TYPES_H */
