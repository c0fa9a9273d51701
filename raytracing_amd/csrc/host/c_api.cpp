// c_api.cpp — include/rt_host.h over rta::Renderer.
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../../include/rt_host.h"
#include "accel_build.h"
#include "meshes.h"
#include "gltf_file.h"
#include "image_decode.h"
#include "renderer.h"
#include "scene_file.h"

struct rth_renderer {
  rta::Renderer r;
};
struct rth_mesh {
  rta::Mesh m;
};

namespace {
rta::Camera to_camera(const rth_camera *c) {
  rta::Camera cam;
  cam.position = rta::vec3(c->position), cam.look_at = rta::vec3(c->look_at), cam.v_up = rta::vec3(c->v_up);
  cam.fov = c->fov, cam.defocus_angle = c->defocus_angle, cam.focus_dist = c->focus_dist;
  cam.changed = c->changed != 0;
  return cam;
}
const rt_material_handle kNoMaterial = {0, RT_MATERIAL_NONE};
} // namespace

extern "C" {

int rth_create(int device_id, uint32_t width, uint32_t height, rth_renderer **out) {
  if (!out || width == 0 || height == 0)
    return RT_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  rth_renderer *h = new (std::nothrow) rth_renderer();
  if (!h)
    return RT_ERR_OUT_OF_MEMORY;
  int rc = h->r.init(device_id, width, height);
  if (rc != RT_OK) {
    fprintf(stderr, "[rta] rth_create: %s\n", h->r.last_error().c_str());
    h->r.shutdown();
    delete h;
    return rc;
  }
  *out = h;
  return RT_OK;
}

void rth_destroy(rth_renderer *r) {
  if (!r)
    return;
  r->r.shutdown();
  delete r;
}

int rth_resize(rth_renderer *r, uint32_t w, uint32_t h) {
  return r && w && h ? r->r.resize(w, h) : RT_ERR_INVALID_ARGUMENT;
}

int rth_render(rth_renderer *r, rth_camera *camera) {
  if (!r || !camera)
    return RT_ERR_INVALID_ARGUMENT;
  rta::Camera cam = to_camera(camera);
  int rc = r->r.render(cam);
  camera->changed = cam.changed ? 1 : 0;
  return rc;
}

int rth_render_frames(rth_renderer *r, rth_camera *camera, uint32_t frames) {
  if (!r || !camera)
    return RT_ERR_INVALID_ARGUMENT;
  rta::Camera cam = to_camera(camera);
  int rc = r->r.render_frames(cam, frames);
  camera->changed = cam.changed ? 1 : 0;
  return rc;
}

int rth_set_option(rth_renderer *r, int option, uint32_t value) {
  if (!r)
    return RT_ERR_INVALID_ARGUMENT;
  switch (option) {
  case RTH_SAMPLES_PER_PIXEL:
    if (value == 0)
      return RT_ERR_INVALID_ARGUMENT;
    r->r.samples_per_pixel = value;
    break;
  case RTH_MAX_DEPTH:
    r->r.max_depth = value;
    break;
  case RTH_RNG_MODE:
    if (value > RT_RNG_PCG)
      return RT_ERR_INVALID_ARGUMENT;
    r->r.rng_mode = value;
    break;
  case RTH_AABB_MODE:
    if (value > RT_AABB_RCP)
      return RT_ERR_INVALID_ARGUMENT;
    r->r.aabb_mode = value;
    break;
  case RTH_FRAME_STRIDE:
    if (value == 0)
      return RT_ERR_INVALID_ARGUMENT;
    r->r.frame_stride = value;
    break;
  case RTH_FRAME_OFFSET:
    r->r.frame_offset = value;
    break;
  case RTH_ROW_BEGIN:
    r->r.row_begin = value;
    break;
  case RTH_ROW_END:
    r->r.row_end = value;
    break;
  case RTH_ROW_GROUP_STRIDE:
    r->r.row_group_stride = value;
    break;
  case RTH_RENDER_FLAGS:
    r->r.render_flags = value;
    break;
  case RTH_TLAS_GPU_MIN_INSTANCES:
    r->r.tlas_gpu_min_instances = value;
    return RT_OK; // the nodes are the same either way: the running mean stays valid
  case RTH_KERNEL_VARIANT:
    if (!r->r.context())
      return RT_ERR_NO_DEVICE;
    return rt_set_kernel_variant(r->r.context(), value);
  default:
    return RT_ERR_INVALID_ARGUMENT;
  }
  r->r.frame_index = 0; // any of these invalidates the running mean
  return RT_OK;
}

uint32_t rth_get_frame_index(rth_renderer *r) { return r ? r->r.frame_index : 0; }
void rth_reset_accumulation(rth_renderer *r) {
  if (r)
    r->r.frame_index = 0;
}

rt_material_handle rth_add_lambert_material(rth_renderer *r, const float *rgb) {
  return r && rgb ? r->r.add_lambert_material(rta::vec3(rgb)) : kNoMaterial;
}
rt_material_handle rth_add_lambert_texture(rth_renderer *r, int32_t w, int32_t h, const uint8_t *px) {
  return r ? r->r.add_lambert_material(w, h, px) : kNoMaterial;
}
rt_material_handle rth_add_lambert_image(rth_renderer *r, const char *file_path) {
  return r && file_path ? r->r.add_lambert_material(std::string(file_path)) : kNoMaterial;
}
int rth_write_png_rgb8(const char *file_path, uint32_t width, uint32_t height, const uint8_t *rgb8) {
  return file_path && rta::write_png_rgb8(file_path, width, height, rgb8) ? 0 : -1;
}
int rth_decode_image(const char *file_path, uint32_t *width, uint32_t *height, uint8_t *rgba8, uint64_t capacity,
                     char *err, size_t err_len) {
  rta::Image img;
  std::string e;
  if (!file_path || !width || !height) {
    e = "rth_decode_image: NULL argument";
  } else if (rta::load_image_rgba8(file_path, img, e)) {
    *width = img.width, *height = img.height;
    if (rgba8 && capacity >= img.rgba.size())
      memcpy(rgba8, img.rgba.data(), img.rgba.size());
    return 0;
  }
  if (err && err_len)
    snprintf(err, err_len, "%s", e.c_str());
  return -1;
}
rt_material_handle rth_add_metal_material(rth_renderer *r, const float *rgb, float fuzz) {
  return r && rgb ? r->r.add_metal_material(rta::vec3(rgb), fuzz) : kNoMaterial;
}
rt_material_handle rth_add_dielectric_material(rth_renderer *r, float ri) {
  return r ? r->r.add_dielectric_material(ri) : kNoMaterial;
}
rt_material_handle rth_add_emissive_material(rth_renderer *r, const float *rgb) {
  return r && rgb ? r->r.add_emissive_material(rta::vec3(rgb)) : kNoMaterial;
}
void rth_remove_material(rth_renderer *r, rt_material_handle m) {
  if (r)
    r->r.remove_material(m);
}
rt_material_handle rth_default_material(rth_renderer *r) {
  return r ? r->r.default_material : kNoMaterial;
}

uint32_t rth_add_blas(rth_renderer *r, const float *positions, const float *normals, const float *uvs,
                      const uint32_t *indices, uint64_t index_count) {
  if (!r || !positions || !normals || !uvs || !indices)
    return UINT32_MAX;
  return r->r.add_blas(positions, normals, uvs, indices, (size_t)index_count);
}
uint32_t rth_add_blas_mesh(rth_renderer *r, const rth_mesh *mesh) {
  if (!r || !mesh)
    return UINT32_MAX;
  const rta::Mesh &m = mesh->m;
  return r->r.add_blas(m.positions.data(), m.normals.data(), m.uvs.data(), m.indices.data(),
                       m.indices.size());
}
uint32_t rth_add_blas_instance(rth_renderer *r, uint32_t blas_index, const float *t16,
                               rt_material_handle material) {
  return r && t16 ? r->r.add_blas_instance(blas_index, rta::mat4(t16), material) : UINT32_MAX;
}
void rth_set_blas_instance_transform(rth_renderer *r, uint32_t id, const float *t16) {
  if (r && t16)
    r->r.set_blas_instance_transform(id, rta::mat4(t16));
}
void rth_remove_blas(rth_renderer *r, uint32_t id) {
  if (r)
    r->r.remove_blas(id);
}
void rth_remove_blas_instance(rth_renderer *r, uint32_t id) {
  if (r)
    r->r.remove_blas_instance(id);
}
uint32_t rth_primitive_blas(rth_renderer *r, int which) {
  if (!r)
    return UINT32_MAX;
  return which == 0 ? r->r.plane_blas_index : which == 1 ? r->r.cube_blas_index : r->r.sphere_blas_index;
}
uint32_t rth_total_triangle_count(rth_renderer *r) { return r ? r->r.total_triangle_count : 0; }

int rth_scene_desc(rth_renderer *r, rt_scene_desc *out) {
  if (!r || !out)
    return RT_ERR_INVALID_ARGUMENT;
  *out = r->r.scene();
  return RT_OK;
}
int rth_camera_uniforms(rth_renderer *r, rth_camera *camera, rt_camera_uniforms *out) {
  if (!r || !camera || !out)
    return RT_ERR_INVALID_ARGUMENT;
  rta::Camera cam = to_camera(camera);
  *out = r->r.camera_uniforms(cam);
  return RT_OK;
}
int rth_read_image(rth_renderer *r, float *rgba) {
  return r && rgba ? r->r.read_image(rgba) : RT_ERR_INVALID_ARGUMENT;
}
int rth_wait(rth_renderer *r) { return r ? r->r.wait() : RT_ERR_INVALID_ARGUMENT; }
rt_ctx *rth_context(rth_renderer *r) { return r ? r->r.context() : nullptr; }
const char *rth_last_error(rth_renderer *r) { return r ? r->r.last_error().c_str() : "NULL renderer"; }

void rth_trs(const float *position, float angle, const float *axis, const float *scale, float *out16) {
  rta::quat q = rta::angle_axis(angle, rta::vec3(axis));
  if (angle == 0.f)
    q = rta::quat(); // identity, as a default-constructed Transform
  rta::trs(rta::vec3(position), q, rta::vec3(scale)).store(out16);
}

rth_mesh *rth_make_primitive(int which) {
  rth_mesh *m = new rth_mesh();
  m->m = which == 0 ? rta::make_plane(1.f, 1.f, 1, 1)
                    : which == 1 ? rta::make_cube(1.f, 1.f, 1.f) : rta::make_sphere(0.5f, 64, 32);
  return m;
}
rth_mesh *rth_make_dragon(uint32_t u_seg, uint32_t v_seg, uint32_t seed) {
  rth_mesh *m = new rth_mesh();
  m->m = rta::make_dragon(u_seg, v_seg, seed);
  return m;
}
rth_mesh *rth_make_subdivided(const rth_mesh *base, uint32_t levels, float amplitude, uint32_t seed) {
  if (!base || levels > 10)
    return nullptr;
  rth_mesh *m = new rth_mesh();
  m->m = rta::subdivide_displace(base->m, levels, amplitude, seed);
  return m;
}
int rth_load_obj(const char *path, rth_mesh **groups, float *group_kd, int32_t *group_has_material,
                 int max_groups, char *err, int err_len) {
  rta::ObjModel model;
  std::string e;
  if (!path || !rta::load_obj(path, model, e)) {
    if (err && err_len > 0)
      snprintf(err, (size_t)err_len, "%s", path ? e.c_str() : "NULL path");
    return -1;
  }
  int n = (int)model.groups.size();
  for (int i = 0; i < n && i < max_groups; ++i) {
    if (groups) {
      groups[i] = new rth_mesh();
      groups[i]->m = std::move(model.groups[(size_t)i]);
    }
    int mi = model.group_material[(size_t)i];
    if (group_has_material)
      group_has_material[i] = mi >= 0;
    if (group_kd) {
      rta::ObjMaterial def;
      const rta::ObjMaterial &mat = mi >= 0 ? model.materials[(size_t)mi] : def;
      memcpy(group_kd + 3 * i, mat.kd, 12);
    }
  }
  return n;
}
int64_t rth_load_scene_file(rth_renderer *r, const char *json_path, rth_scene_info *out, char *err,
                            int err_len) {
  auto fail = [&](const std::string &e) -> int64_t {
    if (err && err_len > 0)
      snprintf(err, (size_t)err_len, "%s", e.c_str());
    return RT_ERR_INVALID_ARGUMENT;
  };
  if (!r || !json_path || !out)
    return fail("NULL argument");
  rta::SceneFile scene;
  std::string e;
  if (!rta::load_scene_file(json_path, scene, e))
    return fail(e);
  int64_t tris = rta::instantiate_scene(scene, r->r, e);
  if (tris < 0)
    return fail(e);
  const rta::Camera &c = scene.camera;
  rth_camera cam = {{c.position.x, c.position.y, c.position.z}, {c.look_at.x, c.look_at.y, c.look_at.z},
                    {c.v_up.x, c.v_up.y, c.v_up.z}, c.fov, c.defocus_angle, c.focus_dist, 1};
  out->camera = cam;
  out->screen_width = scene.screen_width, out->screen_height = scene.screen_height;
  out->samples_per_pixel = scene.samples_per_pixel, out->max_depth = scene.max_depth;
  out->model_count = (uint32_t)scene.models.size();
  return tris;
}
void rth_mesh_free(rth_mesh *m) { delete m; }
uint32_t rth_mesh_vertex_count(const rth_mesh *m) { return m ? (uint32_t)(m->m.positions.size() / 3) : 0; }
uint32_t rth_mesh_index_count(const rth_mesh *m) { return m ? (uint32_t)m->m.indices.size() : 0; }
const float *rth_mesh_positions(const rth_mesh *m) { return m ? m->m.positions.data() : nullptr; }
const float *rth_mesh_normals(const rth_mesh *m) { return m ? m->m.normals.data() : nullptr; }
const float *rth_mesh_uvs(const rth_mesh *m) { return m ? m->m.uvs.data() : nullptr; }
const uint32_t *rth_mesh_indices(const rth_mesh *m) { return m ? m->m.indices.data() : nullptr; }

uint32_t rth_build_blas(rt_bvh_node *nodes, const rt_triangle_geom *tris, const float *centroids,
                        uint32_t *tri_ids, uint32_t tri_count, uint32_t first) {
  return rta::build_blas(nodes, tris, centroids, tri_ids, tri_count, first);
}
void rth_refit_blas(rt_bvh_node *nodes, uint32_t nodes_count, const rt_triangle_geom *tris, const uint32_t *tri_ids) {
  rta::refit_blas(nodes, nodes_count, tris, tri_ids);
}
uint32_t rth_build_tlas(rt_tlas_node *tlas_nodes, const rt_blas_instance *instances,
                        const uint32_t *instance_ids, uint32_t n, const rt_blas *blas,
                        const rt_bvh_node *bvh_nodes) {
  return rta::build_tlas(tlas_nodes, instances, instance_ids, n, blas, bvh_nodes);
}
int64_t rth_import_gltf(rth_renderer *r, const char *path, const float *parent_global16, uint32_t *counts,
                        char *err, size_t err_len) {
  std::string e;
  int64_t n = -1;
  if (!r || !path) {
    e = "rth_import_gltf: NULL argument";
  } else {
    rta::GltfImportInfo info;
    n = rta::import_gltf(path, r->r, parent_global16 ? rta::mat4(parent_global16) : rta::mat4(1.f), e, &info);
    if (n >= 0 && counts)
      counts[0] = info.nodes, counts[1] = info.meshes, counts[2] = info.primitives, counts[3] = info.materials;
  }
  if (n < 0 && err && err_len)
    snprintf(err, err_len, "%s", e.c_str());
  return n;
}
void rth_tlas_leaves(rt_tlas_node *leaves, const rt_blas_instance *instances, const uint32_t *instance_ids,
                     uint32_t n, const rt_blas *blas, const rt_bvh_node *bvh_nodes) {
  rta::tlas_leaves(leaves, instances, instance_ids, n, blas, bvh_nodes);
}
uint32_t rth_tlas_cluster(rt_tlas_node *nodes, uint32_t n) { return rta::tlas_cluster(nodes, n); }
int rth_last_tlas_build(rth_renderer *r, float *kernel_ms) {
  if (!r)
    return 0;
  if (kernel_ms)
    *kernel_ms = r->r.last_tlas_on_gpu ? r->r.last_tlas_kernel_ms : 0.f;
  return r->r.last_tlas_on_gpu ? 1 : 0;
}
void rth_mat4_inverse(const float *m16, float *out16) { rta::inverse(rta::mat4(m16)).store(out16); }

} // extern "C"
