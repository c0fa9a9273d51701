// renderer.cpp — see renderer.h.
#include "renderer.h"

#include "image_decode.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "accel_build.h"
#include "meshes.h"

namespace rta {

// ---- FreeIndexPool (src/Core/FreeIndexPool.cpp) -----------------------------
void FreeIndexPool::init(uint32_t cap) {
  capacity = cap;
  free_indices_.resize(cap);
  for (uint32_t i = 0; i < cap; ++i)
    free_indices_[i] = i;
  head_ = 0;
  size = 0;
}
uint32_t FreeIndexPool::obtain_new() {
  if (head_ >= capacity)
    return UINT32_MAX;
  ++size;
  return free_indices_[head_++];
}
void FreeIndexPool::release(uint32_t index) {
  if (index >= capacity || head_ == 0)
    return;
  free_indices_[--head_] = index;
  --size;
}

// ---- RangeAllocator ---------------------------------------------------------
void RangeAllocator::init(uint32_t capacity) {
  free_.assign(1, {0u, capacity});
  high_water_ = 0;
}
bool RangeAllocator::allocate(uint32_t count, uint32_t *offset) {
  if (count == 0)
    count = 1;
  for (size_t i = 0; i < free_.size(); ++i) {
    if (free_[i].second >= count) {
      *offset = free_[i].first;
      free_[i].first += count;
      free_[i].second -= count;
      if (free_[i].second == 0)
        free_.erase(free_.begin() + (long)i);
      high_water_ = std::max(high_water_, *offset + count);
      return true;
    }
  }
  return false;
}
void RangeAllocator::release(uint32_t offset, uint32_t count) {
  if (count == 0)
    count = 1;
  auto it = std::lower_bound(free_.begin(), free_.end(), std::make_pair(offset, 0u));
  it = free_.insert(it, {offset, count});
  if (it + 1 != free_.end() && it->first + it->second == (it + 1)->first) {
    it->second += (it + 1)->second;
    free_.erase(it + 1);
  }
  if (it != free_.begin() && (it - 1)->first + (it - 1)->second == it->first) {
    (it - 1)->second += it->second;
    free_.erase(it);
  }
}

bool Renderer::MaterialPool::remove(uint32_t index) {
  if (index >= reference_counts.size())
    return false;
  uint32_t &rc = reference_counts[index];
  rc = std::max<uint32_t>(rc, 1) - 1;
  if (rc == 0 && live.count(index)) {
    pool.release(index);
    live.erase(index);
    return true;
  }
  return false;
}

bool Renderer::warn(const std::string &msg) {
  last_error_ = msg; // the reference logs (HWARN) and carries on
  fprintf(stderr, "[rta warn] %s\n", msg.c_str());
  return false;
}

// ---- init / shutdown (Renderer.cpp:215-460) ---------------------------------
int Renderer::init(int device_id, uint32_t w, uint32_t h) {
  width = w, height = h;
  blas_use_count_.assign(MAX_BLAS_COUNT, 0);
  tri_id_allocator_.init((uint32_t)MAX_TRIANGLE_COUNT);
  bvh_nodes_allocator_.init((uint32_t)(MAX_TRIANGLE_COUNT * 2 - 1));
  lambert_.init(MAX_MATERIAL_COUNT), metal_.init(MAX_MATERIAL_COUNT);
  dielectric_.init(MAX_MATERIAL_COUNT), emissive_.init(MAX_MATERIAL_COUNT);
  blases_index_pool_.init(MAX_BLAS_COUNT);
  blas_inst_index_pool_.init(MAX_BLAS_COUNT);
  if (device_id >= 0) {
    int rc = rt_create(device_id, &ctx_);
    if (rc != RT_OK) {
      last_error_ = rt_last_error_string();
      return rc;
    }
    if ((rc = rt_resize(ctx_, w, h)) != RT_OK) {
      last_error_ = rt_last_error_string();
      return rc;
    }
  }
  // primitives in the reference's creation order: plane 0, cube 1, sphere 2 (:332-334)
  Mesh plane = make_plane(1.f, 1.f, 1, 1);
  plane_blas_index = add_blas(plane.positions.data(), plane.normals.data(), plane.uvs.data(),
                              plane.indices.data(), plane.indices.size());
  blas_use_count_[plane_blas_index] = 1; // can never be deleted by the user (:991)
  Mesh cube = make_cube(1.f, 1.f, 1.f);
  cube_blas_index = add_blas(cube.positions.data(), cube.normals.data(), cube.uvs.data(),
                             cube.indices.data(), cube.indices.size());
  blas_use_count_[cube_blas_index] = 1;
  Mesh sphere = make_sphere(0.5f, 64, 32);
  sphere_blas_index = add_blas(sphere.positions.data(), sphere.normals.data(), sphere.uvs.data(),
                               sphere.indices.data(), sphere.indices.size());
  blas_use_count_[sphere_blas_index] = 1;
  default_material = add_lambert_material(vec3(0.7f)); // :405
  ++lambert_.reference_counts[default_material.index];
  return RT_OK;
}

void Renderer::shutdown() {
  if (ctx_)
    rt_destroy(ctx_);
  ctx_ = nullptr;
  *this = Renderer();
}

int Renderer::resize(uint32_t w, uint32_t h) {
  width = w, height = h;
  frame_index = 0; // :485
  if (!ctx_)
    return RT_OK;
  int rc = rt_resize(ctx_, w, h);
  if (rc != RT_OK)
    last_error_ = rt_last_error_string();
  return rc;
}

// ---- materials (Renderer.cpp:686-754, Material.cpp:158-220) -----------------
rt_material_handle Renderer::add_lambert_material(int32_t w, int32_t h, const uint8_t *pixels) {
  uint32_t index = lambert_.pool.obtain_new();
  if (index == UINT32_MAX || w <= 0 || h <= 0 || !pixels) {
    warn("add_lambert_material: pool exhausted or bad image");
    return {0, RT_MATERIAL_NONE};
  }
  if (lambert_mats_.size() <= index) {
    lambert_mats_.resize(index + 1);
    textures_.resize(index + 1);
    texture_pixels_.resize(index + 1);
  }
  lambert_mats_[index] = {index}; // Material.cpp:194
  texture_pixels_[index].assign(pixels, pixels + (size_t)w * h * 4);
  textures_[index] = {0, (uint32_t)w, (uint32_t)h, 0};
  lambert_.live.insert(index);
  geometry_dirty_ = true;
  return {index, RT_MATERIAL_LAMBERT};
}

rt_material_handle Renderer::add_lambert_material(vec3 albedo) {
  auto clamp01 = [](float v) { return gmin(gmax(v, 0.f), 1.f); }; // glm::clamp
  uint8_t px[4] = {(uint8_t)(clamp01(albedo.x) * 255.f), (uint8_t)(clamp01(albedo.y) * 255.f),
                   (uint8_t)(clamp01(albedo.z) * 255.f), 255}; // :696-699
  return add_lambert_material(1, 1, px);
}

rt_material_handle Renderer::add_lambert_material(const std::string &file_path) {
  Image img;
  std::string err;
  if (!load_image_rgba8(file_path, img, err)) { // HASSERT_MSGS(raw_bdata, "Failed to load image"), :710
    warn("add_lambert_material: failed to load image: " + err);
    return {0, RT_MATERIAL_NONE};
  }
  return add_lambert_material((int32_t)img.width, (int32_t)img.height, img.rgba.data());
}

rt_material_handle Renderer::add_metal_material(vec3 albedo, float fuzz) {
  uint32_t index = metal_.pool.obtain_new();
  if (index == UINT32_MAX) {
    warn("add_metal_material: pool exhausted");
    return {0, RT_MATERIAL_NONE};
  }
  if (metal_mats_.size() <= index)
    metal_mats_.resize(index + 1);
  metal_mats_[index] = {{albedo.x, albedo.y, albedo.z, fuzz}};
  metal_.live.insert(index);
  geometry_dirty_ = true;
  return {index, RT_MATERIAL_METAL};
}

rt_material_handle Renderer::add_dielectric_material(float ri) {
  uint32_t index = dielectric_.pool.obtain_new();
  if (index == UINT32_MAX) {
    warn("add_dielectric_material: pool exhausted");
    return {0, RT_MATERIAL_NONE};
  }
  if (dielectric_mats_.size() <= index)
    dielectric_mats_.resize(index + 1);
  dielectric_mats_[index] = {ri};
  dielectric_.live.insert(index);
  geometry_dirty_ = true;
  return {index, RT_MATERIAL_DIELECTRIC};
}

rt_material_handle Renderer::add_emissive_material(vec3 intensity) {
  uint32_t index = emissive_.pool.obtain_new();
  if (index == UINT32_MAX) {
    warn("add_emissive_material: pool exhausted");
    return {0, RT_MATERIAL_NONE};
  }
  if (emissive_mats_.size() <= index)
    emissive_mats_.resize(index + 1);
  emissive_mats_[index] = {{intensity.x, intensity.y, intensity.z, 1.f}};
  emissive_.live.insert(index);
  geometry_dirty_ = true;
  return {index, RT_MATERIAL_EMISSIVE};
}

void Renderer::remove_material(const rt_material_handle &m) {
  switch (m.type) {
  case RT_MATERIAL_LAMBERT:
    lambert_.remove(m.index);
    break;
  case RT_MATERIAL_METAL:
    metal_.remove(m.index);
    break;
  case RT_MATERIAL_DIELECTRIC:
    dielectric_.remove(m.index);
    break;
  case RT_MATERIAL_EMISSIVE:
    emissive_.remove(m.index);
    break;
  default:
    warn("remove_material: unknown MaterialType");
    break;
  }
}

// ---- geometry (Renderer.cpp:756-946) ----------------------------------------
uint32_t Renderer::add_blas(const float *positions, const float *normals, const float *uvs,
                            const uint32_t *indices, size_t index_count) {
  const uint32_t trig_count = (uint32_t)(index_count / 3);
  uint32_t tri_id_index = 0;
  if (trig_count == 0 || !tri_id_allocator_.allocate(trig_count, &tri_id_index)) {
    warn("add_blas: empty mesh or triangle pool exhausted");
    return UINT32_MAX;
  }
  const size_t need = tri_id_allocator_.high_water();
  if (tri_geom_.size() < need) {
    tri_geom_.resize(need);
    tri_shading_.resize(need);
    centroids_.resize(need * 3);
    tri_ids_.resize(need);
  }
  for (uint32_t k = 0; k < trig_count; ++k) { // :773-787
    const uint32_t ia = indices[3 * k], ib = indices[3 * k + 1], ic = indices[3 * k + 2];
    const vec3 a(positions + 3 * (size_t)ia), b(positions + 3 * (size_t)ib), c(positions + 3 * (size_t)ic);
    rt_triangle_geom &g = tri_geom_[tri_id_index + k];
    a.store(g.v0), b.store(g.v1), c.store(g.v2);
    g.v0[3] = g.v1[3] = g.v2[3] = 1.f;
    rt_triangle_shading &s = tri_shading_[tri_id_index + k];
    memcpy(s.n0, normals + 3 * (size_t)ia, 12), memcpy(s.n1, normals + 3 * (size_t)ib, 12);
    memcpy(s.n2, normals + 3 * (size_t)ic, 12);
    s.n0[3] = s.n1[3] = s.n2[3] = 1.f;
    memcpy(s.uv0, uvs + 2 * (size_t)ia, 8), memcpy(s.uv1, uvs + 2 * (size_t)ib, 8);
    memcpy(s.uv2, uvs + 2 * (size_t)ic, 8);
    s.padding[0] = s.padding[1] = 0.f;
    ((a + b + c) * 0.3333f).store(&centroids_[3 * (size_t)(tri_id_index + k)]); // sic, :783
    tri_ids_[tri_id_index + k] = tri_id_index + k;
  }
  std::vector<rt_bvh_node> scratch((size_t)trig_count * 2 - 1); // :808
  const uint32_t blas_index = blases_index_pool_.obtain_new();
  if (blas_index == UINT32_MAX) {
    tri_id_allocator_.release(tri_id_index, trig_count);
    warn("add_blas: BLAS pool exhausted");
    return UINT32_MAX;
  }
  const uint32_t nodes_count = build_blas(scratch.data(), tri_geom_.data(), centroids_.data(),
                                          tri_ids_.data(), trig_count, tri_id_index);
  uint32_t node_offset = 0;
  if (!bvh_nodes_allocator_.allocate(nodes_count, &node_offset)) {
    blases_index_pool_.release(blas_index);
    tri_id_allocator_.release(tri_id_index, trig_count);
    warn("add_blas: BVH node pool exhausted");
    return UINT32_MAX;
  }
  if (bvh_nodes_.size() < bvh_nodes_allocator_.high_water())
    bvh_nodes_.resize(bvh_nodes_allocator_.high_water());
  memcpy(&bvh_nodes_[node_offset], scratch.data(), sizeof(rt_bvh_node) * nodes_count); // :826
  if (blases_.size() <= blas_index)
    blases_.resize(blas_index + 1, rt_blas{0, 0, 0, 0});
  blases_[blas_index] = {node_offset, nodes_count, trig_count, 0};
  blas_allocations_[blas_index] = {tri_id_index, trig_count, node_offset, nodes_count};
  total_triangle_count += trig_count;
  geometry_dirty_ = true;
  return blas_index;
}

uint32_t Renderer::add_blas_instance(uint32_t blas_index, const mat4 &transform,
                                     rt_material_handle material) {
  if (!blas_allocations_.count(blas_index)) {
    warn("add_blas_instance: invalid blas_index");
    return UINT32_MAX;
  }
  const uint32_t index = blas_inst_index_pool_.obtain_new();
  if (index == UINT32_MAX) {
    warn("add_blas_instance: instance pool exhausted");
    return UINT32_MAX;
  }
  if (blas_instances_.size() <= index)
    blas_instances_.resize(index + 1, rt_blas_instance{});
  rt_blas_instance &inst = blas_instances_[index];
  inst.blas_id = blas_index;
  transform.store(inst.transform);
  inverse(transform).store(inst.inv_transform); // BLASInstance::set_transform, BVHNode.cpp:179-182
  inst.material = material;
  inst.padding = 0;
  rebuild_tlas_ = true;
  blas_instance_ids_.insert(index);
  switch (material.type) { // :873-894
  case RT_MATERIAL_LAMBERT:
    if (material.index < lambert_.reference_counts.size())
      ++lambert_.reference_counts[material.index];
    break;
  case RT_MATERIAL_METAL:
    if (material.index < metal_.reference_counts.size())
      ++metal_.reference_counts[material.index];
    break;
  case RT_MATERIAL_DIELECTRIC:
    if (material.index < dielectric_.reference_counts.size())
      ++dielectric_.reference_counts[material.index];
    break;
  case RT_MATERIAL_EMISSIVE:
    if (material.index < emissive_.reference_counts.size())
      ++emissive_.reference_counts[material.index];
    break;
  default:
    warn("add_blas_instance: unknown MaterialType");
    break;
  }
  blas_use_count_[blas_index]++;
  return index;
}

void Renderer::set_blas_instance_transform(uint32_t id, const mat4 &transform) {
  if (!blas_instance_ids_.count(id)) { // :904-908
    warn("set_blas_instance_transform: trying to update an invalid blas_instance_id");
    return;
  }
  rt_blas_instance &inst = blas_instances_[id];
  transform.store(inst.transform);
  inverse(transform).store(inst.inv_transform);
  rebuild_tlas_ = true;
}

void Renderer::remove_blas(uint32_t blas_id) {
  auto it = blas_allocations_.find(blas_id);
  if (it == blas_allocations_.end()) { // :919-923
    warn("remove_blas: trying to remove a blas_id with no allocation data");
    return;
  }
  --blas_use_count_[blas_id];
  if (blas_use_count_[blas_id] != 0)
    return;
  tri_id_allocator_.release(it->second.tri_offset, it->second.tri_count);
  bvh_nodes_allocator_.release(it->second.node_offset, it->second.node_count);
  total_triangle_count -= it->second.tri_count;
  blases_[blas_id] = {0, 0, 0, 0};
  blases_index_pool_.release(blas_id);
  blas_allocations_.erase(it);
  geometry_dirty_ = true;
}

void Renderer::remove_blas_instance(uint32_t id) {
  if (!blas_instance_ids_.count(id)) {
    warn("remove_blas_instance: invalid blas_instance_id");
    return;
  }
  remove_material(blas_instances_[id].material); // :940
  blas_inst_index_pool_.release(id);
  blas_instance_ids_.erase(id);
  remove_blas(blas_instances_[id].blas_id);
  rebuild_tlas_ = true;
}

// ---- TLAS + publication (Renderer.cpp:994-1015) -----------------------------
void Renderer::build_tlas() {
  std::vector<uint32_t> ids(blas_instance_ids_.begin(), blas_instance_ids_.end()); // ascending
  const uint32_t n = (uint32_t)ids.size();
  tlas_nodes_.assign(std::max<size_t>(ids.size() * 2, 1), rt_tlas_node{});
  last_tlas_on_gpu = false;
  if (n) {
    rta::tlas_leaves(tlas_nodes_.data() + 1, blas_instances_.data(), ids.data(), n, blases_.data(), bvh_nodes_.data());
    // The clustering loop is O(n^2)..O(n^3) and sequential; from about 1500 instances on, one GPU workgroup with
    // the live list in registers finishes it sooner than a host core (same nodes, bit for bit).
    if (ctx_ && tlas_gpu_min_instances && n >= tlas_gpu_min_instances && n <= RT_TLAS_GPU_MAX_LEAVES) {
      std::vector<rt_tlas_node> leaves(tlas_nodes_.begin() + 1, tlas_nodes_.begin() + 1 + n);
      uint32_t count = 0;
      if (rt_build_tlas_cluster(ctx_, leaves.data(), n, tlas_nodes_.data(), &count, &last_tlas_kernel_ms) == RT_OK) {
        tlas_node_count_ = count;
        last_tlas_on_gpu = true;
      } else {
        warn(std::string("build_tlas: GPU clustering refused (") + rt_last_error_string() + "), building on the host");
      }
    }
    if (!last_tlas_on_gpu)
      tlas_node_count_ = rta::tlas_cluster(tlas_nodes_.data(), n);
  } else {
    tlas_node_count_ = 0;
  }
  rebuild_tlas_ = false;
  instances_dirty_ = true;
  frame_index = 0;
}

const rt_scene_desc &Renderer::scene() {
  if (rebuild_tlas_)
    build_tlas();
  // texel blob: textures of live lambert slots, 4-byte aligned
  texel_blob_.clear();
  for (size_t i = 0; i < textures_.size(); ++i) {
    textures_[i].offset = (uint32_t)texel_blob_.size();
    if (texture_pixels_[i].empty()) { // never-used slot: 1x1 black keeps indices valid
      textures_[i].width = textures_[i].height = 1;
      texel_blob_.insert(texel_blob_.end(), 4, 0);
    } else {
      texel_blob_.insert(texel_blob_.end(), texture_pixels_[i].begin(), texture_pixels_[i].end());
    }
  }
  rt_scene_desc &d = desc_;
  memset(&d, 0, sizeof d);
  d.tri_geom = tri_geom_.data(), d.tri_shading = tri_shading_.data(), d.tri_ids = tri_ids_.data();
  d.tri_capacity = (uint32_t)tri_ids_.size();
  d.bvh_nodes = bvh_nodes_.data(), d.bvh_node_count = (uint32_t)bvh_nodes_.size();
  d.blas = blases_.data(), d.blas_count = (uint32_t)blases_.size();
  d.instances = blas_instances_.data(), d.instance_count = (uint32_t)blas_instances_.size();
  d.tlas_nodes = tlas_nodes_.data(), d.tlas_node_count = tlas_node_count_;
  d.lamberts = lambert_mats_.data(), d.lambert_count = (uint32_t)lambert_mats_.size();
  d.metals = metal_mats_.data(), d.metal_count = (uint32_t)metal_mats_.size();
  d.dielectrics = dielectric_mats_.data(), d.dielectric_count = (uint32_t)dielectric_mats_.size();
  d.emissives = emissive_mats_.data(), d.emissive_count = (uint32_t)emissive_mats_.size();
  d.textures = textures_.data(), d.texture_count = (uint32_t)textures_.size();
  d.texels = texel_blob_.data(), d.texel_bytes = texel_blob_.size();
  return d;
}

int Renderer::sync_device() {
  if (!ctx_) {
    last_error_ = "renderer was created host-only (device_id < 0): no GPU path, and no CPU fallback";
    return RT_ERR_NO_DEVICE;
  }
  if (rebuild_tlas_)
    build_tlas();
  int rc = RT_OK;
  if (geometry_dirty_) {
    rc = rt_upload_scene(ctx_, &scene());
    if (rc == RT_OK)
      geometry_dirty_ = instances_dirty_ = false;
  } else if (instances_dirty_) {
    rc = rt_update_instances(ctx_, blas_instances_.data(), (uint32_t)blas_instances_.size(),
                             tlas_nodes_.data(), tlas_node_count_);
    if (rc == RT_OK)
      instances_dirty_ = false;
  }
  if (rc != RT_OK)
    last_error_ = rt_last_error_string();
  return rc;
}

// Renderer::render camera block, Renderer.cpp:506-537 (+ RTIOW defocus disk when
// defocus_angle > 0; the reference's kernel is always a pinhole)
rt_camera_uniforms Renderer::camera_uniforms(Camera &camera) const {
  const bool defocus = camera.defocus_angle > 0.f;
  float focal_length = defocus ? camera.focus_dist : length(camera.position - camera.look_at);
  float theta = camera.fov * 3.1415926535897932385f / 180.f; // degrees_to_radians, Defines.hpp:86-89
  float h = tanf(theta / 2.f);
  float viewport_height = 2.f * h * focal_length;
  float viewport_width = viewport_height * (static_cast<float>(width) / height);
  camera.w = normalize(camera.position - camera.look_at);
  camera.u = normalize(cross(camera.v_up, camera.w));
  camera.v = cross(camera.w, camera.u);
  vec3 viewport_u = viewport_width * camera.u;
  vec3 viewport_v = viewport_height * -camera.v;
  vec3 pixel_delta_u = viewport_u / static_cast<float>(width);
  vec3 pixel_delta_v = viewport_v / static_cast<float>(height);
  vec3 viewport_upper_left =
      camera.position - (focal_length * camera.w) - 0.5f * (viewport_u + viewport_v);
  vec3 pixel00 = viewport_upper_left + 0.5f * (pixel_delta_u + pixel_delta_v);
  rt_camera_uniforms u;
  memset(&u, 0, sizeof u);
  pixel00.store(u.pixel00_loc), u.pixel00_loc[3] = 1.f;
  pixel_delta_u.store(u.pixel_delta_u), u.pixel_delta_u[3] = 1.f;
  pixel_delta_v.store(u.pixel_delta_v), u.pixel_delta_v[3] = 1.f;
  camera.position.store(u.camera_center), u.camera_center[3] = 1.f;
  if (defocus) {
    float defocus_radius =
        camera.focus_dist * tanf((camera.defocus_angle / 2.f) * 3.1415926535897932385f / 180.f);
    (camera.u * defocus_radius).store(u.defocus_disk_u), u.defocus_disk_u[3] = 1.f;
    (camera.v * defocus_radius).store(u.defocus_disk_v), u.defocus_disk_v[3] = 1.f;
  }
  return u;
}

// Renderer::render, Renderer.cpp:488-652
int Renderer::render(Camera &camera) { return render_frames(camera, 1u); }

// `frames` iterations of the reference's frame loop (PathTracer.cpp:396-411: render(cam) once per displayed frame) with
// the scene and camera as they are now, handed to the device in one call (rt_render_frames): the image and
// frame_index afterwards are those of `frames` successive render() calls.
int Renderer::render_frames(Camera &camera, uint32_t frames) {
  if (frames == 0)
    return RT_OK;
  if (camera.changed) {
    frame_index = 0;
    camera.changed = false;
  }
  int rc = sync_device(); // lambert_mats.update + build_tlas of the reference
  if (rc != RT_OK)
    return rc;
  rt_camera_uniforms u = camera_uniforms(camera);
  if ((rc = rt_set_camera(ctx_, &u)) != RT_OK) {
    last_error_ = rt_last_error_string();
    return rc;
  }
  if (frame_index == 0 && (rc = rt_clear_accum(ctx_)) != RT_OK) { // :584-606
    last_error_ = rt_last_error_string();
    return rc;
  }
  rt_render_params pc;
  memset(&pc, 0, sizeof pc);
  pc.image_width = width, pc.image_height = height;
  pc.frame_index = frame_offset + frame_index * frame_stride;
  pc.accum_count = frame_index;
  pc.sqrt_spp = (uint32_t)sqrtf((float)samples_per_pixel); // :625 (truncating)
  if (pc.sqrt_spp == 0)
    pc.sqrt_spp = 1;
  pc.max_depth = max_depth;
  pc.rng_mode = rng_mode, pc.aabb_mode = aabb_mode;
  pc.row_begin = row_begin, pc.row_end = row_end, pc.row_group_stride = row_group_stride;
  pc.flags = render_flags;
  if (frame_stride == 1) {
    if ((rc = rt_render_frames(ctx_, &pc, frames)) != RT_OK) {
      last_error_ = rt_last_error_string();
      return rc;
    }
    frame_index += frames;
    return RT_OK;
  }
  for (uint32_t k = 0; k < frames; ++k) { // frames dealt round-robin to several GPUs are not consecutive: one by one
    pc.frame_index = frame_offset + frame_index * frame_stride;
    pc.accum_count = frame_index;
    if ((rc = rt_render(ctx_, &pc)) != RT_OK) {
      last_error_ = rt_last_error_string();
      return rc;
    }
    ++frame_index;
  }
  return RT_OK;
}

int Renderer::read_image(float *rgba32f) {
  if (!ctx_) {
    last_error_ = "host-only renderer has no image";
    return RT_ERR_NO_DEVICE;
  }
  int rc = rt_read_accum(ctx_, rgba32f);
  if (rc != RT_OK)
    last_error_ = rt_last_error_string();
  return rc;
}

int Renderer::wait() {
  if (!ctx_)
    return RT_ERR_NO_DEVICE;
  int rc = rt_sync(ctx_);
  if (rc != RT_OK)
    last_error_ = rt_last_error_string();
  return rc;
}

} // namespace rta
