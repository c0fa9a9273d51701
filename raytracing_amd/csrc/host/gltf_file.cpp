// gltf_file.cpp — glTF 2.0 import as the reference's scene graph does it: load_gltf_scene
// (src/SceneGraph.cpp:168-407), gltf_load_texture (:42-166) and the world transforms of
// SceneGraph::update_transforms (:544-566) with Transform::get_mat4 / set_transform (src/Transform.hpp:15-33).
//
// The reference parses with fastgltf (third-party submodule, not in the snapshot; options: no extensions,
// LoadExternalBuffers, GenerateMeshIndices).  This file reads the same container formats from the glTF 2.0
// specification — .gltf JSON with external or base64 buffers, and .glb — and reproduces what the reference does with
// the asset:
//   * one Lambert per material, in asset order: the baseColorTexture's image as an RGBA8 texture, else a 1x1 texture
//     of clamp(baseColorFactor) * 255 including alpha (:199-223);
//   * breadth-first walk of the default scene's nodes (:225-288); a node's local matrix is T*R*S of its TRS, or its
//     `matrix` decomposed by Transform::set_transform and recomposed (:258-278, a lossy round trip that is kept);
//   * one BLAS + one instance per triangle primitive (:289-399): POSITION, NORMAL (or the reference's flat normals,
//     three per triangle in index order), TEXCOORD_0 (or zeros), indices widened to u32 (or 0..n-1);
//   * instance transform = product of the local matrices down the hierarchy, parent * local (:557), with the extra
//     identity levels the scene graph inserts for several scene roots and for meshes with several primitives.
#include "gltf_file.h"

#include <stdio.h>
#include <string.h>

#include <deque>

#include "image_decode.h"
#include "scene_file.h"

namespace rta {
namespace {

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp)
    return false;
  uint8_t buf[65536];
  size_t n;
  out.clear();
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0)
    out.insert(out.end(), buf, buf + n);
  fclose(fp);
  return true;
}

bool base64_decode(const std::string &s, size_t from, std::vector<uint8_t> &out) {
  out.clear();
  uint32_t acc = 0;
  int bits = 0;
  for (size_t i = from; i < s.size(); ++i) {
    const char ch = s[i];
    int v;
    if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
    else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
    else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
    else if (ch == '+' || ch == '-') v = 62;
    else if (ch == '/' || ch == '_') v = 63;
    else if (ch == '=') break;
    else if (ch == '\n' || ch == '\r' || ch == ' ') continue;
    else return false;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out.push_back((uint8_t)(acc >> bits));
    }
  }
  return true;
}

std::string percent_decode(const std::string &s) {
  std::string out;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '%' && i + 2 < s.size() + 0 && isxdigit((unsigned char)s[i + 1]) && isxdigit((unsigned char)s[i + 2])) {
      out.push_back((char)strtol(s.substr(i + 1, 2).c_str(), nullptr, 16));
      i += 2;
    } else {
      out.push_back(s[i]);
    }
  }
  return out;
}

// uri -> bytes: "data:<mime>;base64,<payload>" or a path relative to the asset's directory
bool load_uri(const std::string &uri, const std::string &dir, std::vector<uint8_t> &out, std::string &err) {
  if (uri.compare(0, 5, "data:") == 0) {
    const size_t comma = uri.find(',');
    if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma ||
        !base64_decode(uri, comma + 1, out)) {
      err = "bad data: URI";
      return false;
    }
    return true;
  }
  if (uri.find("://") != std::string::npos) {
    err = "URI is not local: " + uri; // HASSERT_MSG(filePath.uri.isLocalPath()), :84-87
    return false;
  }
  const std::string path = dir + "/" + percent_decode(uri);
  if (!read_file(path, out)) {
    err = "cannot open " + path;
    return false;
  }
  return true;
}

struct View {
  size_t buffer = 0, offset = 0, length = 0, stride = 0;
};
struct Accessor {
  int view = -1;
  size_t offset = 0, count = 0;
  int component = 0, width = 0; // componentType, number of components
  bool normalized = false;
};

struct Asset {
  Json doc;
  std::string dir;
  std::vector<std::vector<uint8_t>> buffers;
  std::vector<View> views;
  std::vector<Accessor> accessors;
  std::string err;

  bool fail(const std::string &m) {
    if (err.empty())
      err = m;
    return false;
  }
  static const Json *arr(const Json *j, const char *key) {
    const Json *a = j ? j->find(key) : nullptr;
    return a && a->kind == Json::Array ? a : nullptr;
  }
  static bool get_index(const Json *obj, const char *key, size_t limit, int &out) { // optional non-negative integer
    out = -1;
    const Json *v = obj ? obj->find(key) : nullptr;
    if (!v)
      return true;
    if (v->kind != Json::Number || v->num < 0 || v->num != (double)(size_t)v->num || (size_t)v->num >= limit)
      return false;
    out = (int)v->num;
    return true;
  }
  static size_t get_size(const Json *obj, const char *key, size_t def, bool &ok) {
    const Json *v = obj ? obj->find(key) : nullptr;
    if (!v)
      return def;
    if (v->kind != Json::Number || v->num < 0 || v->num > 9.0e15 || v->num != (double)(size_t)v->num) {
      ok = false;
      return def;
    }
    return (size_t)v->num;
  }

  bool load_buffers(const std::vector<uint8_t> &glb_bin, bool has_glb_bin) {
    const Json *bs = arr(&doc, "buffers");
    for (size_t i = 0; bs && i < bs->arr.size(); ++i) {
      const Json &b = bs->arr[i];
      bool ok = true;
      const size_t len = get_size(&b, "byteLength", 0, ok);
      std::vector<uint8_t> bytes;
      const Json *uri = b.find("uri");
      if (uri && uri->kind == Json::String) {
        std::string e;
        if (!load_uri(uri->str, dir, bytes, e))
          return fail("buffer " + std::to_string(i) + ": " + e);
      } else if (i == 0 && has_glb_bin) {
        bytes = glb_bin;
      } else {
        return fail("buffer " + std::to_string(i) + " has no uri");
      }
      if (!ok || bytes.size() < len)
        return fail("buffer " + std::to_string(i) + " is shorter than its byteLength");
      buffers.push_back(std::move(bytes));
    }
    const Json *vs = arr(&doc, "bufferViews");
    for (size_t i = 0; vs && i < vs->arr.size(); ++i) {
      const Json &v = vs->arr[i];
      View w;
      bool ok = true;
      int buf;
      if (!get_index(&v, "buffer", buffers.size(), buf) || buf < 0)
        return fail("bufferView " + std::to_string(i) + ": bad buffer index");
      w.buffer = (size_t)buf;
      w.offset = get_size(&v, "byteOffset", 0, ok), w.length = get_size(&v, "byteLength", 0, ok);
      w.stride = get_size(&v, "byteStride", 0, ok);
      if (!ok || w.offset > buffers[w.buffer].size() || w.length > buffers[w.buffer].size() - w.offset)
        return fail("bufferView " + std::to_string(i) + " exceeds its buffer");
      views.push_back(w);
    }
    const Json *as = arr(&doc, "accessors");
    for (size_t i = 0; as && i < as->arr.size(); ++i) {
      const Json &a = as->arr[i];
      Accessor acc;
      bool ok = true;
      if (!get_index(&a, "bufferView", views.size(), acc.view))
        return fail("accessor " + std::to_string(i) + ": bad bufferView");
      if (a.find("sparse"))
        return fail("accessor " + std::to_string(i) + ": sparse accessors are not supported");
      acc.offset = get_size(&a, "byteOffset", 0, ok), acc.count = get_size(&a, "count", 0, ok);
      acc.component = (int)get_size(&a, "componentType", 0, ok);
      const Json *n = a.find("normalized");
      acc.normalized = n && n->kind == Json::Bool && n->b;
      const Json *t = a.find("type");
      const std::string type = t && t->kind == Json::String ? t->str : "";
      acc.width = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
      if (!ok || acc.width == 0 || acc.count > (1u << 28))
        return fail("accessor " + std::to_string(i) + ": bad description");
      accessors.push_back(acc);
    }
    return true;
  }

  static size_t component_bytes(int c) {
    return c == 5120 || c == 5121 ? 1 : c == 5122 || c == 5123 ? 2 : c == 5125 || c == 5126 ? 4 : 0;
  }
  // element k of an accessor -> pointer to its first component, or nullptr (view-less accessors read as zeros)
  bool locate(const Accessor &a, const uint8_t *&base, size_t &stride) {
    const size_t elem = component_bytes(a.component) * (size_t)a.width;
    if (elem == 0)
      return fail("unsupported accessor component type");
    if (a.view < 0) {
      base = nullptr, stride = 0;
      return true;
    }
    const View &v = views[(size_t)a.view];
    // byteStride: glTF 2.0 allows 4..252 in steps of 4 and no less than the element; anything else is refused
    // here, before it is multiplied (a crafted 2^40 stride times 2^24 elements wraps size_t)
    if (v.stride && (v.stride < 4 || v.stride > 252 || (v.stride & 3u) || v.stride < elem))
      return fail("bufferView byteStride must be a multiple of 4 in 4..252 and at least the element size");
    stride = v.stride ? v.stride : elem;
    // the last element must end inside the view; written without a multiplication that could overflow
    if (a.count && (a.offset > v.length || elem > v.length - a.offset ||
                    a.count - 1 > (v.length - a.offset - elem) / stride))
      return fail("accessor exceeds its bufferView");
    base = buffers[v.buffer].data() + v.offset + a.offset;
    return true;
  }

  // fastgltf::iterateAccessor<u32>: unsigned index types widened
  bool read_indices(size_t accessor, std::vector<uint32_t> &out) {
    const Accessor &a = accessors[accessor];
    const uint8_t *base;
    size_t stride;
    if (a.width != 1 || !(a.component == 5121 || a.component == 5123 || a.component == 5125))
      return fail("index accessor must be a SCALAR of an unsigned integer type");
    if (!locate(a, base, stride))
      return false;
    out.resize(a.count);
    for (size_t k = 0; k < a.count; ++k) {
      const uint8_t *p = base ? base + k * stride : nullptr;
      uint32_t v = 0;
      if (p) {
        if (a.component == 5121) v = p[0];
        else if (a.component == 5123) v = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
        else memcpy(&v, p, 4);
      }
      out[k] = v;
    }
    return true;
  }
  // fastgltf::iterateAccessor<fvecN>: floats as stored; normalized integers divided by the type's maximum
  bool read_floats(size_t accessor, int width, std::vector<float> &out) {
    const Accessor &a = accessors[accessor];
    const uint8_t *base;
    size_t stride;
    if (a.width != width)
      return fail("vertex attribute has the wrong number of components");
    if (!locate(a, base, stride))
      return false;
    if (a.component != 5126 && !a.normalized)
      return fail("vertex attribute is neither FLOAT nor a normalized integer");
    out.resize(a.count * (size_t)width);
    const size_t cb = component_bytes(a.component);
    for (size_t k = 0; k < a.count; ++k)
      for (int c = 0; c < width; ++c) {
        const uint8_t *p = base ? base + k * stride + (size_t)c * cb : nullptr;
        float f = 0.f;
        if (p) {
          switch (a.component) {
          case 5126: memcpy(&f, p, 4); break;
          case 5121: f = (float)p[0] / 255.0f; break;
          case 5123: f = (float)(uint16_t)(p[0] | (p[1] << 8)) / 65535.0f; break;
          case 5120: f = (float)(int8_t)p[0] / 127.0f; f = f < -1.f ? -1.f : f; break;
          case 5122: f = (float)(int16_t)(p[0] | (p[1] << 8)) / 32767.0f; f = f < -1.f ? -1.f : f; break;
          default: return fail("unsupported vertex attribute component type");
          }
        }
        out[k * (size_t)width + (size_t)c] = f;
      }
    return true;
  }
};

float num_f(const Json &j, float def) { return j.kind == Json::Number ? (float)j.num : def; }

// glm::quat_cast(mat3), gtc/quaternion.inl
quat quat_cast(const vec3 (&m)[3]) {
  const float fourXSquaredMinus1 = m[0].x - m[1].y - m[2].z;
  const float fourYSquaredMinus1 = m[1].y - m[0].x - m[2].z;
  const float fourZSquaredMinus1 = m[2].z - m[0].x - m[1].y;
  const float fourWSquaredMinus1 = m[0].x + m[1].y + m[2].z;
  int biggest = 0;
  float fourBiggestSquaredMinus1 = fourWSquaredMinus1;
  if (fourXSquaredMinus1 > fourBiggestSquaredMinus1) fourBiggestSquaredMinus1 = fourXSquaredMinus1, biggest = 1;
  if (fourYSquaredMinus1 > fourBiggestSquaredMinus1) fourBiggestSquaredMinus1 = fourYSquaredMinus1, biggest = 2;
  if (fourZSquaredMinus1 > fourBiggestSquaredMinus1) fourBiggestSquaredMinus1 = fourZSquaredMinus1, biggest = 3;
  const float biggestVal = sqrtf(fourBiggestSquaredMinus1 + 1.0f) * 0.5f;
  const float mult = 0.25f / biggestVal;
  quat q;
  switch (biggest) {
  case 0: q.w = biggestVal, q.x = (m[1].z - m[2].y) * mult, q.y = (m[2].x - m[0].z) * mult, q.z = (m[0].y - m[1].x) * mult; break;
  case 1: q.w = (m[1].z - m[2].y) * mult, q.x = biggestVal, q.y = (m[0].y + m[1].x) * mult, q.z = (m[2].x + m[0].z) * mult; break;
  case 2: q.w = (m[2].x - m[0].z) * mult, q.x = (m[0].y + m[1].x) * mult, q.y = biggestVal, q.z = (m[1].z + m[2].y) * mult; break;
  default: q.w = (m[0].y - m[1].x) * mult, q.x = (m[2].x + m[0].z) * mult, q.y = (m[1].z + m[2].y) * mult, q.z = biggestVal; break;
  }
  return q;
}

// Transform::set_transform followed by get_mat4, src/Transform.hpp:15-33
mat4 decompose_recompose(const mat4 &matrix) {
  const vec3 position(matrix[3].x, matrix[3].y, matrix[3].z);
  const vec3 c0(matrix[0].x, matrix[0].y, matrix[0].z), c1(matrix[1].x, matrix[1].y, matrix[1].z),
      c2(matrix[2].x, matrix[2].y, matrix[2].z);
  const vec3 scl(length(c0), length(c1), length(c2));
  const vec3 rot[3] = {c0 / scl.x, c1 / scl.y, c2 / scl.z};
  return trs(position, quat_cast(rot), scl);
}

mat4 node_local_matrix(const Json &node) {
  const Json *m = node.find("matrix");
  if (m && m->kind == Json::Array && m->arr.size() == 16) { // :268-277
    float f[16];
    for (int i = 0; i < 16; ++i)
      f[i] = num_f(m->arr[(size_t)i], 0.f);
    return decompose_recompose(mat4(f));
  }
  vec3 t(0.f, 0.f, 0.f), s(1.f, 1.f, 1.f); // :258-266, fastgltf's TRS defaults
  quat q;
  const Json *jt = node.find("translation"), *jr = node.find("rotation"), *js = node.find("scale");
  if (jt && jt->kind == Json::Array && jt->arr.size() == 3)
    t = vec3(num_f(jt->arr[0], 0.f), num_f(jt->arr[1], 0.f), num_f(jt->arr[2], 0.f));
  if (jr && jr->kind == Json::Array && jr->arr.size() == 4) // glTF order x, y, z, w (gltf_set_quat, :28-34)
    q.x = num_f(jr->arr[0], 0.f), q.y = num_f(jr->arr[1], 0.f), q.z = num_f(jr->arr[2], 0.f), q.w = num_f(jr->arr[3], 1.f);
  if (js && js->kind == Json::Array && js->arr.size() == 3)
    s = vec3(num_f(js->arr[0], 1.f), num_f(js->arr[1], 1.f), num_f(js->arr[2], 1.f));
  return trs(t, q, s);
}

uint8_t factor_byte(float v) { // static_cast<u8>(std::clamp(v, 0.0f, 1.0f) * 255.0f), :212-219
  const float c = v < 0.0f ? 0.0f : (1.0f < v ? 1.0f : v);
  return (uint8_t)(c * 255.0f);
}

} // namespace

int64_t import_gltf(const std::string &path, Renderer &renderer, const mat4 &parent_global, std::string &err,
                    GltfImportInfo *info) {
  std::vector<uint8_t> file;
  if (!read_file(path, file)) {
    err = "cannot open " + path;
    return -1;
  }
  Asset as;
  const size_t slash = path.find_last_of("/\\");
  as.dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
  std::string json_text;
  std::vector<uint8_t> glb_bin;
  bool has_bin = false;
  if (file.size() >= 12 && !memcmp(file.data(), "glTF", 4)) { // GLB container
    auto le32 = [&](size_t at) { return (uint32_t)file[at] | ((uint32_t)file[at + 1] << 8) | ((uint32_t)file[at + 2] << 16) | ((uint32_t)file[at + 3] << 24); };
    if (le32(4) != 2 || le32(8) > file.size()) {
      err = path + ": bad GLB header";
      return -1;
    }
    size_t pos = 12;
    const size_t end = le32(8);
    while (pos + 8 <= end) {
      const uint32_t clen = le32(pos), ctype = le32(pos + 4);
      if (clen > end - pos - 8) {
        err = path + ": GLB chunk exceeds the file";
        return -1;
      }
      if (ctype == 0x4E4F534Au && json_text.empty())
        json_text.assign((const char *)&file[pos + 8], clen);
      else if (ctype == 0x004E4942u && !has_bin)
        glb_bin.assign(file.begin() + (long)pos + 8, file.begin() + (long)pos + 8 + clen), has_bin = true;
      pos += 8 + (size_t)((clen + 3u) & ~3u);
    }
  } else {
    json_text.assign((const char *)file.data(), file.size());
  }
  std::string jerr;
  if (!json_parse(json_text, as.doc, jerr) || as.doc.kind != Json::Object) {
    err = path + ": " + (jerr.empty() ? "not a glTF document" : jerr);
    return -1;
  }
  if (!as.load_buffers(glb_bin, has_bin)) {
    err = path + ": " + as.err;
    return -1;
  }
  auto failed = [&](const std::string &m) {
    err = path + ": " + (as.err.empty() ? m : as.err);
    return (int64_t)-1;
  };

  GltfImportInfo local{};
  const Json *images = Asset::arr(&as.doc, "images"), *textures = Asset::arr(&as.doc, "textures");
  const Json *materials = Asset::arr(&as.doc, "materials"), *meshes = Asset::arr(&as.doc, "meshes");
  const Json *nodes = Asset::arr(&as.doc, "nodes"), *scenes = Asset::arr(&as.doc, "scenes");
  const size_t n_images = images ? images->arr.size() : 0, n_textures = textures ? textures->arr.size() : 0;
  const size_t n_materials = materials ? materials->arr.size() : 0, n_meshes = meshes ? meshes->arr.size() : 0;
  const size_t n_nodes = nodes ? nodes->arr.size() : 0;

  // ---- materials (:198-223) ----
  std::vector<rt_material_handle> material_handles(n_materials);
  for (size_t i = 0; i < n_materials; ++i) {
    const Json &mat = materials->arr[i];
    const Json *pbr = mat.find("pbrMetallicRoughness");
    const Json *bct = pbr ? pbr->find("baseColorTexture") : nullptr;
    if (bct && bct->kind == Json::Object) {
      int ti, src;
      if (!Asset::get_index(bct, "index", n_textures, ti) || ti < 0)
        return failed("material " + std::to_string(i) + ": bad texture index");
      if (!Asset::get_index(&textures->arr[(size_t)ti], "source", n_images, src) || src < 0)
        return failed("texture " + std::to_string(ti) + " has no image source");
      const Json &img = images->arr[(size_t)src];
      std::vector<uint8_t> bytes;
      const Json *uri = img.find("uri");
      int view;
      if (uri && uri->kind == Json::String) {
        std::string e;
        if (!load_uri(uri->str, as.dir, bytes, e))
          return failed("image " + std::to_string(src) + ": " + e);
      } else if (Asset::get_index(&img, "bufferView", as.views.size(), view) && view >= 0) {
        const View &v = as.views[(size_t)view];
        bytes.assign(as.buffers[v.buffer].begin() + (long)v.offset, as.buffers[v.buffer].begin() + (long)(v.offset + v.length));
      } else {
        return failed("image " + std::to_string(src) + " has neither uri nor bufferView");
      }
      Image decoded;
      std::string e;
      if (!decode_image_rgba8(bytes, decoded, e)) // stbi_load_from_memory(.., STBI_rgb_alpha), :68-71
        return failed("image " + std::to_string(src) + ": " + e);
      material_handles[i] = renderer.add_lambert_material((int32_t)decoded.width, (int32_t)decoded.height, decoded.rgba.data());
    } else {
      float f[4] = {1.f, 1.f, 1.f, 1.f};
      const Json *bcf = pbr ? pbr->find("baseColorFactor") : nullptr;
      if (bcf && bcf->kind == Json::Array && bcf->arr.size() == 4)
        for (int c = 0; c < 4; ++c)
          f[c] = num_f(bcf->arr[(size_t)c], 1.f);
      const uint8_t px[4] = {factor_byte(f[0]), factor_byte(f[1]), factor_byte(f[2]), factor_byte(f[3])};
      material_handles[i] = renderer.add_lambert_material(1, 1, px);
    }
    if (material_handles[i].type == RT_MATERIAL_NONE)
      return failed("material " + std::to_string(i) + ": " + renderer.last_error());
  }
  local.materials = (uint32_t)n_materials;

  // ---- nodes, breadth first from the default scene's roots (:225-288) ----
  if (!scenes || scenes->arr.empty())
    return failed("the asset has no scene");
  int scene_index;
  if (!Asset::get_index(&as.doc, "scene", scenes->arr.size(), scene_index))
    return failed("bad default scene index");
  const Json *roots = Asset::arr(&scenes->arr[(size_t)(scene_index < 0 ? 0 : scene_index)], "nodes");
  const size_t n_roots = roots ? roots->arr.size() : 0;
  // one root: attached to the caller's node; several: under a new node with an identity local matrix (:236-238)
  const mat4 roots_parent = n_roots == 1 ? parent_global : parent_global * mat4(1.f);
  std::vector<mat4> global(n_nodes);
  std::vector<char> visited(n_nodes, 0);
  std::deque<size_t> queue;
  auto node_index = [&](const Json &j, size_t &out) {
    if (j.kind != Json::Number || j.num < 0 || j.num != (double)(size_t)j.num || (size_t)j.num >= n_nodes)
      return false;
    out = (size_t)j.num;
    return true;
  };
  std::vector<mat4> parent_of(n_nodes);
  for (size_t i = 0; i < n_roots; ++i) {
    size_t k;
    if (!node_index(roots->arr[i], k) || visited[k])
      return failed("bad scene root");
    visited[k] = 1, parent_of[k] = roots_parent;
    queue.push_back(k);
  }
  int64_t triangles = 0;
  while (!queue.empty()) {
    const size_t k = queue.front();
    queue.pop_front();
    const Json &node = nodes->arr[k];
    global[k] = parent_of[k] * node_local_matrix(node); // :557
    ++local.nodes;
    const Json *children = Asset::arr(&node, "children");
    for (size_t i = 0; children && i < children->arr.size(); ++i) {
      size_t c;
      if (!node_index(children->arr[i], c) || visited[c])
        return failed("node " + std::to_string(k) + ": bad child (index out of range or a cycle)");
      visited[c] = 1, parent_of[c] = global[k];
      queue.push_back(c);
    }
    int mesh_index;
    if (!Asset::get_index(&node, "mesh", n_meshes, mesh_index))
      return failed("node " + std::to_string(k) + ": bad mesh index");
    if (mesh_index < 0)
      continue;
    ++local.meshes;
    const Json *prims = Asset::arr(&meshes->arr[(size_t)mesh_index], "primitives");
    const size_t n_prims = prims ? prims->arr.size() : 0;
    for (size_t p = 0; p < n_prims; ++p) {
      const Json &prim = prims->arr[p];
      const std::string where = "mesh " + std::to_string(mesh_index) + " primitive " + std::to_string(p);
      bool ok = true;
      if (Asset::get_size(&prim, "mode", 4, ok) != 4 || !ok)
        return failed(where + ": not a triangle list"); // HASSERT_MSG(primitive.type == Triangles), :299
      const Json *attrs = prim.find("attributes");
      int a_pos, a_nrm, a_uv, a_idx, mat_index;
      if (!attrs || !Asset::get_index(attrs, "POSITION", as.accessors.size(), a_pos) || a_pos < 0 ||
          !Asset::get_index(attrs, "NORMAL", as.accessors.size(), a_nrm) ||
          !Asset::get_index(attrs, "TEXCOORD_0", as.accessors.size(), a_uv) ||
          !Asset::get_index(&prim, "indices", as.accessors.size(), a_idx) ||
          !Asset::get_index(&prim, "material", n_materials, mat_index))
        return failed(where + ": bad accessor or material index");
      std::vector<float> positions, normals, uvs;
      std::vector<uint32_t> indices;
      if (!as.read_floats((size_t)a_pos, 3, positions))
        return failed(where);
      const size_t n_vertices = positions.size() / 3;
      if (a_idx >= 0) {
        if (!as.read_indices((size_t)a_idx, indices))
          return failed(where);
      } else { // Options::GenerateMeshIndices
        indices.resize(n_vertices);
        for (size_t i = 0; i < n_vertices; ++i)
          indices[i] = (uint32_t)i;
      }
      for (uint32_t idx : indices)
        if (idx >= n_vertices)
          return failed(where + ": index out of range");
      if (indices.size() < 3)
        return failed(where + ": no triangles");
      if (a_nrm >= 0) {
        if (!as.read_floats((size_t)a_nrm, 3, normals))
          return failed(where);
      } else {
        // :338-352 — one face normal pushed three times per triangle, in index order; the array is then read
        // per VERTEX index by add_blas, exactly as the reference does
        for (size_t i = 0; i + 2 < indices.size(); i += 3) {
          const vec3 p0(&positions[3 * (size_t)indices[i]]), p1(&positions[3 * (size_t)indices[i + 1]]),
              p2(&positions[3 * (size_t)indices[i + 2]]);
          const vec3 n = normalize(cross(p1 - p0, p2 - p0));
          for (int r = 0; r < 3; ++r)
            normals.push_back(n.x), normals.push_back(n.y), normals.push_back(n.z);
        }
      }
      if (normals.size() < positions.size())
        normals.resize(positions.size(), 0.f); // the reference would read past the end here
      if (a_uv >= 0) {
        if (!as.read_floats((size_t)a_uv, 2, uvs))
          return failed(where);
      }
      if (uvs.size() < n_vertices * 2)
        uvs.resize(n_vertices * 2, 0.f); // tex_coords.resize(positions.size()), :384
      const size_t n_index = indices.size() - indices.size() % 3;
      const uint32_t blas = renderer.add_blas(positions.data(), normals.data(), uvs.data(), indices.data(), n_index);
      if (blas == UINT32_MAX)
        return failed(where + ": add_blas failed: " + renderer.last_error());
      const rt_material_handle mat = mat_index >= 0 ? material_handles[(size_t)mat_index] : renderer.default_material;
      const uint32_t inst = renderer.add_blas_instance(blas, mat4(1.f), mat); // :397
      if (inst == UINT32_MAX)
        return failed(where + ": add_blas_instance failed: " + renderer.last_error());
      // several primitives hang under "Mesh_<i>" child nodes with identity local matrices (:302-306)
      const mat4 prim_global = n_prims == 1 ? global[k] : global[k] * mat4(1.f);
      renderer.set_blas_instance_transform(inst, prim_global); // update_transforms, :557-561
      if (info)
        info->instance_ids.push_back(inst);
      ++local.primitives;
      triangles += (int64_t)(n_index / 3);
    }
  }
  if (info) {
    std::vector<uint32_t> ids = std::move(info->instance_ids);
    *info = local;
    info->instance_ids = std::move(ids);
    info->triangles = (uint64_t)triangles;
  }
  return triangles;
}

} // namespace rta
