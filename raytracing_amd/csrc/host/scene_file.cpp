// scene_file.cpp — JSON scene files (see scene_file.h).
#include "scene_file.h"

#include <ctype.h>

#include "gltf_file.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>

#include "meshes.h"

namespace rta {

// ---------------------------------------------------------------------------
// JSON
// ---------------------------------------------------------------------------
const Json *Json::find(const std::string &key) const {
  if (kind != Object)
    return nullptr;
  for (const auto &kv : obj)
    if (kv.first == key)
      return &kv.second;
  return nullptr;
}

namespace {

struct Parser {
  const std::string &s;
  size_t i = 0;
  std::string err;
  int depth = 0;

  explicit Parser(const std::string &text) : s(text) {}

  bool fail(const std::string &msg) {
    if (err.empty()) {
      size_t line = 1;
      for (size_t k = 0; k < i && k < s.size(); ++k)
        line += s[k] == '\n';
      err = "line " + std::to_string(line) + ": " + msg;
    }
    return false;
  }
  void skip_ws() {
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r'))
      ++i;
  }
  bool literal(const char *word) {
    size_t n = strlen(word);
    if (s.compare(i, n, word) != 0)
      return fail(std::string("expected '") + word + "'");
    i += n;
    return true;
  }
  bool string(std::string &out) {
    if (i >= s.size() || s[i] != '"')
      return fail("expected string");
    ++i;
    out.clear();
    while (i < s.size() && s[i] != '"') {
      char c = s[i++];
      if (c == '\\') {
        if (i >= s.size())
          return fail("unterminated escape");
        char e = s[i++];
        switch (e) {
        case '"': out += '"'; break;
        case '\\': out += '\\'; break;
        case '/': out += '/'; break;
        case 'b': out += '\b'; break;
        case 'f': out += '\f'; break;
        case 'n': out += '\n'; break;
        case 'r': out += '\r'; break;
        case 't': out += '\t'; break;
        case 'u': {
          if (i + 4 > s.size())
            return fail("short \\u escape");
          unsigned cp = 0;
          for (int k = 0; k < 4; ++k) {
            char h = s[i++];
            cp <<= 4;
            if (h >= '0' && h <= '9') cp |= (unsigned)(h - '0');
            else if (h >= 'a' && h <= 'f') cp |= (unsigned)(h - 'a' + 10);
            else if (h >= 'A' && h <= 'F') cp |= (unsigned)(h - 'A' + 10);
            else return fail("bad \\u escape");
          }
          // UTF-8 encode (surrogate pairs are passed through as two 3-byte sequences; paths are ASCII in practice)
          if (cp < 0x80) out += (char)cp;
          else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
          else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
          break;
        }
        default: return fail("unknown escape");
        }
      } else {
        out += c;
      }
    }
    if (i >= s.size())
      return fail("unterminated string");
    ++i;
    return true;
  }
  bool value(Json &out) {
    if (++depth > 64)
      return fail("nesting too deep");
    skip_ws();
    if (i >= s.size())
      return fail("unexpected end of input");
    bool ok = true;
    char c = s[i];
    if (c == '{') {
      out.kind = Json::Object;
      ++i;
      skip_ws();
      if (i < s.size() && s[i] == '}') {
        ++i;
      } else {
        while (ok) {
          skip_ws();
          std::string key;
          if (!(ok = string(key)))
            break;
          skip_ws();
          if (i >= s.size() || s[i] != ':') { ok = fail("expected ':'"); break; }
          ++i;
          Json v;
          if (!(ok = value(v)))
            break;
          out.obj.emplace_back(std::move(key), std::move(v));
          skip_ws();
          if (i < s.size() && s[i] == ',') { ++i; continue; }
          if (i < s.size() && s[i] == '}') { ++i; break; }
          ok = fail("expected ',' or '}'");
        }
      }
    } else if (c == '[') {
      out.kind = Json::Array;
      ++i;
      skip_ws();
      if (i < s.size() && s[i] == ']') {
        ++i;
      } else {
        while (ok) {
          Json v;
          if (!(ok = value(v)))
            break;
          out.arr.push_back(std::move(v));
          skip_ws();
          if (i < s.size() && s[i] == ',') { ++i; continue; }
          if (i < s.size() && s[i] == ']') { ++i; break; }
          ok = fail("expected ',' or ']'");
        }
      }
    } else if (c == '"') {
      out.kind = Json::String;
      ok = string(out.str);
    } else if (c == 't') {
      out.kind = Json::Bool, out.b = true;
      ok = literal("true");
    } else if (c == 'f') {
      out.kind = Json::Bool, out.b = false;
      ok = literal("false");
    } else if (c == 'n') {
      out.kind = Json::Null;
      ok = literal("null");
    } else if (c == '-' || (c >= '0' && c <= '9')) {
      const char *begin = s.c_str() + i;
      char *end = nullptr;
      out.kind = Json::Number;
      out.num = strtod(begin, &end);
      if (end == begin)
        ok = fail("bad number");
      else
        i += (size_t)(end - begin);
    } else {
      ok = fail(std::string("unexpected character '") + c + "'");
    }
    --depth;
    return ok;
  }
};

bool number(const Json &obj, const char *key, double &out, std::string &err) {
  const Json *v = obj.find(key);
  if (!v || v->kind != Json::Number) {
    err = std::string("camera.") + key + ": missing or not a number";
    return false;
  }
  out = v->num;
  return true;
}

bool vec(const Json &obj, const char *key, size_t n, float *out, std::string &err, const char *where) {
  const Json *v = obj.find(key);
  if (!v || v->kind != Json::Array || v->arr.size() != n) {
    err = std::string(where) + "." + key + ": expected an array of " + std::to_string(n) + " numbers";
    return false;
  }
  for (size_t k = 0; k < n; ++k) {
    if (v->arr[k].kind != Json::Number) {
      err = std::string(where) + "." + key + ": expected numbers";
      return false;
    }
    out[k] = (float)v->arr[k].num;
  }
  return true;
}

} // namespace

bool json_parse(const std::string &text, Json &out, std::string &err) {
  Parser p(text);
  out = Json();
  if (!p.value(out)) {
    err = p.err;
    return false;
  }
  p.skip_ws();
  if (p.i != text.size()) {
    p.fail("trailing characters after the document");
    err = p.err;
    return false;
  }
  return true;
}

// ---------------------------------------------------------------------------
// Scene files
// ---------------------------------------------------------------------------
bool load_scene_file(const std::string &json_path, SceneFile &out, std::string &err) {
  std::ifstream f(json_path, std::ios::binary);
  if (!f) {
    err = "JSON file not found at path: " + json_path; // Scene.cpp:140-144
    return false;
  }
  std::stringstream ss;
  ss << f.rdbuf();
  Json doc;
  if (!json_parse(ss.str(), doc, err)) {
    err = json_path + ": " + err;
    return false;
  }
  if (doc.kind != Json::Object) {
    err = json_path + ": the document is not an object";
    return false;
  }
  const Json *cam = doc.find("camera");
  if (!cam || cam->kind != Json::Object) {
    err = json_path + ": missing 'camera' object";
    return false;
  }
  out = SceneFile();
  float c[3], l[3], u[3];
  double defocus, focus, width, aspect, spp, depth, vfov;
  if (!vec(*cam, "center", 3, c, err, "camera") || !vec(*cam, "lookat", 3, l, err, "camera") ||
      !vec(*cam, "vup", 3, u, err, "camera") || !number(*cam, "defocus_angle", defocus, err) ||
      !number(*cam, "focus_dist", focus, err) || !number(*cam, "screen_width", width, err) ||
      !number(*cam, "aspect_ratio", aspect, err) || !number(*cam, "samples_per_pixel", spp, err) ||
      !number(*cam, "max_depth", depth, err) || !number(*cam, "vfov_deg", vfov, err)) {
    err = json_path + ": " + err;
    return false;
  }
  if (!(width >= 1 && width <= 65535) || !(aspect > 0) || !(spp >= 1 && spp <= 1e8) ||
      !(depth >= 0 && depth <= 1e6)) {
    err = json_path + ": camera.screen_width / aspect_ratio / samples_per_pixel / max_depth out of range";
    return false;
  }
  out.camera.position = vec3(c[0], c[1], c[2]);
  out.camera.look_at = vec3(l[0], l[1], l[2]);
  out.camera.v_up = vec3(u[0], u[1], u[2]);
  out.camera.fov = (float)vfov;
  out.camera.defocus_angle = (float)defocus;
  out.camera.focus_dist = (float)focus;
  out.camera.changed = true;
  out.screen_width = (uint32_t)width;
  // the dead renderer's init(width, aspect, ...): height = int(width / aspect), at least 1
  int h = (int)((float)out.screen_width / (float)aspect);
  out.screen_height = (uint32_t)(h < 1 ? 1 : h);
  out.samples_per_pixel = (uint32_t)spp;
  out.max_depth = (uint32_t)depth;

  std::string dir;
  size_t slash = json_path.find_last_of('/');
  if (slash != std::string::npos)
    dir = json_path.substr(0, slash + 1);
  const Json *models = doc.find("models"); // optional, Scene.cpp:185-186
  if (models) {
    if (models->kind != Json::Array) {
      err = json_path + ": 'models' is not an array";
      return false;
    }
    for (size_t k = 0; k < models->arr.size(); ++k) {
      const Json &m = models->arr[k];
      const std::string where = "models[" + std::to_string(k) + "]";
      const Json *path = m.find("model_path");
      if (m.kind != Json::Object || !path || path->kind != Json::String) {
        err = json_path + ": " + where + ".model_path: missing or not a string";
        return false;
      }
      SceneModel sm;
      sm.path = (!path->str.empty() && path->str[0] == '/') ? path->str : dir + path->str;
      // each factor is pre-multiplied in the order scale, rotate, translate (Scene.cpp:195-223)
      mat4 t(1.f);
      if (const Json *s = m.find("scale")) {
        if (s->kind != Json::Number) {
          err = json_path + ": " + where + ".scale: not a number";
          return false;
        }
        float sc = (float)s->num;
        t = scale(mat4(1.f), vec3(sc, sc, sc)) * t;
      }
      if (m.find("rotation")) {
        float r[4];
        if (!vec(m, "rotation", 4, r, err, where.c_str())) {
          err = json_path + ": " + err;
          return false;
        }
        float len = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (!(len > 0.f)) {
          err = json_path + ": " + where + ".rotation: zero axis";
          return false;
        }
        float inv = 1.0f / len;
        t = to_mat4(angle_axis(radians(r[3]), vec3(r[0] * inv, r[1] * inv, r[2] * inv))) * t;
      }
      if (m.find("translation")) {
        float tr[3];
        if (!vec(m, "translation", 3, tr, err, where.c_str())) {
          err = json_path + ": " + err;
          return false;
        }
        t = translate(mat4(1.f), vec3(tr[0], tr[1], tr[2])) * t;
      }
      sm.transform = t;
      out.models.push_back(sm);
    }
  }
  return true;
}

int64_t instantiate_scene(const SceneFile &scene, Renderer &renderer, std::string &err) {
  int64_t triangles = 0;
  // faces without a material: the loader's default Lambert, Scene.cpp:182
  rt_material_handle fallback = renderer.add_lambert_material(vec3(0.8f, 0.0f, 0.8f));
  for (const SceneModel &m : scene.models) {
    const size_t dot = m.path.find_last_of('.');
    std::string ext = dot == std::string::npos ? std::string() : m.path.substr(dot);
    for (char &ch : ext)
      ch = (char)tolower((unsigned char)ch);
    if (ext == ".gltf" || ext == ".glb") { // a glTF asset attached under a node that carries the model's transform
      const int64_t n = import_gltf(m.path, renderer, m.transform, err);
      if (n < 0)
        return -1;
      triangles += n;
      continue;
    }
    ObjModel obj;
    if (!load_obj(m.path, obj, err))
      return -1;
    for (size_t g = 0; g < obj.groups.size(); ++g) {
      const Mesh &mesh = obj.groups[g];
      if (mesh.indices.empty())
        continue;
      rt_material_handle mat = fallback;
      int mi = obj.group_material[g];
      if (mi >= 0) { // every .mtl entry becomes a Lambert: from map_Kd if present, else from Kd (Scene.cpp:58-70)
        const ObjMaterial &om = obj.materials[(size_t)mi];
        if (!om.map_kd.empty()) {
          const size_t slash = m.path.find_last_of("/\\");
          const std::string dir = slash == std::string::npos ? std::string(".") : m.path.substr(0, slash);
          mat = renderer.add_lambert_material(dir + "/" + om.map_kd); // Scene.cpp:63-65
          if (mat.type == RT_MATERIAL_NONE) {
            err = m.path + ": material '" + om.name + "': " + renderer.last_error();
            return -1;
          }
        } else {
          mat = renderer.add_lambert_material(vec3(om.kd[0], om.kd[1], om.kd[2]));
        }
      }
      uint32_t blas = renderer.add_blas(mesh.positions.data(), mesh.normals.data(), mesh.uvs.data(),
                                        mesh.indices.data(), mesh.indices.size());
      if (blas == UINT32_MAX) {
        err = m.path + ": add_blas failed: " + renderer.last_error();
        return -1;
      }
      if (renderer.add_blas_instance(blas, m.transform, mat) == UINT32_MAX) {
        err = m.path + ": add_blas_instance failed: " + renderer.last_error();
        return -1;
      }
      triangles += mesh.triangle_count();
    }
  }
  return triangles;
}

} // namespace rta
