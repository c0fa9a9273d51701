// scene_file.h — the reference's scene description files: a JSON document with
// a `camera` object and a `models` array of OBJ instances (README.md:53-68,
// Scenes/TestScene.json, loader semantics of the dead src/Scene.cpp:134-233).
#ifndef RTA_SCENE_FILE_H
#define RTA_SCENE_FILE_H

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "renderer.h"

namespace rta {

// Minimal JSON value (objects, arrays, numbers, strings, true/false/null).
struct Json {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj; // insertion order kept
  const Json *find(const std::string &key) const;
};
// Returns false and sets err ("line L: message") on malformed input.
bool json_parse(const std::string &text, Json &out, std::string &err);

struct SceneModel {
  std::string path; // as resolved against the JSON file's directory
  mat4 transform{1.f};
};
struct SceneFile {
  Camera camera;
  uint32_t screen_width = 0, screen_height = 0;
  uint32_t samples_per_pixel = 1, max_depth = 1;
  std::vector<SceneModel> models;
};
// Parses a scene file.  All camera keys are required, `models` and each model's
// `scale` / `rotation` / `translation` are optional (Scene.cpp:152-223).
bool load_scene_file(const std::string &json_path, SceneFile &out, std::string &err);

// Loads every model into the renderer: one BLAS + one instance per OBJ material
// group, Lambert from the map_Kd image if the material has one, else from Kd, (0.8, 0, 0.8) for faces without a material
// (Scene.cpp:58-70, 121-125, 182).  Returns the number of triangles added, or -1.
int64_t instantiate_scene(const SceneFile &scene, Renderer &renderer, std::string &err);

} // namespace rta
#endif
