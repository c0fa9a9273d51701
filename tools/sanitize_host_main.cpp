#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "image_decode.h"
#include "gltf_file.h"
#include "renderer.h"
#include "meshes.h"
#include "scene_file.h"
int main(int argc, char **argv) {
  int ok = 0, bad = 0;
  for (int i = 1; i < argc; ++i) {
    std::string p = argv[i], err;
    if (p.size() > 4 && (p.substr(p.size() - 4) == ".png" || p.substr(p.size() - 4) == ".pgm")) {
      rta::Image img;
      if (rta::load_image_rgba8(p, img, err)) {
        ++ok;
        // the PNG writer on what was decoded, read back by the decoder
        std::vector<uint8_t> rgb((size_t)img.width * img.height * 3);
        for (size_t k = 0; k < (size_t)img.width * img.height; ++k)
          memcpy(&rgb[3 * k], &img.rgba[4 * k], 3);
        const std::string tmp = p + ".rewritten.png";
        rta::Image back;
        if (!rta::write_png_rgb8(tmp, img.width, img.height, rgb.data()) || !rta::load_image_rgba8(tmp, back, err) ||
            back.width != img.width || back.height != img.height) {
          fprintf(stderr, "PNG writer round trip failed for %s\n", p.c_str());
          return 3;
        }
        for (size_t k = 0; k < (size_t)img.width * img.height; ++k)
          if (memcmp(&back.rgba[4 * k], &rgb[3 * k], 3) != 0) {
            fprintf(stderr, "PNG writer round trip differs for %s\n", p.c_str());
            return 3;
          }
        remove(tmp.c_str());
      } else {
        ++bad;
      }
    } else if (p.size() > 4 && p.substr(p.size() - 4) == ".jpg") {
      rta::Image img;
      (rta::load_image_rgba8(p, img, err) ? ok : bad)++;
    } else if (p.size() > 4 && p.substr(p.size() - 4) == ".obj") {
      rta::ObjModel m;
      (rta::load_obj(p, m, err) ? ok : bad)++;
    } else if (p.size() > 5 && p.substr(p.size() - 5) == ".json") {
      rta::SceneFile sf;
      if (rta::load_scene_file(p, sf, err)) {
        rta::Renderer r;
        r.init(-1, 8, 8);
        if (rta::instantiate_scene(sf, r, err) >= 0) {
          const rt_scene_desc &d = r.scene();
          uint32_t id = r.add_blas_instance(r.cube_blas_index, rta::mat4(1.f), r.default_material);
          r.set_blas_instance_transform(id, rta::mat4(2.f));
          r.remove_blas_instance(id);
          r.remove_blas_instance(0);
          (void)r.scene();
          ok += d.instance_count > 0;
        } else bad++;
      } else bad++;
    } else {
      rta::Renderer r;
      r.init(-1, 8, 8);
      (rta::import_gltf(p, r, rta::mat4(1.f), err) >= 0 ? ok : bad)++;
    }
  }
  printf("ok %d bad %d\n", ok, bad);
  return 0;
}
