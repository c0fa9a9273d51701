// image_decode.h — image files to RGBA8 for Renderer::add_lambert_material(path) (src/Renderer.cpp:705-716,
// which calls stbi_load(path, &w, &h, &comp, 4)).  PNG, binary PNM (image_decode.cpp) and JPEG (image_jpeg.cpp).
#ifndef RTA_IMAGE_DECODE_H
#define RTA_IMAGE_DECODE_H

#include <stdint.h>

#include <string>
#include <vector>

namespace rta {

struct Image {
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> rgba; // width * height * 4, rows top to bottom (no vertical flip, Renderer.cpp:707)
};

bool decode_image_rgba8(const std::vector<uint8_t> &file_bytes, Image &out, std::string &err);
bool load_image_rgba8(const std::string &path, Image &out, std::string &err);
bool decode_jpeg(const std::vector<uint8_t> &file_bytes, Image &out, std::string &err); // image_jpeg.cpp

// 8-bit RGB PNG (image_encode.cpp): rgb = width * height * 3 bytes, rows top to bottom
bool write_png_rgb8(const std::string &path, uint32_t width, uint32_t height, const uint8_t *rgb);

} // namespace rta
#endif
