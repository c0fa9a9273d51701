// cli_main.cpp — `renderer <scene>.json -hip [-o out.png]`: the reference's
// command line (README.md:21-50) with a `-hip` renderer beside `-gpu` / `-cpu`.
//
//   renderer scenes/TestScene.json -hip -o cornell.png
//
// The scene file is loaded by scene_file.cpp, assembled through rta::Renderer
// (the hlx::Renderer mirror) and rendered by librt_hip.so in ONE launch of
// floor(sqrt(samples_per_pixel))^2 stratified samples, as Renderer.cpp:619-648
// dispatches it.  The image is tone-mapped like the reference's presentation
// pass (Assets/Shaders/FullScreen.slang:38-55) and written as an 8-bit PNG;
// `--raw file` additionally dumps the linear rgba32f accumulation image.
//
// Beyond the reference's switches: `--rng wang|pcg` (the reference's sequential Wang hash, default, or
// the keyed PCG RNG), `--frames F` (F progressive frames of the file's samples per pixel in one rt_render_frames
// call — the reference's way to high sample counts), `--kernel K`, and `--gpus N` / `--devices a,b,..`: N ranks, one
// host thread and one rt_ctx each (rt_mgpu_*, rt_multi.cpp), interleaved 8-row groups, the image all-gathered over
// RCCL; the image is the one-GPU image bit for bit.  `--devices 0,0` rehearses two ranks on one GPU (peer copies).
//
// `-cpu` and `-gpu` are recognised and refused: the Vulkan renderer is out of
// scope, and the CPU restatement of this repository (oracle/) is test
// infrastructure, never a product path.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "image_decode.h"
#include "renderer.h"
#include "scene_file.h"

namespace {

// fragMain + agxToneMap, FullScreen.slang:38-55; byte conversion as Image.hpp:62-66
uint8_t tone_map_byte(float c) {
  c = c > 0.f ? c : 0.f; // max(color, 0)
  const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
  float m = (c * (a * c + b)) / (c * (cc * c + d) + e);
  if (!(m > 0.f))
    return 0;
  if (m >= 1.f)
    return 255;
  return (uint8_t)(m * 255.f);
}

int usage(const char *argv0) {
  fprintf(stderr,
          "usage: %s <scene.json> -hip [-o out.png] [--raw out.f32] [--device N | --gpus N | --devices a,b,..] [--width W] "
          "[--spp S] [--depth D] [--frames F] [--rng wang|pcg] [--kernel 0|8|9] [--sky white] [--rcp]\n"
          "  exactly one renderer must be chosen; this build provides -hip (MI355X, librt_hip.so)\n",
          argv0);
  return 2;
}

} // namespace

int main(int argc, char **argv) {
  std::string scene_path, out_png = "image.png", out_raw;
  bool hip = false;
  int device = 0;
  long width = 0, spp = 0, depth = -1, frames = 1, kernel = -1;
  bool rcp = false, pcg = false, sky_white = false;
  std::vector<int> devices; // --gpus / --devices: the ranks of the multi-GPU driver
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto need = [&](const char *what) -> const char * {
      if (i + 1 >= argc) {
        fprintf(stderr, "%s needs a value\n", what);
        exit(2);
      }
      return argv[++i];
    };
    if (a == "-hip") hip = true;
    else if (a == "-cpu") {
      fprintf(stderr, "-cpu: this build has no CPU renderer (its CPU restatement, oracle/, only checks the GPU "
                      "path in the test-suite); use -hip\n");
      return 2;
    } else if (a == "-gpu") {
      fprintf(stderr, "-gpu: the reference's Vulkan renderer is not part of this build; use -hip\n");
      return 2;
    } else if (a == "-o") out_png = need("-o");
    else if (a == "--raw") out_raw = need("--raw");
    else if (a == "--device") device = atoi(need("--device"));
    else if (a == "--width") width = atol(need("--width"));
    else if (a == "--spp") spp = atol(need("--spp"));
    else if (a == "--depth") depth = atol(need("--depth"));
    else if (a == "--rcp") rcp = true;
    else if (a == "--frames") frames = atol(need("--frames"));
    else if (a == "--kernel") kernel = atol(need("--kernel"));
    else if (a == "--rng") {
      std::string v = need("--rng");
      if (v != "wang" && v != "pcg") return usage(argv[0]);
      pcg = v == "pcg";
    } else if (a == "--sky") {
      if (std::string(need("--sky")) != "white") return usage(argv[0]);
      sky_white = true; // RT_RENDER_SKY_WHITE_HORIZON: the sky of the revision behind the reference's Screenshot3.png
    } else if (a == "--gpus") {
      long n = atol(need("--gpus"));
      if (n < 1 || n > 64) return usage(argv[0]);
      devices.clear();
      for (long k = 0; k < n; ++k) devices.push_back((int)k);
    } else if (a == "--devices") {
      devices.clear();
      for (const char *p = need("--devices"); *p;) {
        devices.push_back(atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
      if (devices.empty()) return usage(argv[0]);
    }
    else if (!a.empty() && a[0] == '-') return usage(argv[0]);
    else if (scene_path.empty()) scene_path = a;
    else return usage(argv[0]);
  }
  if (scene_path.empty() || !hip || frames < 1 || (kernel >= 0 && kernel != 0 && kernel != 8 && kernel != 9))
    return usage(argv[0]);
  const bool multi = devices.size() > 1; // (--gpus 1 is the single-context path, bit-equal by construction)
  if (devices.size() == 1)
    device = devices[0];

  std::string err;
  rta::SceneFile scene;
  if (!rta::load_scene_file(scene_path, scene, err)) {
    fprintf(stderr, "error: %s\n", err.c_str());
    return 1;
  }
  if (width > 0) { // keep the file's aspect ratio
    double aspect = (double)scene.screen_width / (double)scene.screen_height;
    scene.screen_width = (uint32_t)width;
    int h = (int)((double)width / aspect);
    scene.screen_height = (uint32_t)(h < 1 ? 1 : h);
  }
  if (spp > 0)
    scene.samples_per_pixel = (uint32_t)spp;
  if (depth >= 0)
    scene.max_depth = (uint32_t)depth;

  rta::Renderer r;
  if (r.init(multi ? -1 : device, scene.screen_width, scene.screen_height) != RT_OK) { // multi: the scene is built once on the host
    fprintf(stderr, "error: %s\n", r.last_error().c_str());
    return 1;
  }
  auto t0 = std::chrono::steady_clock::now();
  int64_t tris = rta::instantiate_scene(scene, r, err);
  if (tris < 0) {
    fprintf(stderr, "error: %s\n", err.c_str());
    return 1;
  }
  r.samples_per_pixel = scene.samples_per_pixel;
  r.max_depth = scene.max_depth;
  r.aabb_mode = rcp ? RT_AABB_RCP : RT_AABB_DIV;
  r.rng_mode = pcg ? RT_RNG_PCG : RT_RNG_WANG;
  r.render_flags = sky_white ? RT_RENDER_SKY_WHITE_HORIZON : 0u;
  auto t1 = std::chrono::steady_clock::now();
  std::vector<float> img((size_t)scene.screen_width * scene.screen_height * 4);
  rt_stats st{};
  rt_mgpu *mg = nullptr;
  if (!multi) {
    if (kernel >= 0 && rt_set_kernel_variant(r.context(), (uint32_t)kernel) != RT_OK) {
      fprintf(stderr, "error: %s\n", rt_last_error_string());
      return 1;
    }
    if (r.render_frames(scene.camera, (uint32_t)frames) != RT_OK || r.read_image(img.data()) != RT_OK) {
      fprintf(stderr, "error: %s\n", r.last_error().c_str());
      return 1;
    }
    rt_get_stats(r.context(), &st);
  } else {
    // one rank per entry of `devices`: the host arrays of the renderer above go to every rank, rank r renders the 8-row
    // groups r, r + n, ..., the groups are all-gathered, and rank 0's (complete) image is read back
    rt_render_params pc;
    memset(&pc, 0, sizeof pc);
    pc.image_width = scene.screen_width, pc.image_height = scene.screen_height;
    pc.sqrt_spp = (uint32_t)sqrtf((float)scene.samples_per_pixel); // Renderer.cpp:625 (truncating)
    if (pc.sqrt_spp == 0)
      pc.sqrt_spp = 1;
    pc.max_depth = scene.max_depth, pc.rng_mode = r.rng_mode, pc.aabb_mode = r.aabb_mode, pc.flags = r.render_flags;
    const rt_camera_uniforms cu = r.camera_uniforms(scene.camera);
    int rc = rt_mgpu_create(devices.data(), (uint32_t)devices.size(), RT_MGPU_EXCHANGE_AUTO, &mg);
    if (rc == RT_OK) rc = rt_mgpu_upload_scene(mg, &r.scene());
    if (rc == RT_OK) rc = rt_mgpu_set_camera(mg, &cu);
    if (rc == RT_OK && kernel >= 0) rc = rt_mgpu_set_kernel_variant(mg, (uint32_t)kernel);
    if (rc == RT_OK) rc = rt_mgpu_resize(mg, pc.image_width, pc.image_height);
    if (rc == RT_OK) rc = rt_mgpu_render_frames(mg, &pc, (uint32_t)frames);
    if (rc == RT_OK) rc = rt_mgpu_read_accum(mg, 0, img.data());
    if (rc != RT_OK) {
      fprintf(stderr, "error: %s\n", rt_last_error_string());
      if (mg) rt_mgpu_destroy(mg);
      return 1;
    }
    double slowest = 0.0;
    for (uint32_t k = 0; k < devices.size(); ++k) {
      rt_stats sk{};
      rt_mgpu_get_stats(mg, k, &sk);
      rt_mgpu_share sh;
      rt_mgpu_partition(pc.image_height, (uint32_t)devices.size(), k, &sh);
      fprintf(stderr, "rank %u (device %d): %u rows in %u groups, kernel %u, %.3f ms, %llu rays\n", k, devices[k], sh.rows, sh.groups,
              sk.last_kernel, sk.last_kernel_ms, (unsigned long long)sk.rays_traced);
      st.rays_traced += sk.rays_traced;
      slowest = sk.last_kernel_ms > slowest ? sk.last_kernel_ms : slowest;
    }
    st.last_kernel_ms = slowest;
    fprintf(stderr, "%zu ranks, image exchange: %s\n", devices.size(), rt_mgpu_uses_rccl(mg) ? "ncclAllGather (RCCL)" : "peer copies");
  }
  auto t2 = std::chrono::steady_clock::now();
  uint32_t sq = (uint32_t)sqrt((double)scene.samples_per_pixel);
  fprintf(stderr, "%s: %lld triangles, %ux%u, %ld frame(s) of %u spp (%ux%u strata), depth %u, %s RNG\n", scene_path.c_str(),
          (long long)tris, scene.screen_width, scene.screen_height, frames, sq * sq, sq, sq, scene.max_depth, pcg ? "keyed PCG" : "Wang");
  fprintf(stderr, "scene build %.3f s, render + read-back %.3f s (kernel %.3f ms, %.1f Mrays/s)\n",
          std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
          st.last_kernel_ms, st.last_kernel_ms > 0 ? (double)st.rays_traced / st.last_kernel_ms / 1e3 : 0.0);

  if (!out_raw.empty()) {
    FILE *f = fopen(out_raw.c_str(), "wb");
    if (!f || fwrite(img.data(), sizeof(float), img.size(), f) != img.size() || fclose(f) != 0) {
      fprintf(stderr, "error: cannot write %s\n", out_raw.c_str());
      return 1;
    }
  }
  std::vector<uint8_t> rgb((size_t)scene.screen_width * scene.screen_height * 3);
  for (size_t p = 0; p < (size_t)scene.screen_width * scene.screen_height; ++p)
    for (int c = 0; c < 3; ++c)
      rgb[p * 3 + c] = tone_map_byte(img[p * 4 + c]);
  if (!rta::write_png_rgb8(out_png, scene.screen_width, scene.screen_height, rgb.data())) {
    fprintf(stderr, "error: cannot write %s\n", out_png.c_str());
    return 1;
  }
  fprintf(stderr, "wrote %s\n", out_png.c_str());
  if (mg)
    rt_mgpu_destroy(mg);
  r.shutdown();
  return 0;
}
