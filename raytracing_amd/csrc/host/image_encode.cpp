// image_encode.cpp — 8-bit RGB PNG writer of the command line (the reference writes its output through
// stb_image_write, vendor/stb_image_write.h; any PNG reader returns the same pixels from either file).
// Row filters chosen by the minimum-sum-of-absolute-residuals heuristic, zlib stream with one fixed-Huffman DEFLATE
// block and greedy LZ77 matches (RFC 1950 / 1951).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "image_decode.h"

namespace rta {
namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k)
        c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i)
    crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24)), v.push_back((uint8_t)(x >> 16)), v.push_back((uint8_t)(x >> 8)),
      v.push_back((uint8_t)x);
}

void chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data) {
  put_be32(png, (uint32_t)data.size());
  size_t start = png.size();
  png.insert(png.end(), type, type + 4);
  png.insert(png.end(), data.begin(), data.end());
  uint32_t crc = crc32_update(0xFFFFFFFFu, png.data() + start, png.size() - start) ^ 0xFFFFFFFFu;
  put_be32(png, crc);
}

// DEFLATE with the fixed Huffman code (RFC 1951 3.2.6) and greedy LZ77 matches found through a hash of three bytes:
// small, dependency-free, and every PNG reader inflates it.
struct BitWriter {
  std::vector<uint8_t> &out;
  uint32_t acc = 0;
  int n = 0;
  void bits(uint32_t v, int count) { // least significant bit first
    acc |= v << n;
    n += count;
    while (n >= 8) {
      out.push_back((uint8_t)acc);
      acc >>= 8;
      n -= 8;
    }
  }
  void huff(uint32_t code, int len) { // Huffman codes are sent most significant bit first
    uint32_t r = 0;
    for (int i = 0; i < len; ++i)
      r |= ((code >> i) & 1u) << (len - 1 - i);
    bits(r, len);
  }
  void flush() {
    if (n > 0)
      out.push_back((uint8_t)acc);
    acc = 0, n = 0;
  }
};

void fixed_literal(BitWriter &w, uint32_t sym) { // literal / length alphabet of the fixed code
  if (sym < 144) w.huff(0x30 + sym, 8);
  else if (sym < 256) w.huff(0x190 + (sym - 144), 9);
  else if (sym < 280) w.huff(sym - 256, 7);
  else w.huff(0xC0 + (sym - 280), 8);
}

void deflate_fixed(const std::vector<uint8_t> &in, std::vector<uint8_t> &out) {
  static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  BitWriter w{out};
  w.bits(1, 1), w.bits(1, 2); // one final block, fixed Huffman
  const size_t n = in.size();
  std::vector<int64_t> head(1u << 15, -1);
  auto hash3 = [&](size_t i) { return ((uint32_t)in[i] * 0x9E37u ^ (uint32_t)in[i + 1] * 0x79B9u ^ (uint32_t)in[i + 2] * 0x7F4Au) & 0x7FFFu; };
  size_t i = 0;
  while (i < n) {
    size_t best_len = 0, best_dist = 0;
    if (i + 3 <= n) {
      const uint32_t h = hash3(i);
      const int64_t cand = head[h];
      head[h] = (int64_t)i;
      if (cand >= 0 && i - (size_t)cand <= 32768) {
        size_t l = 0;
        const size_t maxl = n - i < 258 ? n - i : 258;
        while (l < maxl && in[(size_t)cand + l] == in[i + l])
          ++l;
        if (l >= 3)
          best_len = l, best_dist = i - (size_t)cand;
      }
    }
    if (best_len) {
      int lc = 28;
      while (len_base[lc] > best_len)
        --lc;
      fixed_literal(w, 257 + (uint32_t)lc);
      w.bits((uint32_t)(best_len - len_base[lc]), len_extra[lc]);
      int dc = 29;
      while (dist_base[dc] > best_dist)
        --dc;
      w.huff((uint32_t)dc, 5);
      w.bits((uint32_t)(best_dist - dist_base[dc]), dist_extra[dc]);
      for (size_t k = 1; k < best_len && i + k + 3 <= n; ++k) // keep the hash chain heads fresh inside the match
        head[hash3(i + k)] = (int64_t)(i + k);
      i += best_len;
    } else {
      fixed_literal(w, in[i]);
      ++i;
    }
  }
  fixed_literal(w, 256);
  w.flush();
}

} // namespace

bool write_png_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb) {
  if (w == 0 || h == 0 || !rgb)
    return false;
  // scanlines with the filter that minimises the sum of absolute residuals (the usual PNG heuristic)
  const size_t stride = (size_t)w * 3;
  std::vector<uint8_t> raw;
  raw.reserve((size_t)h * (stride + 1));
  std::vector<uint8_t> cand[5];
  for (auto &c : cand)
    c.resize(stride);
  const std::vector<uint8_t> zero(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t *cur = rgb + (size_t)y * stride;
    const uint8_t *up = y ? cur - stride : zero.data();
    uint64_t cost[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= 3 ? cur[i - 3] : 0, b = up[i], c = i >= 3 ? up[i - 3] : 0;
      const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
      const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      const int pred[5] = {0, a, b, (a + b) >> 1, paeth};
      for (int f = 0; f < 5; ++f) {
        const uint8_t v = (uint8_t)(cur[i] - pred[f]);
        cand[f][i] = v;
        cost[f] += v < 128 ? v : 256 - v;
      }
    }
    int best = 0;
    for (int f = 1; f < 5; ++f)
      if (cost[f] < cost[best])
        best = f;
    raw.push_back((uint8_t)best);
    raw.insert(raw.end(), cand[best].begin(), cand[best].end());
  }
  std::vector<uint8_t> z;
  z.push_back(0x78), z.push_back(0x01); // zlib header
  deflate_fixed(raw, z);
  uint32_t a = 1, b = 0; // adler32 of the uncompressed data
  for (size_t i = 0; i < raw.size(); ++i) {
    a = (a + raw[i]) % 65521u;
    b = (b + a) % 65521u;
  }
  put_be32(z, (b << 16) | a);

  std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, w), put_be32(ihdr, h);
  ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
  chunk(png, "IHDR", ihdr);
  chunk(png, "IDAT", z);
  chunk(png, "IEND", {});
  FILE *f = fopen(path.c_str(), "wb");
  if (!f)
    return false;
  bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
  ok = fclose(f) == 0 && ok;
  return ok;
}

} // namespace rta
