// image_decode.cpp — image files to RGBA8 for Renderer::add_lambert_material(path), src/Renderer.cpp:705-716.
//
// The reference calls stb_image (`stbi_load(path, &w, &h, &comp, 4)`, no vertical flip) — a third-party header
// (vendor/stb_image.h v2.x) that is not copied here.  This file restates the published formats it reads for the
// two lossless ones, where every conforming decoder must return the same bytes:
//   * PNG (ISO/IEC 15948): all colour types, bit depths 1-16, palette + tRNS, colour-key tRNS, Adam7 interlace,
//     zlib/DEFLATE (RFC 1950/1951) with stored, fixed and dynamic Huffman blocks.  Conversion to four 8-bit channels
//     as stb_image does it: grey -> (g,g,g,255), grey+alpha -> (g,g,g,a), RGB -> (r,g,b,255); grey of depth 1/2/4 is
//     scaled by 255/(2^depth-1); 16-bit samples keep their high byte; a tRNS colour key gives alpha 0.
//   * binary PNM (P5 / P6), 8 or 16 bit.
// JPEG is lossy — the pixels depend on the decoder's IDCT, upsampling and colour arithmetic — and lives in
// image_jpeg.cpp, which restates stb_image's published integer arithmetic for those stages (baseline / extended
// sequential and progressive).
#include "image_decode.h"

#include <stdio.h>
#include <string.h>

namespace rta {
namespace {

// ---- DEFLATE (RFC 1951) ---------------------------------------------------------------------------------------
struct BitReader {
  const uint8_t *p, *end;
  uint32_t acc = 0;
  int nbits = 0;
  bool overrun = false;
  uint32_t bits(int n) {
    while (nbits < n) {
      uint32_t byte = 0;
      if (p < end)
        byte = *p++;
      else
        overrun = true;
      acc |= byte << nbits;
      nbits += 8;
    }
    uint32_t v = acc & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1u));
    acc >>= n;
    nbits -= n;
    return v;
  }
  void align_byte() {
    acc >>= (nbits & 7);
    nbits -= (nbits & 7);
  }
};

struct Huffman { // canonical code, decoded bit by bit (count / symbol tables, RFC 1951 3.2.2)
  uint16_t count[16];
  uint16_t symbol[288];
  bool build(const uint8_t *lengths, int n) {
    memset(count, 0, sizeof count);
    for (int i = 0; i < n; ++i)
      count[lengths[i]]++;
    count[0] = 0;
    int left = 1;
    for (int len = 1; len < 16; ++len) {
      left <<= 1;
      left -= count[len];
      if (left < 0)
        return false; // over-subscribed
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len)
      offs[len + 1] = (uint16_t)(offs[len] + count[len]);
    for (int i = 0; i < n; ++i)
      if (lengths[i])
        symbol[offs[lengths[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(BitReader &br) const {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
      code |= (int)br.bits(1);
      int c = count[len];
      if (code - c < first)
        return symbol[index + (code - first)];
      index += c;
      first += c;
      first <<= 1;
      code <<= 1;
    }
    return -1;
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_codes(BitReader &br, const Huffman &lit, const Huffman &dist, std::vector<uint8_t> &out, size_t limit) {
  for (;;) {
    int sym = lit.decode(br);
    if (sym < 0 || br.overrun)
      return false;
    if (sym < 256) {
      if (out.size() >= limit)
        return false;
      out.push_back((uint8_t)sym);
    } else if (sym == 256) {
      return true;
    } else {
      sym -= 257;
      if (sym >= 29)
        return false;
      size_t len = kLenBase[sym] + br.bits(kLenExtra[sym]);
      int ds = dist.decode(br);
      if (ds < 0 || ds >= 30)
        return false;
      size_t d = kDistBase[ds] + br.bits(kDistExtra[ds]);
      if (d > out.size() || out.size() + len > limit)
        return false;
      size_t from = out.size() - d;
      for (size_t i = 0; i < len; ++i)
        out.push_back(out[from + i]);
    }
  }
}

// zlib stream (RFC 1950) -> bytes; at most `limit` bytes are produced
bool zlib_inflate(const uint8_t *data, size_t n, std::vector<uint8_t> &out, size_t limit) {
  if (n < 2 || (data[0] & 0x0F) != 8 || ((data[0] << 8) | data[1]) % 31 != 0 || (data[1] & 0x20))
    return false;
  BitReader br{data + 2, data + n};
  out.clear();
  out.reserve(limit);
  bool last = false;
  while (!last) {
    last = br.bits(1) != 0;
    uint32_t type = br.bits(2);
    if (type == 0) {
      br.align_byte();
      uint32_t len = br.bits(16), nlen = br.bits(16);
      if ((len ^ 0xFFFFu) != nlen || br.overrun || out.size() + len > limit)
        return false;
      for (uint32_t i = 0; i < len; ++i)
        out.push_back((uint8_t)br.bits(8));
      if (br.overrun)
        return false;
    } else if (type == 1) {
      uint8_t l[288], d[30];
      for (int i = 0; i < 144; ++i) l[i] = 8;
      for (int i = 144; i < 256; ++i) l[i] = 9;
      for (int i = 256; i < 280; ++i) l[i] = 7;
      for (int i = 280; i < 288; ++i) l[i] = 8;
      for (int i = 0; i < 30; ++i) d[i] = 5;
      Huffman lit, dist;
      lit.build(l, 288), dist.build(d, 30);
      if (!inflate_codes(br, lit, dist, out, limit))
        return false;
    } else if (type == 2) {
      int hlit = (int)br.bits(5) + 257, hdist = (int)br.bits(5) + 1, hclen = (int)br.bits(4) + 4;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0};
      for (int i = 0; i < hclen; ++i)
        cl[order[i]] = (uint8_t)br.bits(3);
      Huffman clen;
      if (hlit > 286 || hdist > 30 || !clen.build(cl, 19))
        return false;
      uint8_t lengths[286 + 30];
      int i = 0;
      while (i < hlit + hdist) {
        int sym = clen.decode(br);
        if (sym < 0 || br.overrun)
          return false;
        if (sym < 16) {
          lengths[i++] = (uint8_t)sym;
        } else {
          int rep, val = 0;
          if (sym == 16) {
            if (i == 0)
              return false;
            val = lengths[i - 1];
            rep = 3 + (int)br.bits(2);
          } else if (sym == 17) {
            rep = 3 + (int)br.bits(3);
          } else {
            rep = 11 + (int)br.bits(7);
          }
          if (i + rep > hlit + hdist)
            return false;
          while (rep--)
            lengths[i++] = (uint8_t)val;
        }
      }
      Huffman lit, dist;
      if (!lit.build(lengths, hlit) || !dist.build(lengths + hlit, hdist))
        return false;
      if (!inflate_codes(br, lit, dist, out, limit))
        return false;
    } else {
      return false;
    }
  }
  return true;
}

// ---- PNG -------------------------------------------------------------------------------------------------------
uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
  int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undoes the scanline filters of one (sub-)image of w x h pixels in place; `raw` has h * (1 + stride) bytes.
// Returns the h * stride filtered-out bytes.
bool unfilter(const uint8_t *raw, size_t raw_len, uint32_t w, uint32_t h, int channels, int depth,
              std::vector<uint8_t> &pix) {
  const size_t stride = ((size_t)w * channels * depth + 7) / 8;
  const size_t bpp = (size_t)(channels * depth + 7) / 8;
  if (raw_len < (stride + 1) * h)
    return false;
  pix.assign(stride * h, 0);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t *in = raw + (stride + 1) * y;
    const int filter = in[0];
    ++in;
    uint8_t *cur = pix.data() + stride * y;
    const uint8_t *up = y ? cur - stride : nullptr;
    if (filter > 4)
      return false;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int v = in[i];
      switch (filter) {
      case 1: v += a; break;
      case 2: v += b; break;
      case 3: v += (a + b) >> 1; break;
      case 4: v += paeth(a, b, c); break;
      default: break;
      }
      cur[i] = (uint8_t)v;
    }
  }
  return true;
}

struct PngInfo {
  uint32_t w = 0, h = 0;
  int depth = 0, color = 0, interlace = 0, channels = 0;
  uint8_t palette[256][4];
  int pal_len = 0;
  bool has_key = false;
  uint16_t key[3] = {0, 0, 0};
};

// one decoded sub-image (packed samples) -> RGBA8 pixels scattered into the full image
void emit_pixels(const PngInfo &pi, const std::vector<uint8_t> &pix, uint32_t sw, uint32_t sh, uint32_t x0, uint32_t y0,
                 uint32_t dx, uint32_t dy, uint8_t *rgba) {
  const size_t stride = ((size_t)sw * pi.channels * pi.depth + 7) / 8;
  static const int scale_of_depth[9] = {0, 0xFF, 0x55, 0, 0x11, 0, 0, 0, 0x01};
  for (uint32_t y = 0; y < sh; ++y) {
    const uint8_t *row = pix.data() + stride * y;
    for (uint32_t x = 0; x < sw; ++x) {
      uint16_t s[4] = {0, 0, 0, 0}; // samples at file precision
      for (int c = 0; c < pi.channels; ++c) {
        const size_t k = (size_t)x * pi.channels + c;
        if (pi.depth == 16)
          s[c] = (uint16_t)((row[2 * k] << 8) | row[2 * k + 1]);
        else if (pi.depth == 8)
          s[c] = row[k];
        else { // 1, 2, 4 bits, most significant first
          const size_t bit = k * pi.depth;
          s[c] = (uint16_t)((row[bit >> 3] >> (8 - pi.depth - (bit & 7))) & ((1 << pi.depth) - 1));
        }
      }
      uint8_t *o = rgba + ((size_t)(y0 + y * dy) * pi.w + (x0 + x * dx)) * 4;
      if (pi.color == 3) {
        const int idx = s[0];
        if (idx < pi.pal_len)
          memcpy(o, pi.palette[idx], 4);
        else
          o[0] = o[1] = o[2] = 0, o[3] = 255;
        continue;
      }
      // colour key is compared at file precision (16 bit) or on the scaled 8-bit value, as stb_image does
      auto to8 = [&](uint16_t v) -> uint8_t {
        return pi.depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)(pi.depth < 8 ? v * scale_of_depth[pi.depth] : v);
      };
      bool keyed = false;
      if (pi.has_key) {
        const int nc = pi.color == 0 ? 1 : 3;
        keyed = true;
        for (int c = 0; c < nc; ++c) {
          if (pi.depth == 16)
            keyed = keyed && s[c] == pi.key[c];
          else
            keyed = keyed && to8(s[c]) == (uint8_t)((pi.key[c] & 255) * (pi.depth < 8 ? scale_of_depth[pi.depth] : 1));
        }
      }
      switch (pi.color) {
      case 0: o[0] = o[1] = o[2] = to8(s[0]), o[3] = keyed ? 0 : 255; break;
      case 4: o[0] = o[1] = o[2] = to8(s[0]), o[3] = to8(s[1]); break;
      case 2: o[0] = to8(s[0]), o[1] = to8(s[1]), o[2] = to8(s[2]), o[3] = keyed ? 0 : 255; break;
      default: o[0] = to8(s[0]), o[1] = to8(s[1]), o[2] = to8(s[2]), o[3] = to8(s[3]); break;
      }
    }
  }
}

bool decode_png(const std::vector<uint8_t> &f, Image &img, std::string &err) {
  PngInfo pi;
  std::vector<uint8_t> idat;
  size_t pos = 8;
  bool have_ihdr = false, done = false;
  while (!done) {
    if (pos + 8 > f.size()) {
      err = "PNG: truncated file";
      return false;
    }
    const uint32_t len = be32(&f[pos]);
    const uint8_t *type = &f[pos + 4], *data = &f[pos + 8];
    if ((size_t)len > f.size() - pos - 8 || f.size() - pos - 8 - len < 4) {
      err = "PNG: truncated chunk";
      return false;
    }
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13 || have_ihdr) {
        err = "PNG: bad IHDR";
        return false;
      }
      pi.w = be32(data), pi.h = be32(data + 4);
      pi.depth = data[8], pi.color = data[9], pi.interlace = data[12];
      const bool depth_ok = pi.depth == 1 || pi.depth == 2 || pi.depth == 4 || pi.depth == 8 || pi.depth == 16;
      const bool color_ok = pi.color == 0 || pi.color == 2 || pi.color == 3 || pi.color == 4 || pi.color == 6;
      if (!depth_ok || !color_ok || data[10] != 0 || data[11] != 0 || pi.interlace > 1 || pi.w == 0 || pi.h == 0 ||
          (pi.color == 3 && pi.depth == 16) || ((pi.color == 2 || pi.color == 4 || pi.color == 6) && pi.depth < 8)) {
        err = "PNG: unsupported header";
        return false;
      }
      if (pi.w > (1u << 24) || pi.h > (1u << 24) || (uint64_t)pi.w * pi.h > (1ull << 26)) { // stb_image: 2^24 a side; here also 64 Mpixel
        err = "PNG: image too large";
        return false;
      }
      pi.channels = pi.color == 0 ? 1 : pi.color == 2 ? 3 : pi.color == 3 ? 1 : pi.color == 4 ? 2 : 4;
      have_ihdr = true;
    } else if (!have_ihdr) {
      err = "PNG: first chunk is not IHDR";
      return false;
    } else if (!memcmp(type, "PLTE", 4)) {
      if (len > 256 * 3 || len % 3) {
        err = "PNG: bad PLTE";
        return false;
      }
      pi.pal_len = (int)(len / 3);
      for (int i = 0; i < pi.pal_len; ++i)
        pi.palette[i][0] = data[3 * i], pi.palette[i][1] = data[3 * i + 1], pi.palette[i][2] = data[3 * i + 2],
        pi.palette[i][3] = 255;
    } else if (!memcmp(type, "tRNS", 4)) {
      if (pi.color == 3) {
        if ((int)len > pi.pal_len) {
          err = "PNG: bad tRNS";
          return false;
        }
        for (uint32_t i = 0; i < len; ++i)
          pi.palette[i][3] = data[i];
      } else if (pi.color == 0 || pi.color == 2) {
        if (len != (pi.color == 0 ? 2u : 6u)) {
          err = "PNG: bad tRNS";
          return false;
        }
        pi.has_key = true;
        for (uint32_t i = 0; i < len / 2; ++i)
          pi.key[i] = (uint16_t)((data[2 * i] << 8) | data[2 * i + 1]);
      } else {
        err = "PNG: tRNS with an alpha channel";
        return false;
      }
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      done = true;
    } else if (!(type[0] & 0x20)) {
      err = "PNG: unknown critical chunk";
      return false;
    }
    pos += 12 + (size_t)len;
  }
  if (pi.color == 3 && pi.pal_len == 0) {
    err = "PNG: palette image without PLTE";
    return false;
  }
  // expected size of the filtered data
  static const uint32_t xorig[7] = {0, 4, 0, 2, 0, 1, 0}, yorig[7] = {0, 0, 4, 0, 2, 0, 1};
  static const uint32_t xspc[7] = {8, 8, 4, 4, 2, 2, 1}, yspc[7] = {8, 8, 8, 4, 4, 2, 2};
  auto sub_bytes = [&](uint32_t sw, uint32_t sh) -> size_t {
    return sw && sh ? (((size_t)sw * pi.channels * pi.depth + 7) / 8 + 1) * sh : 0;
  };
  size_t expect = 0;
  if (pi.interlace)
    for (int p = 0; p < 7; ++p)
      expect += sub_bytes((pi.w - xorig[p] + xspc[p] - 1) / xspc[p], (pi.h - yorig[p] + yspc[p] - 1) / yspc[p]);
  else
    expect = sub_bytes(pi.w, pi.h);
  std::vector<uint8_t> raw;
  if (!zlib_inflate(idat.data(), idat.size(), raw, expect + 64) || raw.size() < expect) {
    err = "PNG: corrupt or truncated image data";
    return false;
  }
  img.width = pi.w, img.height = pi.h;
  img.rgba.assign((size_t)pi.w * pi.h * 4, 0);
  std::vector<uint8_t> pix;
  if (!pi.interlace) {
    if (!unfilter(raw.data(), raw.size(), pi.w, pi.h, pi.channels, pi.depth, pix)) {
      err = "PNG: bad scanline filter";
      return false;
    }
    emit_pixels(pi, pix, pi.w, pi.h, 0, 0, 1, 1, img.rgba.data());
  } else {
    size_t off = 0;
    for (int p = 0; p < 7; ++p) {
      const uint32_t sw = (pi.w - xorig[p] + xspc[p] - 1) / xspc[p], sh = (pi.h - yorig[p] + yspc[p] - 1) / yspc[p];
      if (!sw || !sh)
        continue;
      if (!unfilter(raw.data() + off, raw.size() - off, sw, sh, pi.channels, pi.depth, pix)) {
        err = "PNG: bad scanline filter";
        return false;
      }
      emit_pixels(pi, pix, sw, sh, xorig[p], yorig[p], xspc[p], yspc[p], img.rgba.data());
      off += sub_bytes(sw, sh);
    }
  }
  return true;
}

// ---- binary PNM (P5 grey, P6 RGB) --------------------------------------------------------------------------------
bool pnm_token(const std::vector<uint8_t> &f, size_t &pos, uint32_t &value) {
  for (;;) { // whitespace and '#' comments
    while (pos < f.size() && (f[pos] == ' ' || f[pos] == '\t' || f[pos] == '\n' || f[pos] == '\r' || f[pos] == '\f' || f[pos] == '\v'))
      ++pos;
    if (pos < f.size() && f[pos] == '#')
      while (pos < f.size() && f[pos] != '\n' && f[pos] != '\r')
        ++pos;
    else
      break;
  }
  if (pos >= f.size() || f[pos] < '0' || f[pos] > '9')
    return false;
  uint64_t v = 0;
  while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') {
    v = v * 10 + (f[pos++] - '0');
    if (v > 0x7FFFFFFFu)
      return false;
  }
  value = (uint32_t)v;
  return true;
}

bool decode_pnm(const std::vector<uint8_t> &f, Image &img, std::string &err) {
  const int channels = f[1] == '5' ? 1 : 3;
  size_t pos = 2;
  uint32_t w, h, maxv;
  if (!pnm_token(f, pos, w) || !pnm_token(f, pos, h) || !pnm_token(f, pos, maxv) || pos >= f.size() || w == 0 ||
      h == 0 || maxv == 0 || maxv > 65535 || (uint64_t)w * h > (1ull << 26)) {
    err = "PNM: bad header";
    return false;
  }
  ++pos; // the single whitespace byte behind maxval
  const size_t bytes = maxv > 255 ? 2 : 1, need = (size_t)w * h * channels * bytes;
  if (f.size() - pos < need) {
    err = "PNM: truncated image data";
    return false;
  }
  img.width = w, img.height = h;
  img.rgba.resize((size_t)w * h * 4);
  const uint8_t *p = &f[pos];
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    uint8_t s[3];
    for (int c = 0; c < channels; ++c)
      s[c] = p[(i * channels + c) * bytes]; // 16 bit, big endian: the high byte (stb_image's 16 -> 8 reduction)
    uint8_t *o = &img.rgba[i * 4];
    if (channels == 1)
      o[0] = o[1] = o[2] = s[0];
    else
      o[0] = s[0], o[1] = s[1], o[2] = s[2];
    o[3] = 255;
  }
  return true;
}

} // namespace

bool decode_image_rgba8(const std::vector<uint8_t> &file, Image &out, std::string &err) {
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() >= 8 && !memcmp(file.data(), png_sig, 8))
    return decode_png(file, out, err);
  if (file.size() >= 3 && file[0] == 'P' && (file[1] == '5' || file[1] == '6'))
    return decode_pnm(file, out, err);
  if (file.size() >= 3 && file[0] == 0xFF && file[1] == 0xD8 && file[2] == 0xFF) { // image_jpeg.cpp
    return decode_jpeg(file, out, err);
  }
  err = "unsupported image format (PNG, JPEG and binary PNM are decoded)";
  return false;
}

bool load_image_rgba8(const std::string &path, Image &out, std::string &err) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) {
    err = "cannot open " + path;
    return false;
  }
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0)
    file.insert(file.end(), buf, buf + n);
  fclose(fp);
  if (!decode_image_rgba8(file, out, err)) {
    err = path + ": " + err;
    return false;
  }
  return true;
}

} // namespace rta
