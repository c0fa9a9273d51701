// image_jpeg.cpp — baseline / extended-sequential / progressive Huffman JPEG (ITU-T T.81, JFIF) to RGBA8 for
// Renderer::add_lambert_material(path) (src/Renderer.cpp:705-716: stbi_load(path, &w, &h, &comp, 4)) and glTF images
// (SceneGraph.cpp:42-166).
//
// JPEG is lossy: the entropy decoding is fixed by T.81, but the pixels depend on the decoder's inverse DCT, chroma
// upsampling and colour conversion.  The reference's decoder is stb_image (vendor/stb_image.h v2.x, not copied, not
// built, not run here), so the three arithmetic stages restate ITS published integer arithmetic, each stated where it
// is used: the "islow"-derived IDCT with 12-bit constants and its two roundings (stb_image.h:2426-2524; its SSE2 path
// is documented there as bit-identical to the scalar one), the centred 3:1 triangle upsampling for 2x1, 1x2 and 2x2
// subsampling and nearest neighbour for other ratios (:3449-3655), and the 20-bit fixed-point YCbCr -> RGB with the
// masked green term (:3657-3683).  PARITY UNPINNED: no run of stb_image is available to compare with; the tests check
// against libjpeg (Pillow) within the few levels two conforming decoders differ by, and exactly where the arithmetic
// leaves no freedom (flat blocks).
// Decoded: SOF0 / SOF1 / SOF2 (baseline, extended sequential, progressive), 8-bit samples, 1 component (grey -> g,g,g,255) or 3 (YCbCr, or RGB when an Adobe APP14 marker
// or the component ids 'R','G','B' say so), interleaved and non-interleaved scans, restart intervals, 8- and 16-bit
// quantisation tables.  Refused with a message: lossless / arithmetic-coded frames, 12-bit samples, 4 components
// (CMYK / YCCK).
#include "image_decode.h"

#include <string.h>

namespace rta {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable { // T.81 Annex C / F.2.2.3: codes of each length in increasing order
  bool defined = false;
  uint8_t count[17] = {0};
  uint8_t symbol[256] = {0};
  int32_t min_code[17], max_code[18], first_index[17];
  bool build() {
    int32_t code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      first_index[len] = k;
      min_code[len] = code;
      code += count[len];
      k += count[len];
      max_code[len] = count[len] ? code - 1 : -1;
      if (code > (1 << len))
        return false; // over-subscribed
      code <<= 1;
    }
    max_code[17] = 0x7FFFFFFF;
    return k <= 256;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int dc_pred = 0;
  uint32_t x = 0, y = 0;   // samples that carry image data
  uint32_t w2 = 0, h2 = 0; // allocated plane (whole MCUs)
  std::vector<uint8_t> plane;
  std::vector<short> coef; // progressive frames: every block's 64 coefficients in natural order, summed over the scans
};

struct EntropyReader { // the scan's bit stream: 0xFF00 is a stuffed 0xFF, any other marker ends the data (zeros follow)
  const uint8_t *p, *end;
  uint32_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 24) {
      uint32_t byte = 0;
      if (!hit_marker && p < end) {
        byte = *p;
        if (byte == 0xFF) {
          const uint32_t next = p + 1 < end ? p[1] : 0xD9;
          if (next == 0x00) {
            p += 2;
          } else {
            hit_marker = true; // leave p on the marker
            byte = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= byte << (24 - nbits);
      nbits += 8;
    }
  }
  int bit() {
    if (nbits < 1)
      fill();
    const int b = (int)(acc >> 31);
    acc <<= 1;
    --nbits;
    return b;
  }
  int receive(int n) { // n <= 16
    if (n == 0)
      return 0;
    if (nbits < n)
      fill();
    const int v = (int)(acc >> (32 - n));
    acc <<= n;
    nbits -= n;
    return v;
  }
  void reset() {
    acc = 0;
    nbits = 0;
    hit_marker = false;
  }
};

int decode_symbol(EntropyReader &br, const HuffTable &t) { // T.81 F.2.2.3 DECODE; -1: no such code
  int32_t code = 0;
  for (int len = 1; len <= 16; ++len) {
    code = (code << 1) | br.bit();
    if (t.max_code[len] >= 0 && code <= t.max_code[len] && code >= t.min_code[len])
      return t.symbol[t.first_index[len] + (code - t.min_code[len])];
  }
  return -1;
}

inline int extend(int v, int n) { // T.81 F.2.2.1 EXTEND
  return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
}

inline uint8_t clamp255(int x) { return x < 0 ? 0 : (x > 255 ? 255 : (uint8_t)x); }

// stb_image's inverse DCT (stb_image.h:2426-2524, "derived from jidctint -- DCT_ISLOW"): constants rounded to 12 bits,
// a column pass that keeps 2 extra bits (rounding +512 then >> 10; a column whose AC terms are all zero is d0 * 4),
// a row pass that removes 17 bits with rounding 65536 and adds the level shift 128 before the shift.
inline int f2f(double x) { return (int)(x * 4096 + 0.5); }
// (unsigned arithmetic: a crafted file can drive the intermediate sums past 31 bits, which must wrap like stb_image's
//  two's-complement ints do in practice instead of being undefined behaviour; for every valid file the values are the same)
struct Idct1D {
  int x0, x1, x2, x3, t0, t1, t2, t3;
  Idct1D(int s0_, int s1_, int s2_, int s3_, int s4_, int s5_, int s6_, int s7_) {
    typedef uint32_t U;
    static const U c0 = (U)f2f(0.5411961f), c1 = (U)f2f(-1.847759065f), c2 = (U)f2f(0.765366865f), c3 = (U)f2f(1.175875602f),
                   c4 = (U)f2f(0.298631336f), c5 = (U)f2f(2.053119869f), c6 = (U)f2f(3.072711026f), c7 = (U)f2f(1.501321110f),
                   c8 = (U)f2f(-0.899976223f), c9 = (U)f2f(-2.562915447f), c10 = (U)f2f(-1.961570560f), c11 = (U)f2f(-0.390180644f);
    const U s0 = (U)s0_, s1 = (U)s1_, s2 = (U)s2_, s3 = (U)s3_, s4 = (U)s4_, s5 = (U)s5_, s6 = (U)s6_, s7 = (U)s7_;
    U p1 = (s2 + s6) * c0;
    const U e2 = p1 + s6 * c1, e3 = p1 + s2 * c2;
    const U e0 = (s0 + s4) * 4096u, e1 = (s0 - s4) * 4096u;
    x0 = (int)(e0 + e3), x3 = (int)(e0 - e3), x1 = (int)(e1 + e2), x2 = (int)(e1 - e2);
    U o0 = s7, o1 = s5, o2 = s3, o3 = s1;
    U p3 = o0 + o2, p4 = o1 + o3, p2 = o1 + o2;
    p1 = o0 + o3;
    const U p5 = (p3 + p4) * c3;
    o0 *= c4, o1 *= c5, o2 *= c6, o3 *= c7;
    p1 = p5 + p1 * c8, p2 = p5 + p2 * c9, p3 *= c10, p4 *= c11;
    t3 = (int)(o3 + p1 + p4), t2 = (int)(o2 + p2 + p3), t1 = (int)(o1 + p2 + p4), t0 = (int)(o0 + p1 + p3);
  }
};
inline int wadd(int a, int b) { return (int)((uint32_t)a + (uint32_t)b); }
inline int wsub(int a, int b) { return (int)((uint32_t)a - (uint32_t)b); }

void idct_block(uint8_t *out, size_t stride, const short d[64]) {
  int val[64];
  for (int i = 0; i < 8; ++i) {
    const short *c = d + i;
    int *v = val + i;
    if (!c[8] && !c[16] && !c[24] && !c[32] && !c[40] && !c[48] && !c[56]) {
      const int dc = (int)((uint32_t)(int)c[0] * 4u);
      for (int k = 0; k < 8; ++k)
        v[8 * k] = dc;
    } else {
      Idct1D r(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
      const int x0 = wadd(r.x0, 512), x1 = wadd(r.x1, 512), x2 = wadd(r.x2, 512), x3 = wadd(r.x3, 512);
      v[0] = wadd(x0, r.t3) >> 10, v[56] = wsub(x0, r.t3) >> 10;
      v[8] = wadd(x1, r.t2) >> 10, v[48] = wsub(x1, r.t2) >> 10;
      v[16] = wadd(x2, r.t1) >> 10, v[40] = wsub(x2, r.t1) >> 10;
      v[24] = wadd(x3, r.t0) >> 10, v[32] = wsub(x3, r.t0) >> 10;
    }
  }
  for (int i = 0; i < 8; ++i) {
    const int *v = val + 8 * i;
    uint8_t *o = out + stride * i;
    Idct1D r(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const int bias = 65536 + (128 << 17);
    const int x0 = wadd(r.x0, bias), x1 = wadd(r.x1, bias), x2 = wadd(r.x2, bias), x3 = wadd(r.x3, bias);
    o[0] = clamp255(wadd(x0, r.t3) >> 17), o[7] = clamp255(wsub(x0, r.t3) >> 17);
    o[1] = clamp255(wadd(x1, r.t2) >> 17), o[6] = clamp255(wsub(x1, r.t2) >> 17);
    o[2] = clamp255(wadd(x2, r.t1) >> 17), o[5] = clamp255(wsub(x2, r.t1) >> 17);
    o[3] = clamp255(wadd(x3, r.t0) >> 17), o[4] = clamp255(wsub(x3, r.t0) >> 17);
  }
}

// One row of upsampled samples (stb_image.h:3449-3655).  `near` is the source row closer to the output row, `far` the
// other vertical neighbour; w = source samples.
void upsample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, uint32_t w, int hs, int vs) {
  if (hs == 1 && vs == 1) {
    memcpy(out, near, w);
  } else if (hs == 1 && vs == 2) {
    for (uint32_t i = 0; i < w; ++i)
      out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
  } else if (hs == 2 && vs == 1) {
    if (w == 1) {
      out[0] = out[1] = near[0];
      return;
    }
    out[0] = near[0];
    out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
    uint32_t i;
    for (i = 1; i + 1 < w; ++i) {
      const int n = 3 * near[i] + 2;
      out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
      out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
    }
    // sic: stb_image weights this sample towards near[w-2] (stb_image.h:3494), not towards near[w-1] as the centred
    // filter would; the reference's textures carry that column, so it is reproduced
    out[2 * i] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
    out[2 * i + 1] = near[w - 1];
  } else if (hs == 2 && vs == 2) {
    if (w == 1) {
      out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
      return;
    }
    int t1 = 3 * near[0] + far[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (uint32_t i = 1; i < w; ++i) {
      const int t0 = t1;
      t1 = 3 * near[i] + far[i];
      out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
      out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
  } else { // any other ratio: nearest neighbour horizontally (the row choice is the caller's)
    for (uint32_t i = 0; i < w; ++i)
      for (int j = 0; j < hs; ++j)
        out[(size_t)i * hs + j] = near[i];
  }
}

inline int fixed20(double x) { return ((int)(x * 4096.0f + 0.5f)) << 8; } // stb_image.h:3659

struct Decoder {
  const std::vector<uint8_t> &f;
  std::string &err;
  size_t pos = 2;
  uint16_t quant[4][64];
  bool quant_defined[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];
  Component comp[3];
  int ncomp = 0;
  uint32_t width = 0, height = 0, mcu_x = 0, mcu_y = 0;
  int hmax = 1, vmax = 1;
  uint32_t restart_interval = 0;
  bool have_frame = false, jfif = false, progressive = false;
  uint32_t eob_run = 0; // progressive AC scans: blocks still covered by the last end-of-band run (T.81 G.1.2.2)
  int adobe_transform = -1;
  bool rgb_ids = false;

  Decoder(const std::vector<uint8_t> &file, std::string &e) : f(file), err(e) { memset(quant, 0, sizeof quant); }
  bool fail(const char *m) {
    err = std::string("JPEG: ") + m;
    return false;
  }
  uint32_t u16(size_t p) const { return ((uint32_t)f[p] << 8) | f[p + 1]; }

  bool read_dqt(size_t p, size_t end) {
    while (p < end) {
      const int pq = f[p] >> 4, tq = f[p] & 15;
      ++p;
      if (pq > 1 || tq > 3)
        return fail("bad quantisation table header");
      if (p + (pq ? 128 : 64) > end)
        return fail("quantisation table cut short");
      for (int i = 0; i < 64; ++i) {
        quant[tq][kZigzag[i]] = (uint16_t)(pq ? u16(p) : f[p]);
        p += pq ? 2 : 1;
      }
      quant_defined[tq] = true;
    }
    return true;
  }
  bool read_dht(size_t p, size_t end) {
    while (p < end) {
      const int tc = f[p] >> 4, th = f[p] & 15;
      if (tc > 1 || th > 3 || p + 17 > end)
        return fail("bad Huffman table header");
      HuffTable &t = tc ? ac[th] : dc[th];
      int total = 0;
      t.count[0] = 0;
      for (int i = 1; i <= 16; ++i)
        total += (t.count[i] = f[p + i]);
      p += 17;
      if (total > 256 || p + (size_t)total > end)
        return fail("Huffman table cut short");
      memset(t.symbol, 0, sizeof t.symbol);
      memcpy(t.symbol, &f[p], (size_t)total);
      p += (size_t)total;
      if (!t.build())
        return fail("over-subscribed Huffman table");
      t.defined = true;
    }
    return true;
  }
  bool read_sof(size_t p, size_t end, bool prog) {
    progressive = prog;
    if (have_frame)
      return fail("more than one frame");
    if (end - p < 6)
      return fail("frame header cut short");
    if (f[p] != 8)
      return fail("only 8-bit samples are decoded");
    height = u16(p + 1), width = u16(p + 3);
    ncomp = f[p + 5];
    if (width == 0 || height == 0)
      return fail("empty image (a DNL-defined height is not supported)");
    if (ncomp == 4)
      return fail("4-component (CMYK / YCCK) files are not decoded");
    if (ncomp != 1 && ncomp != 3)
      return fail("bad component count");
    if (end - p < 6 + 3 * (size_t)ncomp)
      return fail("frame header cut short");
    if ((uint64_t)width * height > (1ull << 26)) // 8192 x 8192: the planes, coefficients and RGBA of a larger one run to gigabytes
      return fail("image too large (more than 2^26 pixels)");
    static const char rgb[3] = {'R', 'G', 'B'};
    rgb_ids = ncomp == 3;
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.id = f[p + 6 + 3 * i];
      c.h = f[p + 7 + 3 * i] >> 4, c.v = f[p + 7 + 3 * i] & 15;
      c.tq = f[p + 8 + 3 * i];
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
        return fail("bad sampling factors or table index");
      if (ncomp == 3 && c.id != rgb[i])
        rgb_ids = false;
      if (c.h > hmax)
        hmax = c.h;
      if (c.v > vmax)
        vmax = c.v;
    }
    for (int i = 0; i < ncomp; ++i)
      if (hmax % comp[i].h || vmax % comp[i].v)
        return fail("sampling factors that do not divide the largest are not decoded");
    mcu_x = (width + 8u * hmax - 1) / (8u * hmax), mcu_y = (height + 8u * vmax - 1) / (8u * vmax);
    for (int i = 0; i < ncomp; ++i) {
      Component &c = comp[i];
      c.x = (width * c.h + hmax - 1) / hmax, c.y = (height * c.v + vmax - 1) / vmax;
      c.w2 = mcu_x * c.h * 8, c.h2 = mcu_y * c.v * 8;
      c.plane.assign((size_t)c.w2 * c.h2, 0);
      if (progressive)
        c.coef.assign((size_t)c.w2 * c.h2, 0); // (w2 / 8) * (h2 / 8) blocks of 64
    }
    have_frame = true;
    return true;
  }

  bool decode_block(EntropyReader &br, Component &c, uint8_t *dst) {
    short data[64];
    memset(data, 0, sizeof data);
    const uint16_t *q = quant[c.tq];
    const int t = decode_symbol(br, dc[c.td]);
    if (t < 0 || t > 15)
      return fail("bad Huffman code");
    const int diff = t ? extend(br.receive(t), t) : 0;
    if (diff > 65535 || diff < -65535 || c.dc_pred + diff > 32767 || c.dc_pred + diff < -32768)
      return fail("bad DC delta"); // (stb_image refuses these too: the prediction must stay a 16-bit value)
    c.dc_pred += diff;
    const long long dcq = (long long)c.dc_pred * q[0];
    if (dcq > 32767 || dcq < -32768)
      return fail("DC coefficient out of range");
    data[0] = (short)dcq;
    for (int k = 1; k < 64;) {
      const int rs = decode_symbol(br, ac[c.ta]);
      if (rs < 0)
        return fail("bad Huffman code");
      const int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (rs != 0xF0)
          break; // end of block
        k += 16;
      } else {
        k += r;
        if (k > 63)
          return fail("coefficient index past the block");
        const int zz = kZigzag[k++];
        data[zz] = (short)(extend(br.receive(s), s) * q[zz]);
      }
    }
    idct_block(dst, c.w2, data);
    return true;
  }

  // Progressive frames (T.81 Annex G; the decoding procedures are fixed by the standard, stb_image.h:2265-2413 follows
  // them): a scan carries either the DC coefficients (Ss = 0) or one band Ss..Se of the AC coefficients of ONE component,
  // as a first pass (Ah = 0: values shifted left by Al) or as a refinement of one more bit (Ah > 0).
  bool prog_dc(EntropyReader &br, Component &c, short *d, int ah, int al) {
    if (ah == 0) {
      const int t = decode_symbol(br, dc[c.td]);
      if (t < 0 || t > 15)
        return fail("bad Huffman code");
      const int diff = t ? extend(br.receive(t), t) : 0;
      if (c.dc_pred + diff > 32767 || c.dc_pred + diff < -32768)
        return fail("bad DC delta");
      c.dc_pred += diff;
      const long long dcs = (long long)c.dc_pred * (1 << al);
      if (dcs > 32767 || dcs < -32768)
        return fail("DC coefficient out of range");
      d[0] = (short)dcs;
    } else if (br.bit()) {
      d[0] = (short)(d[0] + (1 << al));
    }
    return true;
  }
  static void refine(EntropyReader &br, short *p, short bit) { // one correction bit for a coefficient already nonzero
    if (br.bit() && (*p & bit) == 0)
      *p = (short)(*p > 0 ? *p + bit : *p - bit);
  }
  bool prog_ac(EntropyReader &br, Component &c, short *d, int ss, int se, int ah, int al) {
    const HuffTable &t = ac[c.ta];
    if (ah == 0) {
      if (eob_run) {
        --eob_run;
        return true;
      }
      for (int k = ss; k <= se;) {
        const int rs = decode_symbol(br, t);
        if (rs < 0)
          return fail("bad Huffman code");
        const int s = rs & 15, r = rs >> 4;
        if (s == 0) {
          if (r < 15) { // EOBn: this block and 2^r + bits - 1 more end here
            eob_run = (1u << r) + (r ? (uint32_t)br.receive(r) : 0u) - 1u;
            break;
          }
          k += 16;
        } else {
          k += r;
          if (k > 63)
            return fail("coefficient index past the block");
          d[kZigzag[k++]] = (short)(extend(br.receive(s), s) * (1 << al));
        }
      }
      return true;
    }
    const short bit = (short)(1 << al);
    if (eob_run) {
      --eob_run;
      for (int k = ss; k <= se; ++k)
        if (d[kZigzag[k]] != 0)
          refine(br, &d[kZigzag[k]], bit);
      return true;
    }
    int k = ss;
    do {
      const int rs = decode_symbol(br, t);
      if (rs < 0)
        return fail("bad Huffman code");
      int s = rs & 15, r = rs >> 4;
      if (s == 0) {
        if (r < 15) {
          eob_run = (1u << r) - 1u + (r ? (uint32_t)br.receive(r) : 0u);
          r = 64; // the rest of the band only takes correction bits
        }
      } else {
        if (s != 1)
          return fail("bad Huffman code");
        s = br.bit() ? bit : -bit; // the new coefficient's value: +-2^Al
      }
      while (k <= se) { // skip r zero coefficients (correcting the nonzero ones on the way), then place s
        short *p = &d[kZigzag[k++]];
        if (*p != 0) {
          refine(br, p, bit);
        } else {
          if (r == 0) {
            *p = (short)s;
            break;
          }
          --r;
        }
      }
    } while (k <= se);
    return true;
  }
  void finish_progressive() { // dequantise and transform every block (stb_image.h:3080-3097)
    for (int n = 0; n < ncomp; ++n) {
      Component &c = comp[n];
      const uint32_t bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, cw = c.w2 / 8;
      for (uint32_t j = 0; j < bh; ++j)
        for (uint32_t i = 0; i < bw; ++i) {
          short *d = &c.coef[64 * ((size_t)j * cw + i)];
          for (int k = 0; k < 64; ++k)
            d[k] = (short)(d[k] * quant[c.tq][k]);
          idct_block(&c.plane[(size_t)j * 8 * c.w2 + (size_t)i * 8], c.w2, d);
        }
    }
  }

  bool read_scan(size_t p, size_t end) {
    if (!have_frame)
      return fail("scan before the frame header");
    if (end - p < 1)
      return fail("scan header cut short");
    const int ns = f[p];
    if (ns < 1 || ns > ncomp || end - p < 4 + 2 * (size_t)ns)
      return fail("bad scan header");
    int order[3];
    for (int i = 0; i < ns; ++i) {
      const int id = f[p + 1 + 2 * i], tables = f[p + 2 + 2 * i];
      int which = -1;
      for (int k = 0; k < ncomp; ++k)
        if (comp[k].id == id)
          which = k;
      if (which < 0)
        return fail("scan names an unknown component");
      comp[which].td = tables >> 4, comp[which].ta = tables & 15;
      if (comp[which].td > 3 || comp[which].ta > 3 || !quant_defined[comp[which].tq])
        return fail("scan uses an undefined table");
      order[i] = which;
    }
    const size_t q = p + 1 + 2 * (size_t)ns;
    const int ss = f[q], se = f[q + 1], ah = f[q + 2] >> 4, al = f[q + 2] & 15;
    if (!progressive && (ss != 0 || se != 63 || ah != 0 || al != 0))
      return fail("spectral selection / successive approximation in a sequential frame");
    if (progressive && (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13 || (ss == 0 && se != 0) || (ss != 0 && ns != 1)))
      return fail("bad progressive scan parameters");
    for (int i = 0; i < ns; ++i) { // only the tables the scan reads need to exist
      const Component &c = comp[order[i]];
      const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
      if ((need_dc && !dc[c.td].defined) || (need_ac && !ac[c.ta].defined))
        return fail("scan uses an undefined table");
    }
    eob_run = 0;
    EntropyReader br{&f[0] + end, &f[0] + f.size()};
    for (int k = 0; k < ncomp; ++k)
      comp[k].dc_pred = 0;
    uint32_t todo = restart_interval ? restart_interval : 0xFFFFFFFFu;
    int next_rst = 0;
    auto restart_if_due = [&]() -> bool {
      if (--todo != 0)
        return true;
      // byte-align: the remaining bits of the interval are padding; the next bytes must be RSTn
      if (!br.hit_marker) { // the reader has not met the marker yet: skip padding bytes up to it
        while (br.p < br.end && !(br.p[0] == 0xFF && br.p + 1 < br.end && br.p[1] != 0x00))
          ++br.p;
      }
      while (br.p + 2 < br.end && br.p[0] == 0xFF && br.p[1] == 0xFF) // fill bytes before a marker (T.81 B.1.1.2)
        ++br.p;
      if (br.p + 1 >= br.end || br.p[0] != 0xFF || br.p[1] != (0xD0 + next_rst))
        return true; // no restart marker here: the scan ends (what follows is left to the marker loop)
      br.p += 2;
      next_rst = (next_rst + 1) & 7;
      br.reset();
      for (int k = 0; k < ncomp; ++k)
        comp[k].dc_pred = 0;
      eob_run = 0;
      todo = restart_interval;
      return true;
    };
    auto block = [&](Component &c, size_t bi, size_t bj) -> bool { // block (bi, bj) of the component's plane
      if (!progressive)
        return decode_block(br, c, &c.plane[bj * 8 * c.w2 + bi * 8]);
      short *d = &c.coef[64 * (bj * (c.w2 / 8) + bi)];
      return ss == 0 ? prog_dc(br, c, d, ah, al) : prog_ac(br, c, d, ss, se, ah, al);
    };
    if (ns == 1) { // non-interleaved: the component's own blocks, row by row
      Component &c = comp[order[0]];
      const uint32_t bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
      for (uint32_t j = 0; j < bh; ++j)
        for (uint32_t i = 0; i < bw; ++i) {
          if (!block(c, i, j))
            return false;
          restart_if_due();
        }
    } else {
      for (uint32_t j = 0; j < mcu_y; ++j)
        for (uint32_t i = 0; i < mcu_x; ++i) {
          for (int k = 0; k < ns; ++k) {
            Component &c = comp[order[k]];
            for (int y = 0; y < c.v; ++y)
              for (int x = 0; x < c.h; ++x) {
                if (!block(c, (size_t)i * c.h + x, (size_t)j * c.v + y))
                  return false;
              }
          }
          restart_if_due();
        }
    }
    pos = (size_t)(br.p - &f[0]); // on the marker that ended the data (or at the end of the file)
    return true;
  }

  bool run() {
    bool saw_scan = false;
    while (pos + 4 <= f.size()) {
      if (f[pos] != 0xFF) { // stray bytes between segments (after a scan): resynchronise on the next marker
        ++pos;
        continue;
      }
      const int m = f[pos + 1];
      if (m == 0xFF || m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { // fill byte, stuffed byte, stray RSTn
        pos += m == 0xFF ? 1 : 2;
        continue;
      }
      if (m == 0xD9)
        break;
      const size_t len = u16(pos + 2);
      if (len < 2 || pos + 2 + len > f.size())
        return fail("segment runs past the end of the file");
      const size_t p = pos + 4, end = pos + 2 + len;
      pos = end;
      if (m == 0xDB) {
        if (!read_dqt(p, end))
          return false;
      } else if (m == 0xC4) {
        if (!read_dht(p, end))
          return false;
      } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
        if (!read_sof(p, end, m == 0xC2))
          return false;
      } else if (m >= 0xC3 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
        return fail("lossless, differential and arithmetic-coded frames are not decoded");
      } else if (m == 0xCC) {
        return fail("arithmetic coding is not decoded");
      } else if (m == 0xDD) {
        if (len != 4)
          return fail("bad restart interval segment");
        restart_interval = u16(p);
      } else if (m == 0xE0) {
        if (len >= 7 && !memcmp(&f[p], "JFIF\0", 5))
          jfif = true;
      } else if (m == 0xEE) {
        if (len >= 14 && !memcmp(&f[p], "Adobe", 5))
          adobe_transform = f[p + 11];
      } else if (m == 0xDA) {
        if (!read_scan(p, end))
          return false;
        saw_scan = true;
      } // other APPn / COM segments are skipped
    }
    if (!have_frame || !saw_scan)
      return fail("no image data");
    if (progressive)
      finish_progressive();
    return true;
  }

  void emit(Image &img) {
    img.width = width, img.height = height;
    img.rgba.assign((size_t)width * height * 4, 255);
    // stb_image.h:3879: three components are RGB (no conversion) when the ids say so or an Adobe marker without a
    // JFIF one declares transform 0
    const bool is_rgb = ncomp == 3 && (rgb_ids || (adobe_transform == 0 && !jfif));
    struct Res {
      int hs, vs, ystep;
      uint32_t w_lores, ypos;
      const uint8_t *line0, *line1;
      std::vector<uint8_t> buf;
    } res[3];
    for (int k = 0; k < ncomp; ++k) {
      Res &r = res[k];
      r.hs = hmax / comp[k].h, r.vs = vmax / comp[k].v;
      r.ystep = r.vs >> 1; // the first output row lies in the upper half of the first source row
      r.w_lores = (width + r.hs - 1) / r.hs;
      r.ypos = 0;
      r.line0 = r.line1 = comp[k].plane.data();
      r.buf.assign((size_t)width + 8, 0);
    }
    static const int cr_r = fixed20(1.40200f), cr_g = -fixed20(0.71414f), cb_g = -fixed20(0.34414f), cb_b = fixed20(1.77200f);
    for (uint32_t j = 0; j < height; ++j) {
      const uint8_t *row[3] = {nullptr, nullptr, nullptr};
      for (int k = 0; k < ncomp; ++k) { // the row walk of stb_image.h:3926-3941
        Res &r = res[k];
        const bool y_bot = r.ystep >= (r.vs >> 1);
        upsample_row(r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
        row[k] = r.buf.data();
        if (++r.ystep >= r.vs) {
          r.ystep = 0;
          r.line0 = r.line1;
          if (++r.ypos < comp[k].y)
            r.line1 += comp[k].w2;
        }
      }
      uint8_t *o = &img.rgba[(size_t)j * width * 4];
      if (ncomp == 1) {
        for (uint32_t i = 0; i < width; ++i, o += 4)
          o[0] = o[1] = o[2] = row[0][i];
      } else if (is_rgb) {
        for (uint32_t i = 0; i < width; ++i, o += 4)
          o[0] = row[0][i], o[1] = row[1][i], o[2] = row[2][i];
      } else {
        for (uint32_t i = 0; i < width; ++i, o += 4) { // stb_image.h:3660-3683
          const int y_fixed = (row[0][i] << 20) + (1 << 19);
          const int cb = row[1][i] - 128, cr = row[2][i] - 128;
          const int r = (y_fixed + cr * cr_r) >> 20;
          const int g = (y_fixed + cr * cr_g + (int)((uint32_t)(cb * cb_g) & 0xFFFF0000u)) >> 20;
          const int b = (y_fixed + cb * cb_b) >> 20;
          o[0] = clamp255(r), o[1] = clamp255(g), o[2] = clamp255(b);
        }
      }
    }
  }
};

} // namespace

bool decode_jpeg(const std::vector<uint8_t> &file, Image &out, std::string &err) {
  if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8)
    return err = "JPEG: no SOI marker", false;
  Decoder d(file, err);
  if (!d.run())
    return false;
  d.emit(out);
  return true;
}

} // namespace rta
