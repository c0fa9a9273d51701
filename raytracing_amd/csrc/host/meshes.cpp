// meshes.cpp — see meshes.h.
#include "meshes.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <map>
#include <sstream>
#include <tuple>
#include <unordered_map>

#include "hmath.h"

namespace rta {

namespace {
void push3(std::vector<float> &v, vec3 p) {
  v.push_back(p.x), v.push_back(p.y), v.push_back(p.z);
}
void push2(std::vector<float> &v, float a, float b) { v.push_back(a), v.push_back(b); }
const float kPiF = 3.14159265358979323846264338327950288f;   // glm::pi<float>()
const float kTwoPiF = 6.28318530717958647692528676655900576f; // glm::two_pi<float>()
} // namespace

// The reference's three procedural primitives (generate_sphere / generate_plane / generate_cube,
// src/Renderer.cpp:64-213) produce BLAS 0 / 1 / 2 of every scene, so vertex values, vertex order and triangle
// order must be theirs to the bit (the BVH and every hit id follow from them).  Sphere and plane are both a
// (cols + 1) x (rows + 1) lattice over [0, 1]^2 whose cells split into the triangles (a, below, a + 1) and
// (a + 1, below, below + 1); one lattice builder serves both, the surface comes in as a function of the lattice
// coordinates.  The cube is six quads from a corner-sign table.
namespace {
template <typename Surface> Mesh lattice_mesh(uint32_t cols, uint32_t rows, Surface surface) {
  Mesh m;
  const uint32_t pitch = cols + 1;
  m.positions.reserve((size_t)pitch * (rows + 1) * 3);
  for (uint32_t r = 0; r <= rows; ++r) {
    const float t = float(r) / rows; // second lattice coordinate (v of the sphere, vz of the plane)
    for (uint32_t c = 0; c <= cols; ++c) {
      const float s = float(c) / cols;
      vec3 position, normal;
      surface(s, t, position, normal);
      push3(m.positions, vec3(0.f) + position); // (`center + ...` with the reference's default centre: -0 becomes +0)
      push3(m.normals, normal);
      push2(m.uvs, s, t);
    }
  }
  m.indices.reserve((size_t)cols * rows * 6);
  for (uint32_t r = 0; r < rows; ++r)
    for (uint32_t c = 0; c < cols; ++c) {
      const uint32_t a = r * pitch + c, below = a + pitch;
      const uint32_t cell[6] = {a, below, a + 1, a + 1, below, below + 1};
      m.indices.insert(m.indices.end(), cell, cell + 6);
    }
  return m;
}
} // namespace

Mesh make_sphere(float radius, uint32_t segments, uint32_t rings) { // Renderer.cpp:64-107
  return lattice_mesh(segments, rings, [radius](float u, float v, vec3 &position, vec3 &normal) {
    const float polar = v * kPiF, azimuth = u * kTwoPiF;
    const float sp = sinf(polar), cp = cosf(polar), sa = sinf(azimuth), ca = cosf(azimuth);
    normal = vec3(sp * ca, cp, sp * sa);
    position = radius * normal;
  });
}

Mesh make_plane(float width, float depth, uint32_t x_segments, uint32_t z_segments) { // Renderer.cpp:109-145
  return lattice_mesh(x_segments, z_segments, [width, depth](float ux, float vz, vec3 &position, vec3 &normal) {
    position = vec3((ux - 0.5f) * width, 0.f, (vz - 0.5f) * depth);
    normal = vec3(0.f, 1.f, 0.f);
  });
}

Mesh make_cube(float width, float height, float depth) { // Renderer.cpp:147-213
  // faces in the reference's order (+Z, -Z, -X, +X, +Y, -Y); per face its four corners as signs of the half extents,
  // counter-clockwise seen from outside, with uv (0,0) (1,0) (1,1) (0,1) and the triangles (0,1,2) (0,2,3)
  static const int8_t kFace[6][4][3] = {
      {{-1, -1, +1}, {+1, -1, +1}, {+1, +1, +1}, {-1, +1, +1}}, {{+1, -1, -1}, {-1, -1, -1}, {-1, +1, -1}, {+1, +1, -1}},
      {{-1, -1, -1}, {-1, -1, +1}, {-1, +1, +1}, {-1, +1, -1}}, {{+1, -1, +1}, {+1, -1, -1}, {+1, +1, -1}, {+1, +1, +1}},
      {{-1, +1, +1}, {+1, +1, +1}, {+1, +1, -1}, {-1, +1, -1}}, {{-1, -1, -1}, {+1, -1, -1}, {+1, -1, +1}, {-1, -1, +1}}};
  static const int8_t kNormal[6][3] = {{0, 0, 1}, {0, 0, -1}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
  static const float kUv[4][2] = {{0.f, 0.f}, {1.f, 0.f}, {1.f, 1.f}, {0.f, 1.f}};
  const float half[3] = {width * 0.5f, height * 0.5f, depth * 0.5f};
  Mesh m;
  for (int f = 0; f < 6; ++f) {
    const uint32_t first = (uint32_t)(m.positions.size() / 3);
    for (int k = 0; k < 4; ++k) {
      const int8_t *sg = kFace[f][k];
      push3(m.positions, vec3(0.f) + vec3(sg[0] < 0 ? -half[0] : half[0], sg[1] < 0 ? -half[1] : half[1], sg[2] < 0 ? -half[2] : half[2]));
      push3(m.normals, vec3((float)kNormal[f][0], (float)kNormal[f][1], (float)kNormal[f][2]));
      push2(m.uvs, kUv[k][0], kUv[k][1]);
    }
    for (uint32_t k : {0u, 1u, 2u, 0u, 2u, 3u})
      m.indices.push_back(first + k);
  }
  return m;
}

// ---------------------------------------------------------------------------
// Dragon-class stand-in.
// ---------------------------------------------------------------------------
namespace {
struct D3 {
  double x, y, z;
};
D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
D3 dcross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
D3 dnorm(D3 a) {
  double l = sqrt(ddot(a, a));
  return l > 0 ? a * (1.0 / l) : D3{0, 1, 0};
}
D3 knot(double t) { // (2,3) torus knot, fits a ~2.6 x 1.3 x 2.6 box
  double r = 0.8 + 0.35 * cos(3 * t);
  return {r * cos(2 * t), 0.45 * sin(3 * t) + 0.55, r * sin(2 * t)};
}
// integer lattice hash -> [0,1)
double lattice(uint32_t i, uint32_t j, uint32_t seed) {
  uint32_t h = i * 0x9E3779B1u ^ (j * 0x85EBCA77u + seed * 0xC2B2AE3Du);
  h ^= h >> 15, h *= 0x2C1B3C6Du, h ^= h >> 12, h *= 0x297A2D39u, h ^= h >> 15;
  return (double)h / 4294967296.0;
}
// periodic value noise on a cu x cv lattice
double vnoise(double u, double v, uint32_t cu, uint32_t cv, uint32_t seed) {
  double fu = u * cu, fv = v * cv;
  uint32_t iu = (uint32_t)floor(fu), iv = (uint32_t)floor(fv);
  double a = fu - iu, b = fv - iv;
  a = a * a * (3 - 2 * a), b = b * b * (3 - 2 * b);
  auto L = [&](uint32_t x, uint32_t y) { return lattice(x % cu, y % cv, seed); };
  double n0 = L(iu, iv) * (1 - a) + L(iu + 1, iv) * a;
  double n1 = L(iu, iv + 1) * (1 - a) + L(iu + 1, iv + 1) * a;
  return n0 * (1 - b) + n1 * b;
}
} // namespace

Mesh make_dragon(uint32_t u_seg, uint32_t v_seg, uint32_t seed) {
  Mesh m;
  if (u_seg < 3 || v_seg < 3)
    return m;
  const double two_pi = 6.283185307179586476925286766559;
  const size_t nv = (size_t)u_seg * v_seg;
  std::vector<D3> P(nv);
  for (uint32_t i = 0; i < u_seg; ++i) {
    double u = (double)i / u_seg, t = u * two_pi;
    D3 c = knot(t);
    D3 tan = dnorm(knot(t + 1e-4) - knot(t - 1e-4));
    D3 up = dnorm(dcross(tan, dcross(D3{0, 1, 0}, tan))); // up made orthogonal to the tangent
    D3 side = dcross(tan, up);
    // body radius: thick "torso", thin "neck/tail", periodic over the loop
    double body = 0.16 + 0.06 * sin(t * 2 + 0.7) + 0.03 * sin(t * 5 + 1.9);
    for (uint32_t j = 0; j < v_seg; ++j) {
      double v = (double)j / v_seg, a = v * two_pi;
      double ridge = pow(fmax(0.0, cos(a)), 24.0) * (0.55 + 0.45 * sin(t * 40)); // dorsal crest
      double scales = 0.5 + 0.5 * sin(t * 180 + (j & 1) * 3.14159) * sin(a * 24);   // scale rows
      double r = body * (1.0 + 0.55 * ridge + 0.06 * scales + 0.25 * (vnoise(u, v, 48, 12, seed) - 0.5) +
                         0.10 * (vnoise(u, v, 400, 64, seed + 1) - 0.5) +
                         0.04 * (vnoise(u, v, 1700, 256, seed + 2) - 0.5));
      // spikes: sparse lattice cells pushed outwards
      if (lattice(i / 6, j / 6, seed + 3) > 0.985)
        r *= 1.0 + 0.8 * sin(3.14159 * ((i % 6) + 0.5) / 6.0) * sin(3.14159 * ((j % 6) + 0.5) / 6.0);
      P[(size_t)i * v_seg + j] = c + (up * cos(a) + side * sin(a)) * r;
    }
  }
  // area-weighted vertex normals
  std::vector<D3> N(nv, D3{0, 0, 0});
  m.indices.reserve(nv * 6);
  for (uint32_t i = 0; i < u_seg; ++i)
    for (uint32_t j = 0; j < v_seg; ++j) {
      uint32_t i1 = (i + 1) % u_seg, j1 = (j + 1) % v_seg;
      uint32_t a = i * v_seg + j, b = i1 * v_seg + j, c = i1 * v_seg + j1, d = i * v_seg + j1;
      for (auto tri : {std::tuple<uint32_t, uint32_t, uint32_t>{a, b, c}, {a, c, d}}) {
        uint32_t p = std::get<0>(tri), q = std::get<1>(tri), r = std::get<2>(tri);
        D3 fn = dcross(P[q] - P[p], P[r] - P[p]);
        N[p] = N[p] + fn, N[q] = N[q] + fn, N[r] = N[r] + fn;
        m.indices.push_back(p), m.indices.push_back(q), m.indices.push_back(r);
      }
    }
  m.positions.reserve(nv * 3), m.normals.reserve(nv * 3), m.uvs.reserve(nv * 2);
  for (size_t k = 0; k < nv; ++k) {
    D3 n = dnorm(N[k]);
    m.positions.push_back((float)P[k].x), m.positions.push_back((float)P[k].y),
        m.positions.push_back((float)P[k].z);
    m.normals.push_back((float)n.x), m.normals.push_back((float)n.y), m.normals.push_back((float)n.z);
    m.uvs.push_back((float)((double)(k / v_seg) / u_seg)), m.uvs.push_back((float)((double)(k % v_seg) / v_seg));
  }
  return m;
}

// ---------------------------------------------------------------------------
// subdivided + displaced base mesh (second dragon-class stand-in)
// ---------------------------------------------------------------------------
namespace {
double lattice3(int32_t i, int32_t j, int32_t k, uint32_t seed) {
  uint32_t h = (uint32_t)i * 0x9E3779B1u ^ ((uint32_t)j * 0x85EBCA77u + seed * 0xC2B2AE3Du) ^ ((uint32_t)k * 0x27D4EB2Fu);
  h ^= h >> 15, h *= 0x2C1B3C6Du, h ^= h >> 12, h *= 0x297A2D39u, h ^= h >> 15;
  return (double)h / 4294967296.0;
}
double vnoise3(D3 p, uint32_t seed) {
  const double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
  const int32_t ix = (int32_t)fx, iy = (int32_t)fy, iz = (int32_t)fz;
  double a = p.x - fx, b = p.y - fy, c = p.z - fz;
  a = a * a * (3 - 2 * a), b = b * b * (3 - 2 * b), c = c * c * (3 - 2 * c);
  double v = 0;
  for (int dz = 0; dz < 2; ++dz)
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx)
        v += lattice3(ix + dx, iy + dy, iz + dz, seed) * (dx ? a : 1 - a) * (dy ? b : 1 - b) * (dz ? c : 1 - c);
  return v;
}
} // namespace

Mesh subdivide_displace(const Mesh &base, uint32_t levels, float amplitude, uint32_t seed) {
  std::vector<D3> P, N;
  std::vector<double> U, V;
  const size_t nv0 = base.positions.size() / 3;
  for (size_t k = 0; k < nv0; ++k) {
    P.push_back({base.positions[3 * k], base.positions[3 * k + 1], base.positions[3 * k + 2]});
    N.push_back({base.normals[3 * k], base.normals[3 * k + 1], base.normals[3 * k + 2]});
    U.push_back(base.uvs.size() > 2 * k ? base.uvs[2 * k] : 0.0), V.push_back(base.uvs.size() > 2 * k + 1 ? base.uvs[2 * k + 1] : 0.0);
  }
  std::vector<uint32_t> idx = base.indices;
  for (uint32_t l = 0; l < levels; ++l) {
    std::unordered_map<uint64_t, uint32_t> mid; // edge (lo, hi) -> its midpoint vertex
    mid.reserve(idx.size());
    auto midpoint = [&](uint32_t a, uint32_t b) {
      const uint64_t key = a < b ? ((uint64_t)a << 32) | b : ((uint64_t)b << 32) | a;
      auto it = mid.find(key);
      if (it != mid.end())
        return it->second;
      const uint32_t v = (uint32_t)P.size();
      P.push_back((P[a] + P[b]) * 0.5), N.push_back(dnorm(N[a] + N[b]));
      U.push_back((U[a] + U[b]) * 0.5), V.push_back((V[a] + V[b]) * 0.5);
      mid.emplace(key, v);
      return v;
    };
    std::vector<uint32_t> next;
    next.reserve(idx.size() * 4);
    for (size_t t = 0; t + 2 < idx.size(); t += 3) {
      const uint32_t a = idx[t], b = idx[t + 1], c = idx[t + 2];
      const uint32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
      for (uint32_t v : {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca})
        next.push_back(v);
    }
    idx.swap(next);
  }
  // the base's extent sets the noise scale
  D3 lo = P.empty() ? D3{0, 0, 0} : P[0], hi = lo;
  for (const D3 &p : P)
    lo = {fmin(lo.x, p.x), fmin(lo.y, p.y), fmin(lo.z, p.z)}, hi = {fmax(hi.x, p.x), fmax(hi.y, p.y), fmax(hi.z, p.z)};
  const double extent = fmax(fmax(hi.x - lo.x, hi.y - lo.y), fmax(hi.z - lo.z, 1e-9));
  for (size_t k = 0; k < P.size(); ++k) {
    const D3 q = (P[k] - lo) * (1.0 / extent);
    double d = 0, w = 1, f = 6;
    for (uint32_t o = 0; o < 4; ++o, w *= 0.5, f *= 3.1)
      d += w * (vnoise3(q * f, seed + o) - 0.5);
    P[k] = P[k] + dnorm(N[k]) * (amplitude * extent * d);
  }
  std::vector<D3> FN(P.size(), D3{0, 0, 0});
  for (size_t t = 0; t + 2 < idx.size(); t += 3) {
    const D3 fn = dcross(P[idx[t + 1]] - P[idx[t]], P[idx[t + 2]] - P[idx[t]]);
    FN[idx[t]] = FN[idx[t]] + fn, FN[idx[t + 1]] = FN[idx[t + 1]] + fn, FN[idx[t + 2]] = FN[idx[t + 2]] + fn;
  }
  Mesh m;
  m.indices = std::move(idx);
  m.positions.reserve(P.size() * 3), m.normals.reserve(P.size() * 3), m.uvs.reserve(P.size() * 2);
  for (size_t k = 0; k < P.size(); ++k) {
    const D3 n = ddot(FN[k], FN[k]) > 0 ? dnorm(FN[k]) : dnorm(N[k]);
    m.positions.push_back((float)P[k].x), m.positions.push_back((float)P[k].y), m.positions.push_back((float)P[k].z);
    m.normals.push_back((float)n.x), m.normals.push_back((float)n.y), m.normals.push_back((float)n.z);
    m.uvs.push_back((float)U[k]), m.uvs.push_back((float)V[k]);
  }
  return m;
}

// ---------------------------------------------------------------------------
// OBJ / MTL
// ---------------------------------------------------------------------------
namespace {
std::string dirname_of(const std::string &p) {
  size_t s = p.find_last_of("/\\");
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

void load_mtl(const std::string &path, std::vector<ObjMaterial> &mats) {
  std::ifstream in(path);
  if (!in)
    return; // tinyobj only warns when the .mtl is missing
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#')
      continue;
    if (key == "newmtl") {
      ObjMaterial m;
      ss >> m.name;
      m.kd[0] = m.kd[1] = m.kd[2] = 0.6f; // tinyobj's default diffuse when Kd is absent
      mats.push_back(m);
    } else if (!mats.empty() && key == "Kd") {
      ss >> mats.back().kd[0] >> mats.back().kd[1] >> mats.back().kd[2];
      mats.back().has_kd = true;
    } else if (!mats.empty() && key == "map_Kd") {
      // tinyobj's ParseTextureNameAndOption: options (with their values) are skipped, the rest is the file name
      std::string tok, name;
      while (ss >> tok) {
        int skip = -1;
        if (tok == "-blendu" || tok == "-blendv" || tok == "-clamp" || tok == "-boost" || tok == "-bm" ||
            tok == "-imfchan" || tok == "-type" || tok == "-texres" || tok == "-colorspace")
          skip = 1;
        else if (tok == "-mm")
          skip = 2;
        else if (tok == "-o" || tok == "-s" || tok == "-t")
          skip = 3; // u [v [w]]: tinyobj reads up to three numbers
        if (skip < 0) {
          name += (name.empty() ? "" : " ") + tok;
          continue;
        }
        for (int k = 0; k < skip; ++k) {
          std::streampos at = ss.tellg();
          std::string val;
          if (!(ss >> val))
            break;
          if ((tok == "-o" || tok == "-s" || tok == "-t") && k > 0) { // optional v, w: only numbers are consumed
            char *endp = nullptr;
            strtod(val.c_str(), &endp);
            if (endp == val.c_str() || *endp) {
              ss.clear();
              ss.seekg(at);
              break;
            }
          }
        }
      }
      mats.back().map_kd = name;
    }
  }
}

struct Corner {
  int v, t, n;
};
struct VKey {
  float f[8];
  bool operator<(const VKey &o) const { return memcmp(f, o.f, sizeof f) < 0; }
};
} // namespace

bool load_obj(const std::string &path, ObjModel &out, std::string &err) {
  std::ifstream in(path);
  if (!in) {
    err = "cannot open " + path;
    return false;
  }
  out = ObjModel();
  std::vector<float> vs, vns, vts;
  std::map<int, size_t> group_of_material; // material id -> group index
  std::vector<std::map<VKey, uint32_t>> weld;
  int current_material = -1;
  std::string line;
  size_t line_no = 0;
  while (std::getline(in, line)) {
    ++line_no;
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#')
      continue;
    if (key == "v") {
      float x, y, z;
      if (!(ss >> x >> y >> z)) {
        err = path + ":" + std::to_string(line_no) + ": bad vertex";
        return false;
      }
      vs.push_back(x), vs.push_back(y), vs.push_back(z);
    } else if (key == "vn") {
      float x = 0, y = 0, z = 0;
      ss >> x >> y >> z;
      vns.push_back(x), vns.push_back(y), vns.push_back(z);
    } else if (key == "vt") {
      float u = 0, v = 0;
      ss >> u >> v;
      vts.push_back(u), vts.push_back(v);
    } else if (key == "mtllib") {
      std::string name;
      ss >> name;
      load_mtl(dirname_of(path) + "/" + name, out.materials);
    } else if (key == "usemtl") {
      std::string name;
      ss >> name;
      current_material = -1;
      for (size_t i = 0; i < out.materials.size(); ++i)
        if (out.materials[i].name == name)
          current_material = (int)i;
    } else if (key == "f") {
      std::vector<Corner> corners;
      std::string tok;
      while (ss >> tok) {
        Corner c{0, 0, 0};
        int *slot[3] = {&c.v, &c.t, &c.n};
        size_t pos = 0;
        for (int k = 0; k < 3 && pos <= tok.size(); ++k) {
          size_t slash = tok.find('/', pos);
          std::string part = tok.substr(pos, slash == std::string::npos ? std::string::npos : slash - pos);
          if (!part.empty())
            *slot[k] = atoi(part.c_str());
          if (slash == std::string::npos)
            break;
          pos = slash + 1;
        }
        int nv = (int)(vs.size() / 3), nt = (int)(vts.size() / 2), nn = (int)(vns.size() / 3);
        c.v = c.v > 0 ? c.v - 1 : (c.v < 0 ? nv + c.v : -1);
        c.t = c.t > 0 ? c.t - 1 : (c.t < 0 ? nt + c.t : -1);
        c.n = c.n > 0 ? c.n - 1 : (c.n < 0 ? nn + c.n : -1);
        if (c.v < 0 || c.v >= nv || c.t >= nt || c.n >= nn) {
          err = path + ":" + std::to_string(line_no) + ": face index out of range";
          return false;
        }
        corners.push_back(c);
      }
      if (corners.size() < 3)
        continue;
      auto it = group_of_material.find(current_material);
      if (it == group_of_material.end()) {
        it = group_of_material.emplace(current_material, out.groups.size()).first;
        out.groups.emplace_back();
        out.group_material.push_back(current_material);
        weld.emplace_back();
      }
      Mesh &g = out.groups[it->second];
      std::map<VKey, uint32_t> &w = weld[it->second];
      for (size_t k = 1; k + 1 < corners.size(); ++k) { // fan
        const Corner tri[3] = {corners[0], corners[k], corners[k + 1]};
        vec3 p[3];
        for (int q = 0; q < 3; ++q)
          p[q] = vec3(&vs[3 * (size_t)tri[q].v]);
        vec3 fn = cross(p[1] - p[0], p[2] - p[0]);
        float fl = length(fn);
        fn = fl > 0.f ? fn / fl : vec3(0.f, 1.f, 0.f);
        for (int q = 0; q < 3; ++q) {
          vec3 n = tri[q].n >= 0 ? vec3(&vns[3 * (size_t)tri[q].n]) : fn;
          float u = 0.f, v = 0.f;
          if (tri[q].t >= 0) {
            u = vts[2 * (size_t)tri[q].t];
            v = 1.0f - vts[2 * (size_t)tri[q].t + 1]; // Scene.cpp:91-93
          }
          VKey key{{p[q].x, p[q].y, p[q].z, n.x, n.y, n.z, u, v}};
          auto found = w.find(key);
          if (found == w.end()) {
            found = w.emplace(key, (uint32_t)(g.positions.size() / 3)).first;
            push3(g.positions, p[q]);
            push3(g.normals, n);
            push2(g.uvs, u, v);
          }
          g.indices.push_back(found->second);
        }
      }
    }
  }
  if (out.groups.empty()) {
    err = path + ": no faces";
    return false;
  }
  return true;
}

} // namespace rta
