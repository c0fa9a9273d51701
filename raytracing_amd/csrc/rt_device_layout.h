// rt_device_layout.h — the flattened scene as it lives in HBM.
//
// The reference keeps 32-B BVH nodes, a tri_ids indirection and 144-B
// instances (src/BVHNode.hpp:9-59, src/TLAS.hpp:5-12) and pays, per interior
// visit, one node load plus two dependent child loads.  Here every interior
// node of the TLAS and of every BLAS becomes ONE 64-byte, 64-byte-aligned
// record holding both child boxes and both child references, so a visit is a
// single cache line; triangles are stored in leaf order (no tri_ids hop) with
// their edges precomputed and a terminator flag, so a child reference is one
// dword and the stack holds one dword per pending subtree.
#ifndef RT_DEVICE_LAYOUT_H
#define RT_DEVICE_LAYOUT_H

#include <stdint.h>

// child / stack reference encoding
#define RT_REF_LEAF 0x80000000u     // bit 31: leaf (else: index into pairs[])
#define RT_REF_INST 0x40000000u     // with LEAF: TLAS leaf, low 30 bits = instance id
#define RT_REF_SENTINEL 0xFFFFFFFFu // stack marker: BLAS finished, back to world space
#define RT_REF_NONE 0xFFFFFFFEu     // empty scene

struct alignas(64) rt_dev_pair { // both children of one interior node, one axis per 16 bytes
  float x[4];    // child0.min.x, child0.max.x, child1.min.x, child1.max.x
  float y[4];    // the same for y: a dwordx4 load puts (min,max) of one child in one
  float z[4];    // aligned register pair, which is what the packed-fp32 slab test wants
  uint32_t ref0; // child references
  uint32_t ref1;
  uint32_t pad0;
  uint32_t pad1;
};

struct alignas(16) rt_dev_tri { // 48 B, leaf order
  float v0[3];
  uint32_t id; // global triangle index reported in hits (tri_ids[slot])
  float e1[3]; // v1 - v0, same fp32 subtraction the shader performs
  uint32_t last; // 1 on the last triangle of its leaf
  float e2[3]; // v2 - v0
  uint32_t pad;
};

struct alignas(64) rt_dev_inst { // 64 B: what a TLAS leaf visit needs
  float c0[3]; // columns of inv_transform (xyz rows)
  uint32_t root_ref; // BLAS root: pair index, or LEAF|first for a one-leaf BLAS
  float c1[3];
  uint32_t pad0;
  float c2[3];
  uint32_t pad1;
  float c3[3];
  uint32_t pad2;
};

// What shading one hit needs from its instance, resolved at upload: the reference reads the
// instance (144 B), then its material table entry, then (Lambert) the texture descriptor, then
// the texel — four dependent loads (RayTracing.slang:103-157, Material.slang).  Here the material
// parameters sit next to the matrices, and a 1x1 texture (every constant-colour Lambert,
// Renderer.cpp:691-703) is decoded once on the host, so shading needs ONE record.
struct alignas(16) rt_dev_inst_shade {
  float transform[16];     // column-major, as rt_blas_instance
  float inv_transform[16];
  float param[4];          // lambert: constant albedo rgb (if tex_w * tex_h == 1) | metal: albedo rgb, fuzz |
                           // dielectric: refraction index | emissive: intensity rgb
  uint32_t mtype;          // rt_material_type
  uint32_t tex_offset;     // lambert with a real texture: byte offset of its RGBA8 texels, and extent
  uint32_t tex_w, tex_h;
};

#define RT_LDS_STACK 24   // stack entries per lane kept in LDS
#define RT_SPILL_STACK 104 // deeper entries, per-lane scratch
#define RT_MAX_STACK (RT_LDS_STACK + RT_SPILL_STACK) // 128, as TLAS.slang:17
#define RT_CACHE_PAIRS 1536u // breadth-first top of the scene kept in LDS (96 KiB)

#endif
