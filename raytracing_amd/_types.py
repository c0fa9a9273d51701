"""ctypes mirrors of include/rt_types.h (same field order, same sizes).

Each class cites the reference layout it reproduces; sizes are asserted at
import so a drift between the header and this file fails loudly.
"""
import ctypes as C

import numpy as np

u32, f32, u64 = C.c_uint32, C.c_float, C.c_uint64


class TriangleGeom(C.Structure):  # src/Triangle.hpp:9-18
    _fields_ = [("v0", f32 * 4), ("v1", f32 * 4), ("v2", f32 * 4)]


class TriangleShading(C.Structure):  # src/Triangle.hpp:20-35
    _fields_ = [("n0", f32 * 4), ("n1", f32 * 4), ("n2", f32 * 4),
                ("uv0", f32 * 2), ("uv1", f32 * 2), ("uv2", f32 * 2),
                ("padding", f32 * 2)]


class BvhNode(C.Structure):  # src/BVHNode.hpp:9-15
    _fields_ = [("aabb_min", f32 * 3), ("local_left_first", u32),
                ("aabb_max", f32 * 3), ("tri_count", u32)]


class Blas(C.Structure):  # src/BVHNode.hpp:17-39
    _fields_ = [("bvh_nodes_offset", u32), ("nodes_count", u32),
                ("tri_count", u32), ("padding", u32)]


class MaterialHandle(C.Structure):  # src/Material.hpp:26-29
    _fields_ = [("index", u32), ("type", u32)]


class BlasInstance(C.Structure):  # src/BVHNode.hpp:49-59
    _fields_ = [("transform", f32 * 16), ("inv_transform", f32 * 16),
                ("material", MaterialHandle), ("blas_id", u32), ("padding", u32)]


class TlasNode(C.Structure):  # src/TLAS.hpp:5-12
    _fields_ = [("aabb_min", f32 * 3), ("left_right", u32),
                ("aabb_max", f32 * 3), ("blas_instance_idx", u32)]


class TextureDesc(C.Structure):
    _fields_ = [("offset", u32), ("width", u32), ("height", u32), ("padding", u32)]


class CameraUniforms(C.Structure):  # src/Renderer.cpp:30-34
    _fields_ = [("pixel00_loc", f32 * 4), ("pixel_delta_u", f32 * 4),
                ("pixel_delta_v", f32 * 4), ("camera_center", f32 * 4),
                ("defocus_disk_u", f32 * 4), ("defocus_disk_v", f32 * 4)]


class SceneDesc(C.Structure):  # src/Renderer.cpp:35-45 (the 11 arrays)
    _fields_ = [
        ("tri_geom", C.c_void_p), ("tri_shading", C.c_void_p), ("tri_ids", C.c_void_p),
        ("tri_capacity", u32), ("bvh_node_count", u32),
        ("bvh_nodes", C.c_void_p), ("blas", C.c_void_p),
        ("blas_count", u32), ("instance_count", u32),
        ("instances", C.c_void_p), ("tlas_nodes", C.c_void_p),
        ("tlas_node_count", u32), ("lambert_count", u32),
        ("lamberts", C.c_void_p), ("metals", C.c_void_p),
        ("dielectrics", C.c_void_p), ("emissives", C.c_void_p),
        ("metal_count", u32), ("dielectric_count", u32),
        ("emissive_count", u32), ("texture_count", u32),
        ("textures", C.c_void_p), ("texels", C.c_void_p), ("texel_bytes", u64),
    ]


class RenderParams(C.Structure):  # src/Renderer.cpp:48-62
    _fields_ = [("image_width", u32), ("image_height", u32), ("frame_index", u32),
                ("accum_count", u32), ("sqrt_spp", u32), ("max_depth", u32),
                ("rng_mode", u32), ("aabb_mode", u32), ("row_begin", u32),
                ("row_end", u32), ("row_group_stride", u32), ("flags", u32)]


MATERIAL_LAMBERT, MATERIAL_METAL, MATERIAL_DIELECTRIC, MATERIAL_EMISSIVE, MATERIAL_NONE = 0, 1, 2, 3, 4
RNG_WANG, RNG_PCG = 0, 1
AABB_DIV, AABB_RCP = 0, 1
RENDER_DIELECTRIC_ALWAYS_DRAWS = 1  # rt_render_params.flags
RENDER_SKY_WHITE_HORIZON = 2
INVALID_ID = 0xFFFFFFFF

# numpy views of the same layouts (arrays of these are passed by pointer)
geom_dtype = np.dtype([("v0", "<f4", 4), ("v1", "<f4", 4), ("v2", "<f4", 4)])
shading_dtype = np.dtype([("n0", "<f4", 4), ("n1", "<f4", 4), ("n2", "<f4", 4),
                          ("uv0", "<f4", 2), ("uv1", "<f4", 2), ("uv2", "<f4", 2),
                          ("padding", "<f4", 2)])
bvh_node_dtype = np.dtype([("aabb_min", "<f4", 3), ("local_left_first", "<u4"),
                           ("aabb_max", "<f4", 3), ("tri_count", "<u4")])
blas_dtype = np.dtype([("bvh_nodes_offset", "<u4"), ("nodes_count", "<u4"),
                       ("tri_count", "<u4"), ("padding", "<u4")])
instance_dtype = np.dtype([("transform", "<f4", 16), ("inv_transform", "<f4", 16),
                           ("mat_index", "<u4"), ("mat_type", "<u4"),
                           ("blas_id", "<u4"), ("padding", "<u4")])
tlas_node_dtype = np.dtype([("aabb_min", "<f4", 3), ("left_right", "<u4"),
                            ("aabb_max", "<f4", 3), ("blas_instance_idx", "<u4")])
texture_dtype = np.dtype([("offset", "<u4"), ("width", "<u4"), ("height", "<u4"),
                          ("padding", "<u4")])
ray_dtype = np.dtype([("origin", "<f4", 3), ("t_min", "<f4"),
                      ("direction", "<f4", 3), ("t_max", "<f4")])
hit_dtype = np.dtype([("t", "<f4"), ("tri_id", "<u4"), ("inst_id", "<u4"),
                      ("u", "<f4"), ("v", "<f4"), ("pad", "<u4", 3)])

assert C.sizeof(TriangleGeom) == 48 == geom_dtype.itemsize
assert C.sizeof(TriangleShading) == 80 == shading_dtype.itemsize
assert C.sizeof(BvhNode) == 32 == bvh_node_dtype.itemsize
assert C.sizeof(Blas) == 16 == blas_dtype.itemsize
assert C.sizeof(BlasInstance) == 144 == instance_dtype.itemsize
assert C.sizeof(TlasNode) == 32 == tlas_node_dtype.itemsize
assert C.sizeof(CameraUniforms) == 96
assert C.sizeof(RenderParams) == 48
assert C.sizeof(SceneDesc) == 152
assert ray_dtype.itemsize == 32 and hit_dtype.itemsize == 32


def ptr(a):
    """void* of a numpy array (None -> NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SceneArrays:
    """The 11 reference arrays + textures as numpy arrays, and the matching
    rt_scene_desc.  Keeps the arrays alive for as long as the desc is used."""

    def __init__(self, tri_geom, tri_shading, tri_ids, bvh_nodes, blas, instances,
                 tlas_nodes, tlas_node_count, lamberts=None, metals=None,
                 dielectrics=None, emissives=None, textures=None, texels=None):
        z = lambda dt: np.zeros(0, dtype=dt)
        self.tri_geom, self.tri_shading, self.tri_ids = tri_geom, tri_shading, tri_ids
        self.bvh_nodes, self.blas, self.instances = bvh_nodes, blas, instances
        self.tlas_nodes, self.tlas_node_count = tlas_nodes, int(tlas_node_count)
        self.lamberts = z("<u4") if lamberts is None else np.ascontiguousarray(lamberts, "<u4")
        self.metals = np.zeros((0, 4), "<f4") if metals is None else np.ascontiguousarray(metals, "<f4").reshape(-1, 4)
        self.dielectrics = z("<f4") if dielectrics is None else np.ascontiguousarray(dielectrics, "<f4")
        self.emissives = np.zeros((0, 4), "<f4") if emissives is None else np.ascontiguousarray(emissives, "<f4").reshape(-1, 4)
        self.textures = z(texture_dtype) if textures is None else textures
        self.texels = z("u1") if texels is None else np.ascontiguousarray(texels, "u1")

    def desc(self):
        d = SceneDesc()
        d.tri_geom, d.tri_shading, d.tri_ids = ptr(self.tri_geom), ptr(self.tri_shading), ptr(self.tri_ids)
        d.tri_capacity = len(self.tri_ids)
        d.bvh_node_count = len(self.bvh_nodes)
        d.bvh_nodes, d.blas = ptr(self.bvh_nodes), ptr(self.blas)
        d.blas_count, d.instance_count = len(self.blas), len(self.instances)
        d.instances, d.tlas_nodes = ptr(self.instances), ptr(self.tlas_nodes)
        d.tlas_node_count = self.tlas_node_count
        d.lambert_count = len(self.lamberts)
        d.lamberts, d.metals = ptr(self.lamberts), ptr(self.metals)
        d.dielectrics, d.emissives = ptr(self.dielectrics), ptr(self.emissives)
        d.metal_count, d.dielectric_count = len(self.metals), len(self.dielectrics)
        d.emissive_count, d.texture_count = len(self.emissives), len(self.textures)
        d.textures, d.texels = ptr(self.textures), ptr(self.texels)
        d.texel_bytes = self.texels.nbytes
        return d
