"""Python face of the host renderer (librt_host.so, include/rt_host.h).

`Renderer` keeps the method names of the reference's hlx::Renderer
(src/Renderer.hpp:18-113); all work happens in the C++ library, which in turn
drives the HIP kernels through librt_hip's C ABI.  Nothing here computes.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from . import _types as T

_PKG = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_PKG, "librt_host.so")

OPT_SAMPLES_PER_PIXEL, OPT_MAX_DEPTH, OPT_RNG_MODE, OPT_AABB_MODE = 0, 1, 2, 3
OPT_FRAME_STRIDE, OPT_FRAME_OFFSET, OPT_KERNEL_VARIANT = 4, 5, 6
OPT_ROW_BEGIN, OPT_ROW_END = 7, 8
OPT_TLAS_GPU_MIN_INSTANCES = 9
OPT_ROW_GROUP_STRIDE = 10
OPT_RENDER_FLAGS = 11
PLANE, CUBE, SPHERE = 0, 1, 2

HOST_SYMBOLS = [
    "rth_create", "rth_destroy", "rth_resize", "rth_render", "rth_render_frames", "rth_set_option", "rth_get_frame_index",
    "rth_reset_accumulation", "rth_add_lambert_material", "rth_add_lambert_texture", "rth_add_lambert_image",
    "rth_decode_image", "rth_write_png_rgb8", "rth_import_gltf",
    "rth_add_metal_material", "rth_add_dielectric_material", "rth_add_emissive_material",
    "rth_remove_material", "rth_default_material", "rth_add_blas", "rth_add_blas_mesh",
    "rth_add_blas_instance", "rth_set_blas_instance_transform", "rth_remove_blas",
    "rth_remove_blas_instance", "rth_primitive_blas", "rth_total_triangle_count", "rth_scene_desc",
    "rth_camera_uniforms", "rth_read_image", "rth_wait", "rth_context", "rth_last_error", "rth_trs",
    "rth_make_primitive", "rth_make_dragon", "rth_make_subdivided", "rth_load_obj", "rth_load_scene_file", "rth_mesh_free", "rth_mesh_vertex_count",
    "rth_mesh_index_count", "rth_mesh_positions", "rth_mesh_normals", "rth_mesh_uvs",
    "rth_mesh_indices", "rth_build_blas", "rth_refit_blas", "rth_build_tlas", "rth_tlas_leaves", "rth_tlas_cluster",
    "rth_last_tlas_build", "rth_mat4_inverse",
]


class CameraC(C.Structure):  # rth_camera
    _fields_ = [("position", C.c_float * 3), ("look_at", C.c_float * 3), ("v_up", C.c_float * 3),
                ("fov", C.c_float), ("defocus_angle", C.c_float), ("focus_dist", C.c_float),
                ("changed", C.c_int32)]


class SceneInfoC(C.Structure):  # rth_scene_info
    _fields_ = [("camera", CameraC), ("screen_width", C.c_uint32), ("screen_height", C.c_uint32),
                ("samples_per_pixel", C.c_uint32), ("max_depth", C.c_uint32), ("model_count", C.c_uint32)]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(HOST_LIB_PATH):
        raise ImportError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
    _ffi.lib()  # librt_hip.so first (rpath $ORIGIN resolves it too)
    L = C.CDLL(HOST_LIB_PATH)
    vp, u32, u64, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
    MH = T.MaterialHandle
    L.rth_create.argtypes = [C.c_int, u32, u32, C.POINTER(vp)]
    L.rth_destroy.argtypes = [vp]; L.rth_destroy.restype = None
    L.rth_resize.argtypes = [vp, u32, u32]
    L.rth_render.argtypes = [vp, C.POINTER(CameraC)]
    L.rth_render_frames.argtypes = [vp, C.POINTER(CameraC), u32]
    L.rth_set_option.argtypes = [vp, C.c_int, u32]
    L.rth_get_frame_index.argtypes = [vp]; L.rth_get_frame_index.restype = u32
    L.rth_reset_accumulation.argtypes = [vp]; L.rth_reset_accumulation.restype = None
    L.rth_add_lambert_material.argtypes = [vp, vp]; L.rth_add_lambert_material.restype = MH
    L.rth_add_lambert_texture.argtypes = [vp, C.c_int32, C.c_int32, vp]; L.rth_add_lambert_texture.restype = MH
    L.rth_add_lambert_image.argtypes = [vp, C.c_char_p]; L.rth_add_lambert_image.restype = MH
    L.rth_decode_image.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp, C.c_uint64, C.c_char_p, C.c_size_t]
    L.rth_decode_image.restype = C.c_int
    L.rth_write_png_rgb8.argtypes = [C.c_char_p, u32, u32, vp]; L.rth_write_png_rgb8.restype = C.c_int
    L.rth_import_gltf.argtypes = [vp, C.c_char_p, vp, vp, C.c_char_p, C.c_size_t]; L.rth_import_gltf.restype = C.c_int64
    L.rth_add_metal_material.argtypes = [vp, vp, f32]; L.rth_add_metal_material.restype = MH
    L.rth_add_dielectric_material.argtypes = [vp, f32]; L.rth_add_dielectric_material.restype = MH
    L.rth_add_emissive_material.argtypes = [vp, vp]; L.rth_add_emissive_material.restype = MH
    L.rth_remove_material.argtypes = [vp, MH]; L.rth_remove_material.restype = None
    L.rth_default_material.argtypes = [vp]; L.rth_default_material.restype = MH
    L.rth_add_blas.argtypes = [vp, vp, vp, vp, vp, u64]; L.rth_add_blas.restype = u32
    L.rth_add_blas_mesh.argtypes = [vp, vp]; L.rth_add_blas_mesh.restype = u32
    L.rth_add_blas_instance.argtypes = [vp, u32, vp, MH]; L.rth_add_blas_instance.restype = u32
    L.rth_set_blas_instance_transform.argtypes = [vp, u32, vp]; L.rth_set_blas_instance_transform.restype = None
    L.rth_remove_blas.argtypes = [vp, u32]; L.rth_remove_blas.restype = None
    L.rth_remove_blas_instance.argtypes = [vp, u32]; L.rth_remove_blas_instance.restype = None
    L.rth_primitive_blas.argtypes = [vp, C.c_int]; L.rth_primitive_blas.restype = u32
    L.rth_total_triangle_count.argtypes = [vp]; L.rth_total_triangle_count.restype = u32
    L.rth_scene_desc.argtypes = [vp, C.POINTER(T.SceneDesc)]
    L.rth_camera_uniforms.argtypes = [vp, C.POINTER(CameraC), C.POINTER(T.CameraUniforms)]
    L.rth_read_image.argtypes = [vp, vp]
    L.rth_wait.argtypes = [vp]
    L.rth_context.argtypes = [vp]; L.rth_context.restype = vp
    L.rth_last_error.argtypes = [vp]; L.rth_last_error.restype = C.c_char_p
    L.rth_trs.argtypes = [vp, f32, vp, vp, vp]; L.rth_trs.restype = None
    L.rth_make_primitive.argtypes = [C.c_int]; L.rth_make_primitive.restype = vp
    L.rth_make_dragon.argtypes = [u32, u32, u32]; L.rth_make_dragon.restype = vp
    L.rth_make_subdivided.argtypes = [vp, u32, f32, u32]; L.rth_make_subdivided.restype = vp
    L.rth_load_obj.argtypes = [C.c_char_p, C.POINTER(vp), vp, vp, C.c_int, C.c_char_p, C.c_int]
    L.rth_load_scene_file.argtypes = [vp, C.c_char_p, C.POINTER(SceneInfoC), C.c_char_p, C.c_int]
    L.rth_load_scene_file.restype = C.c_int64
    L.rth_mesh_free.argtypes = [vp]; L.rth_mesh_free.restype = None
    for n in ("rth_mesh_vertex_count", "rth_mesh_index_count"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = u32
    for n in ("rth_mesh_positions", "rth_mesh_normals", "rth_mesh_uvs", "rth_mesh_indices"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = vp
    L.rth_build_blas.argtypes = [vp, vp, vp, vp, u32, u32]; L.rth_build_blas.restype = u32
    L.rth_refit_blas.argtypes = [vp, u32, vp, vp]; L.rth_refit_blas.restype = None
    L.rth_build_tlas.argtypes = [vp, vp, vp, u32, vp, vp]; L.rth_build_tlas.restype = u32
    L.rth_tlas_leaves.argtypes = [vp, vp, vp, u32, vp, vp]; L.rth_tlas_leaves.restype = None
    L.rth_tlas_cluster.argtypes = [vp, u32]; L.rth_tlas_cluster.restype = u32
    L.rth_last_tlas_build.argtypes = [vp, C.POINTER(C.c_float)]; L.rth_last_tlas_build.restype = C.c_int
    L.rth_mat4_inverse.argtypes = [vp, vp]; L.rth_mat4_inverse.restype = None
    _LIB = L
    return L


def _f32(x):
    return np.ascontiguousarray(x, np.float32)


def trs(position=(0, 0, 0), angle=0.0, axis=(0, 1, 0), scale=(1, 1, 1)):
    """Transform::get_mat4 (src/Transform.hpp:15-18) evaluated by the host library."""
    out = np.zeros(16, np.float32)
    lib().rth_trs(T.ptr(_f32(position)), np.float32(angle), T.ptr(_f32(axis)), T.ptr(_f32(scale)), T.ptr(out))
    return out


def decode_image(path):
    """The host library's image decoder (stbi_load(path, .., 4) semantics): uint8 array [h, w, 4]."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    err = C.create_string_buffer(512)
    if lib().rth_decode_image(os.fsencode(path), C.byref(w), C.byref(h), None, 0, err, len(err)) != 0:
        raise RendererError(err.value.decode(errors="replace"))
    px = np.zeros((h.value, w.value, 4), np.uint8)
    if lib().rth_decode_image(os.fsencode(path), C.byref(w), C.byref(h), T.ptr(px), px.size, err, len(err)) != 0:
        raise RendererError(err.value.decode(errors="replace"))
    return px


def write_png_rgb8(path, rgb):
    """The command line's PNG writer: uint8 array [h, w, 3]."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    if lib().rth_write_png_rgb8(os.fsencode(path), rgb.shape[1], rgb.shape[0], T.ptr(rgb)) != 0:
        raise RendererError("cannot write " + str(path))


class Mesh:
    """A mesh owned by the host library (primitives, dragon stand-in, OBJ groups)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def primitive(cls, which):
        return cls(lib().rth_make_primitive(which))

    @classmethod
    def dragon(cls, u_seg=1700, v_seg=256, seed=1234):
        return cls(lib().rth_make_dragon(u_seg, v_seg, seed))

    def subdivided(self, levels, amplitude=0.02, seed=1234):
        """`levels` rounds of 1:4 subdivision of this mesh + seeded displacement along the normals (meshes.h)."""
        h = lib().rth_make_subdivided(self._h, levels, amplitude, seed)
        if not h:
            raise RendererError("rth_make_subdivided failed")
        return Mesh(h)

    @staticmethod
    def load_obj(path, max_groups=64):
        """-> list of (Mesh, kd_rgb, has_material) per material group."""
        L = lib()
        handles = (C.c_void_p * max_groups)()
        kd = np.zeros((max_groups, 3), np.float32)
        has = np.zeros(max_groups, np.int32)
        err = C.create_string_buffer(512)
        n = L.rth_load_obj(os.fsencode(path), handles, T.ptr(kd), T.ptr(has), max_groups, err, 512)
        if n < 0:
            raise IOError(err.value.decode(errors="replace"))
        return [(Mesh(handles[i]), kd[i].copy(), bool(has[i])) for i in range(min(n, max_groups))]

    def _arr(self, fn, count, dtype, width):
        p = fn(self._h)
        if not p or count == 0:
            return np.zeros((0, width) if width > 1 else 0, dtype)
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float if dtype == np.float32 else C.c_uint32)),
                                  shape=(count * width,)).copy()
        return a.reshape(-1, width) if width > 1 else a

    @property
    def positions(self):
        return self._arr(lib().rth_mesh_positions, lib().rth_mesh_vertex_count(self._h), np.float32, 3)

    @property
    def normals(self):
        return self._arr(lib().rth_mesh_normals, lib().rth_mesh_vertex_count(self._h), np.float32, 3)

    @property
    def uvs(self):
        return self._arr(lib().rth_mesh_uvs, lib().rth_mesh_vertex_count(self._h), np.float32, 2)

    @property
    def indices(self):
        return self._arr(lib().rth_mesh_indices, lib().rth_mesh_index_count(self._h), np.uint32, 1)

    @property
    def triangle_count(self):
        return lib().rth_mesh_index_count(self._h) // 3

    def arrays(self):
        return self.positions, self.normals, self.uvs, self.indices

    def __del__(self):
        try:
            if self._h:
                lib().rth_mesh_free(self._h)
                self._h = None
        except Exception:
            pass


class Camera:
    """hlx::Camera's render-relevant fields (src/Camera.hpp:20-25)."""

    def __init__(self, position, look_at, v_up=(0, 1, 0), fov=90.0, defocus_angle=0.0, focus_dist=1.0):
        self.c = CameraC()
        self.c.position[:] = [float(v) for v in position]
        self.c.look_at[:] = [float(v) for v in look_at]
        self.c.v_up[:] = [float(v) for v in v_up]
        self.c.fov, self.c.defocus_angle, self.c.focus_dist = fov, defocus_angle, focus_dist
        self.c.changed = 0


class RendererError(RuntimeError):
    pass


class Renderer:
    """hlx::Renderer over HIP.  device=None creates a host-only renderer (scene
    assembly and the CPU builders; render() raises)."""

    def __init__(self, width, height, device=0):
        self._h = C.c_void_p()
        self.width, self.height = width, height
        rc = lib().rth_create(-1 if device is None else device, width, height, C.byref(self._h))
        if rc != 0:
            raise RendererError(f"rth_create failed ({rc}): {_ffi.lib().rt_last_error_string().decode()}")

    def _ok(self, rc):
        if rc != 0:
            raise RendererError(f"error {rc}: {lib().rth_last_error(self._h).decode(errors='replace')}")

    def shutdown(self):
        if self._h:
            lib().rth_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    # -- reference interface --------------------------------------------------
    def resize(self, width, height):
        self._ok(lib().rth_resize(self._h, width, height))
        self.width, self.height = width, height

    def render(self, camera):
        self._ok(lib().rth_render(self._h, C.byref(camera.c)))

    def render_frames(self, camera, frames):
        """`frames` progressive frames (the reference's frame loop, PathTracer.cpp:396-411) in one device call."""
        self._ok(lib().rth_render_frames(self._h, C.byref(camera.c), frames))

    def add_lambert_material(self, albedo):
        return lib().rth_add_lambert_material(self._h, T.ptr(_f32(albedo)))

    def add_lambert_texture(self, width, height, rgba8):
        px = np.ascontiguousarray(rgba8, np.uint8)
        return lib().rth_add_lambert_texture(self._h, width, height, T.ptr(px))

    def add_lambert_image(self, path):
        """add_lambert_material(file_path), Renderer.cpp:705-716 (PNG / binary PNM)."""
        return lib().rth_add_lambert_image(self._h, os.fsencode(path))

    def import_gltf(self, path, parent_global=None):
        """load_gltf_scene + update_transforms (SceneGraph.cpp:168-407, 544-566): dict of counts."""
        counts = np.zeros(4, np.uint32)
        err = C.create_string_buffer(512)
        pg = None if parent_global is None else T.ptr(np.ascontiguousarray(parent_global, np.float32).reshape(16))
        n = lib().rth_import_gltf(self._h, os.fsencode(path), pg, T.ptr(counts), err, len(err))
        if n < 0:
            raise RendererError(err.value.decode(errors="replace"))
        return {"triangles": int(n), "nodes": int(counts[0]), "meshes": int(counts[1]),
                "primitives": int(counts[2]), "materials": int(counts[3])}

    def add_metal_material(self, albedo, fuzz):
        return lib().rth_add_metal_material(self._h, T.ptr(_f32(albedo)), fuzz)

    def add_dielectric_material(self, refractive_index):
        return lib().rth_add_dielectric_material(self._h, refractive_index)

    def add_emissive_material(self, intensity):
        return lib().rth_add_emissive_material(self._h, T.ptr(_f32(intensity)))

    def remove_material(self, handle):
        lib().rth_remove_material(self._h, handle)

    @property
    def default_material(self):
        return lib().rth_default_material(self._h)

    def add_blas(self, positions, normals=None, uvs=None, indices=None):
        if isinstance(positions, Mesh):
            return lib().rth_add_blas_mesh(self._h, positions._h)
        p, n, u = _f32(positions), _f32(normals), _f32(uvs)
        i = np.ascontiguousarray(indices, np.uint32)
        return lib().rth_add_blas(self._h, T.ptr(p), T.ptr(n), T.ptr(u), T.ptr(i), i.size)

    def add_blas_instance(self, blas_index, transform, material):
        return lib().rth_add_blas_instance(self._h, blas_index, T.ptr(_f32(transform).reshape(16)), material)

    def set_blas_instance_transform(self, instance_id, transform):
        lib().rth_set_blas_instance_transform(self._h, instance_id, T.ptr(_f32(transform).reshape(16)))

    def remove_blas(self, blas_id):
        lib().rth_remove_blas(self._h, blas_id)

    def remove_blas_instance(self, instance_id):
        lib().rth_remove_blas_instance(self._h, instance_id)

    @property
    def plane_blas_index(self):
        return lib().rth_primitive_blas(self._h, PLANE)

    @property
    def cube_blas_index(self):
        return lib().rth_primitive_blas(self._h, CUBE)

    @property
    def sphere_blas_index(self):
        return lib().rth_primitive_blas(self._h, SPHERE)

    @property
    def total_triangle_count(self):
        return lib().rth_total_triangle_count(self._h)

    @property
    def frame_index(self):
        return lib().rth_get_frame_index(self._h)

    # -- beyond the reference -------------------------------------------------
    def load_scene_file(self, json_path):
        """Loads a reference scene file (README.md:53-68) into this renderer.  Returns
        (camera, info) where info has screen_width/height, samples_per_pixel, max_depth,
        model_count and triangles."""
        info = SceneInfoC()
        err = C.create_string_buffer(512)
        n = lib().rth_load_scene_file(self._h, os.fsencode(json_path), C.byref(info), err, len(err))
        if n < 0:
            raise RendererError(err.value.decode(errors="replace"))
        cam = Camera(info.camera.position, info.camera.look_at, info.camera.v_up, info.camera.fov,
                     info.camera.defocus_angle, info.camera.focus_dist)
        return cam, {"screen_width": info.screen_width, "screen_height": info.screen_height,
                     "samples_per_pixel": info.samples_per_pixel, "max_depth": info.max_depth,
                     "model_count": info.model_count, "triangles": int(n)}

    def set_option(self, option, value):
        self._ok(lib().rth_set_option(self._h, option, value))

    def last_tlas_build(self):
        """(ran_on_gpu, kernel_ms) of the last TLAS build (Renderer.cpp:994-1015)."""
        ms = C.c_float(0)
        on_gpu = lib().rth_last_tlas_build(self._h, C.byref(ms))
        return bool(on_gpu), float(ms.value)

    def reset_accumulation(self):
        lib().rth_reset_accumulation(self._h)

    def scene_desc(self):
        d = T.SceneDesc()
        self._ok(lib().rth_scene_desc(self._h, C.byref(d)))
        return d

    def scene_arrays(self):
        """numpy copies of the published arrays (for parity tests and the CPU baseline)."""
        d = self.scene_desc()

        def view(p, n, dt):
            if not p or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(p)
            return np.frombuffer(buf, dtype=dt, count=n).copy()
        return T.SceneArrays(
            view(d.tri_geom, d.tri_capacity, T.geom_dtype), view(d.tri_shading, d.tri_capacity, T.shading_dtype),
            view(d.tri_ids, d.tri_capacity, np.uint32), view(d.bvh_nodes, d.bvh_node_count, T.bvh_node_dtype),
            view(d.blas, d.blas_count, T.blas_dtype), view(d.instances, d.instance_count, T.instance_dtype),
            view(d.tlas_nodes, max(d.tlas_node_count, 1) if d.tlas_nodes else 0, T.tlas_node_dtype), d.tlas_node_count,
            view(d.lamberts, d.lambert_count, np.uint32), view(d.metals, d.metal_count * 4, np.float32),
            view(d.dielectrics, d.dielectric_count, np.float32), view(d.emissives, d.emissive_count * 4, np.float32),
            view(d.textures, d.texture_count, T.texture_dtype), view(d.texels, d.texel_bytes, np.uint8))

    def camera_uniforms(self, camera):
        out = T.CameraUniforms()
        self._ok(lib().rth_camera_uniforms(self._h, C.byref(camera.c), C.byref(out)))
        return out

    def read_image(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._ok(lib().rth_read_image(self._h, T.ptr(out)))
        return out

    def wait(self):
        self._ok(lib().rth_wait(self._h))

    def hip_context_handle(self):
        return lib().rth_context(self._h)
