"""ctypes binding of librt_hip.so (include/rt_hip.h).

The library is built in-tree by __graft_entry__.build() / raytracing_amd/csrc/Makefile.
Loading fails loudly when it is missing; there is no Python or CPU fallback for
any entry point.
"""
import ctypes as C
import os

import numpy as np

from . import _types as T

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "librt_hip.so")

# every symbol include/rt_hip.h declares
SYMBOLS = [
    "rt_create", "rt_destroy", "rt_upload_scene", "rt_update_instances", "rt_set_camera",
    "rt_resize", "rt_bind_accum", "rt_clear_accum", "rt_render", "rt_render_frames", "rt_trace_closest",
    "rt_trace_closest_device", "rt_read_accum", "rt_accum_device_ptr", "rt_sync",
    "rt_get_stats", "rt_reset_stats", "rt_set_stream", "rt_set_kernel_variant", "rt_set_wavefront_budget", "rt_debug_probe_walk", "rt_debug_fast_div", "rt_debug_wavefront_tuning", "rt_debug_wavefront_compaction", "rt_debug_wavefront_record_cache", "rt_debug_wavefront_persistent", "rt_debug_wavefront_bands", "rt_debug_wavefront_counts", "rt_debug_fail_wavefront_allocs", "rt_debug_wavefront_timing", "rt_build_tlas_cluster",
    "rt_build_tlas_cluster_noalloc",
    "rt_last_error_string",
    "rt_device_count",
    "rt_mgpu_partition", "rt_mgpu_create", "rt_mgpu_destroy", "rt_mgpu_ranks", "rt_mgpu_uses_rccl", "rt_mgpu_context",
    "rt_mgpu_upload_scene", "rt_mgpu_update_instances", "rt_mgpu_set_camera", "rt_mgpu_set_kernel_variant", "rt_mgpu_resize",
    "rt_mgpu_clear_accum", "rt_mgpu_render_frames", "rt_mgpu_read_accum", "rt_mgpu_get_stats",
]


KERNEL_VARIANT_AUTO = 0xFFFFFFFF  # RT_KERNEL_VARIANT_AUTO: rt_render chooses (a new context's state)


class Stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("last_kernel_ms", C.c_double),
                ("total_kernel_ms", C.c_double), ("launches", C.c_uint64),
                ("stack_depth_needed", C.c_uint32), ("last_kernel", C.c_uint32),
                ("device_bytes", C.c_uint64)]


class MgpuShare(C.Structure):  # rt_mgpu_share
    _fields_ = [(n, C.c_uint32) for n in ("groups_total", "groups", "groups_per_rank", "rows", "row_begin", "row_end",
                                          "row_group_stride")]


EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librt_hip error {code}: {msg}")
        self.code = code


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                          "(make -C raytracing_amd/csrc); there is no fallback path")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 (same
    # SONAME as /opt/rocm's).  Whichever is mapped first serves both, and a
    # process that maps ROCm's first and then initialises torch's device state
    # ends up with a runtime that reports no device to this library.  Import
    # torch first (when present) so both always share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_destroy.argtypes = [vp]
    L.rt_upload_scene.argtypes = [vp, C.POINTER(T.SceneDesc)]
    L.rt_update_instances.argtypes = [vp, vp, u32, vp, u32]
    L.rt_set_camera.argtypes = [vp, C.POINTER(T.CameraUniforms)]
    L.rt_resize.argtypes = [vp, u32, u32]
    L.rt_bind_accum.argtypes = [vp, vp, u32, u32]
    L.rt_clear_accum.argtypes = [vp]
    L.rt_render.argtypes = [vp, C.POINTER(T.RenderParams)]
    L.rt_render_frames.argtypes = [vp, C.POINTER(T.RenderParams), u32]
    L.rt_trace_closest.argtypes = [vp, vp, u64, vp, u32]
    L.rt_trace_closest_device.argtypes = [vp, vp, u64, vp, u32]
    L.rt_read_accum.argtypes = [vp, vp]
    L.rt_accum_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.rt_sync.argtypes = [vp]
    L.rt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_reset_stats.argtypes = [vp]
    L.rt_set_kernel_variant.argtypes = [vp, u32]
    L.rt_set_stream.argtypes = [vp, vp]
    L.rt_set_wavefront_budget.argtypes = [vp, u32, u64]
    L.rt_debug_fast_div.argtypes = [vp, vp, vp, C.c_uint64, vp, vp]
    L.rt_build_tlas_cluster.argtypes = [vp, vp, u32, vp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.rt_build_tlas_cluster_noalloc.argtypes = [vp, vp, u32, vp, C.POINTER(u32)]
    L.rt_debug_wavefront_tuning.argtypes = [vp, u32, u32, u32, u32]
    L.rt_debug_wavefront_compaction.argtypes = [vp, u32]
    L.rt_debug_wavefront_record_cache.argtypes = [vp, u32]
    L.rt_debug_wavefront_persistent.argtypes = [vp, u32]
    L.rt_debug_wavefront_bands.argtypes = [vp, u32]
    L.rt_debug_wavefront_counts.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4
    L.rt_debug_fail_wavefront_allocs.argtypes = [vp, u32]
    L.rt_debug_wavefront_timing.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(u64)]
    L.rt_debug_probe_walk.argtypes = [vp, u32, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.rt_last_error_string.restype = C.c_char_p
    L.rt_device_count.restype = C.c_int
    for name in SYMBOLS:
        if name not in ("rt_last_error_string",):
            getattr(L, name).restype = C.c_int
    L.rt_mgpu_partition.argtypes = [u32, u32, u32, C.POINTER(MgpuShare)]
    L.rt_mgpu_create.argtypes = [C.POINTER(C.c_int), u32, u32, C.POINTER(vp)]
    L.rt_mgpu_destroy.argtypes = [vp]
    L.rt_mgpu_ranks.argtypes = [vp]; L.rt_mgpu_ranks.restype = u32
    L.rt_mgpu_uses_rccl.argtypes = [vp]
    L.rt_mgpu_context.argtypes = [vp, u32]; L.rt_mgpu_context.restype = vp
    L.rt_mgpu_upload_scene.argtypes = [vp, C.POINTER(T.SceneDesc)]
    L.rt_mgpu_update_instances.argtypes = [vp, vp, u32, vp, u32]
    L.rt_mgpu_set_camera.argtypes = [vp, C.POINTER(T.CameraUniforms)]
    L.rt_mgpu_set_kernel_variant.argtypes = [vp, u32]
    L.rt_mgpu_resize.argtypes = [vp, u32, u32]
    L.rt_mgpu_clear_accum.argtypes = [vp]
    L.rt_mgpu_render_frames.argtypes = [vp, C.POINTER(T.RenderParams), u32]
    L.rt_mgpu_read_accum.argtypes = [vp, u32, vp]
    L.rt_mgpu_get_stats.argtypes = [vp, u32, C.POINTER(Stats)]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise RtError(rc, lib().rt_last_error_string().decode(errors="replace"))


class HipContext:
    """One rt_ctx: one GPU, one host thread."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().rt_create(device, C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene):
        """scene: _types.SceneArrays (the 11 reference arrays + textures)."""
        d = scene.desc()
        _check(lib().rt_upload_scene(self._h, C.byref(d)))

    def update_instances(self, instances, tlas_nodes, tlas_node_count):
        _check(lib().rt_update_instances(self._h, T.ptr(instances), len(instances),
                                         T.ptr(tlas_nodes), int(tlas_node_count)))

    def debug_fast_div(self, n, d):
        """(q_fast, q_ieee) of the slab test's exact fast division, elementwise (test hook)."""
        import numpy as np
        n = np.ascontiguousarray(n, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        assert n.shape == d.shape and n.ndim == 1
        q_fast = np.empty_like(n)
        q_ieee = np.empty_like(n)
        _check(lib().rt_debug_fast_div(self._h, T.ptr(n), T.ptr(d), n.size, T.ptr(q_fast), T.ptr(q_ieee)))
        return q_fast, q_ieee

    def build_tlas_cluster(self, leaves):
        """TLAS::build's clustering loop (TLAS.cpp:38-58) on the GPU: (nodes[2n], node_count, kernel_ms)."""
        import numpy as np
        leaves = np.ascontiguousarray(leaves, T.tlas_node_dtype)
        out = np.zeros(2 * max(leaves.size, 1), T.tlas_node_dtype)
        count, ms = C.c_uint32(0), C.c_float(0)
        _check(lib().rt_build_tlas_cluster(self._h, T.ptr(leaves), leaves.size, T.ptr(out), C.byref(count),
                                           C.byref(ms)))
        return out, int(count.value), float(ms.value)

    def build_tlas_cluster_noalloc(self, leaves):
        """rt_build_tlas_cluster_noalloc (static device scratch, no allocation): (nodes[2n], node_count)."""
        leaves = np.ascontiguousarray(leaves, T.tlas_node_dtype)
        out = np.zeros(2 * max(leaves.size, 1), T.tlas_node_dtype)
        count = C.c_uint32(0)
        _check(lib().rt_build_tlas_cluster_noalloc(self._h, T.ptr(leaves), leaves.size, T.ptr(out), C.byref(count)))
        return out, int(count.value)

    def set_camera(self, cam):
        _check(lib().rt_set_camera(self._h, C.byref(cam)))

    def resize(self, width, height):
        _check(lib().rt_resize(self._h, width, height))

    def bind_accum(self, device_ptr, width, height):
        _check(lib().rt_bind_accum(self._h, C.c_void_p(device_ptr), width, height))

    def clear_accum(self):
        _check(lib().rt_clear_accum(self._h))

    def render(self, params):
        _check(lib().rt_render(self._h, C.byref(params)))

    def render_frames(self, params, frame_count):
        _check(lib().rt_render_frames(self._h, C.byref(params), frame_count))

    def trace_closest(self, rays, aabb_mode=T.AABB_DIV):
        rays = np.ascontiguousarray(rays, T.ray_dtype)
        hits = np.zeros(len(rays), T.hit_dtype)
        _check(lib().rt_trace_closest(self._h, T.ptr(rays), len(rays), T.ptr(hits), aabb_mode))
        return hits

    def trace_closest_device(self, rays_ptr, n, hits_ptr, aabb_mode=T.AABB_DIV):
        _check(lib().rt_trace_closest_device(self._h, C.c_void_p(rays_ptr), n, C.c_void_p(hits_ptr), aabb_mode))

    def read_accum(self, width, height):
        out = np.zeros((height, width, 4), np.float32)
        _check(lib().rt_read_accum(self._h, T.ptr(out)))
        return out

    def accum_device_ptr(self):
        p = C.c_void_p()
        _check(lib().rt_accum_device_ptr(self._h, C.byref(p)))
        return p.value

    def sync(self):
        _check(lib().rt_sync(self._h))

    def stats(self):
        s = Stats()
        _check(lib().rt_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        _check(lib().rt_reset_stats(self._h))

    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t as an integer (torch.cuda.Stream().cuda_stream), or None for the context's own."""
        _check(lib().rt_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_wavefront_budget(self, paths_in_flight, result_bytes):
        _check(lib().rt_set_wavefront_budget(self._h, paths_in_flight, result_bytes))

    def wavefront_compaction(self, enable=True):
        _check(lib().rt_debug_wavefront_compaction(self._h, 1 if enable else 0))

    def wavefront_persistent(self, enable=True):
        _check(lib().rt_debug_wavefront_persistent(self._h, 1 if enable else 0))

    def wavefront_bands(self, mode=2):
        _check(lib().rt_debug_wavefront_bands(self._h, mode))

    def wavefront_record_cache(self, enable=True):
        _check(lib().rt_debug_wavefront_record_cache(self._h, 1 if enable else 0))

    def wavefront_tuning(self, suspend_lanes=0xFFFFFFFF, list_percent=0xFFFFFFFF, share=0xFFFFFFFF, finish_paths=0xFFFFFFFF):
        _check(lib().rt_debug_wavefront_tuning(self._h, suspend_lanes, list_percent, share, finish_paths))

    def wavefront_counts(self):
        """(rounds, rounds whose trace pass walked the live-slot list, rays suspended, paths handed to the finishing
        kernel) since reset_stats."""
        a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().rt_debug_wavefront_counts(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def fail_wavefront_allocs(self, count):
        """Test hook: kernel 9's next `count` device allocations fail as if the device were out of memory."""
        _check(lib().rt_debug_fail_wavefront_allocs(self._h, count))

    def set_kernel_variant(self, v):
        _check(lib().rt_set_kernel_variant(self._h, v))


def mgpu_partition(height, ranks, rank):
    """rt_mgpu_partition: the rank's share of an image cut into 8-row groups (no GPU needed)."""
    sh = MgpuShare()
    _check(lib().rt_mgpu_partition(height, ranks, rank, C.byref(sh)))
    return sh


class MultiGpu:
    """rt_mgpu: one rt_ctx and one host thread per rank, interleaved 8-row groups, one image exchange per launch."""

    def __init__(self, devices, exchange_mode=EXCHANGE_AUTO):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib().rt_mgpu_create(arr, len(devices), exchange_mode, C.byref(self._h)))
        self.ranks = len(devices)

    def close(self):
        if self._h:
            lib().rt_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def uses_rccl(self):
        return bool(lib().rt_mgpu_uses_rccl(self._h))

    def upload_scene(self, scene):
        d = scene.desc()
        _check(lib().rt_mgpu_upload_scene(self._h, C.byref(d)))

    def update_instances(self, instances, tlas_nodes, tlas_node_count):
        _check(lib().rt_mgpu_update_instances(self._h, T.ptr(instances), len(instances), T.ptr(tlas_nodes), int(tlas_node_count)))

    def set_camera(self, cam):
        _check(lib().rt_mgpu_set_camera(self._h, C.byref(cam)))

    def set_kernel_variant(self, v):
        _check(lib().rt_mgpu_set_kernel_variant(self._h, v))

    def resize(self, w, h):
        _check(lib().rt_mgpu_resize(self._h, w, h))

    def clear_accum(self):
        _check(lib().rt_mgpu_clear_accum(self._h))

    def render_frames(self, params, frame_count=1):
        _check(lib().rt_mgpu_render_frames(self._h, C.byref(params), frame_count))

    def read_accum(self, rank, width, height):
        out = np.zeros((height, width, 4), np.float32)
        _check(lib().rt_mgpu_read_accum(self._h, rank, T.ptr(out)))
        return out

    def stats(self, rank):
        s = Stats()
        _check(lib().rt_mgpu_get_stats(self._h, rank, C.byref(s)))
        return s
