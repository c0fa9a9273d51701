"""rt_build_tlas_cluster_noalloc (include/rt_hip.h): the clustering kernel with its device buffers in a static
device array of the library instead of per-call hipMalloc.  Through the C ABI, against rt_build_tlas_cluster (the
existing entry point) bit for bit, on two calls in a row — the second with a smaller input, so that whatever the
first call left in the shared array would show — and against the oracle's TLAS builder."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from . import _scenes as S
from .test_tlas_build import _leaves, _same_struct

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_equals_the_allocating_one_twice_in_a_row():
    ctx = _ffi.HipContext(0)
    try:
        for kind, n, seed in (("random", 3000, 21), ("grid", 343, 22), ("flat", 200, 23), ("random", 1, 24),
                              ("random", 4096, 25), ("random", 2, 26)):
            lv = _leaves(n, seed, kind)
            want, wcnt, _ms = ctx.build_tlas_cluster(lv)
            got, gcnt = ctx.build_tlas_cluster_noalloc(lv)
            assert gcnt == wcnt == 2 * n - 1, (kind, n)
            _same_struct(got[:gcnt], want[:wcnt])
    finally:
        ctx.close()


def test_noalloc_entry_point_equals_the_oracle_and_rejects_oversize():
    ctx = _ffi.HipContext(0)
    try:
        for n_side in (20, 6):  # 401 instances, then 37: larger first
            o = S.mixed_scene(n_side=n_side, seed=9)
            so = o.build()
            n = len(o.instances)
            got, cnt = ctx.build_tlas_cluster_noalloc(so.tlas_nodes[1:1 + n])
            assert cnt == so.tlas_node_count == 2 * n - 1
            _same_struct(got[:cnt], so.tlas_nodes[:cnt])
        with pytest.raises(_ffi.RtError, match="RT_TLAS_GPU_MAX_LEAVES"):
            ctx.build_tlas_cluster_noalloc(_leaves(4097, 0))
        bad = _leaves(8, 0)
        bad["aabb_max"][:] = np.inf
        with pytest.raises(_ffi.RtError):
            ctx.build_tlas_cluster_noalloc(bad)
        out = np.zeros(16, T.tlas_node_dtype)
        cnt = C.c_uint32(0)
        assert _ffi.lib().rt_build_tlas_cluster_noalloc(ctx._h, None, 8, T.ptr(out), C.byref(cnt)) != 0
        assert _ffi.lib().rt_build_tlas_cluster_noalloc(ctx._h, T.ptr(bad), 0, T.ptr(out), C.byref(cnt)) != 0
        # two contexts on one GPU share the array: used one after the other they both work
        other = _ffi.HipContext(0)
        try:
            lv = _leaves(500, 31)
            a, ca = other.build_tlas_cluster_noalloc(lv)
            b, cb = ctx.build_tlas_cluster_noalloc(lv)
            assert ca == cb
            _same_struct(a[:ca], b[:cb])
        finally:
            other.close()
    finally:
        ctx.close()
