#!/usr/bin/env python3
"""Gather-ceiling probe (rt_debug_probe_walk) on the bench scene at 2 ... 8 resident waves per SIMD."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.render(cam); r.wait()
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for waves in (2, 3, 4, 5, 6, 7, 8):
    best = 0.0
    for _ in range(3):
        ms, steps = C.c_double(), C.c_uint64()
        _ffi._check(L.rt_debug_probe_walk(ctx, 256 * waves, 2000, 4, C.byref(ms), C.byref(steps)))
        best = max(best, steps.value / (ms.value * 1e-3))
    print(f"{waves} waves/SIMD: {best/1e9:.1f} G record visits/s", flush=True)
