#!/usr/bin/env python3
"""Memory-path ceiling of the traversal: dependent divergent 64-B gathers over
the real node records of the bench scene (rt_debug_probe_walk)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _ffi, renderer as R
r = R.Renderer(64, 64, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1)
r.render(cam); r.wait()   # uploads the scene
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for blocks_per_cu in (2, 4, 8):
    for loads, name in ((4, "4x16B per lane"), (5, "quad-cooperative 64B")):
        ms, steps = C.c_double(), C.c_uint64()
        for _ in range(2):
            _ffi._check(L.rt_debug_probe_walk(ctx, 256 * blocks_per_cu, 2000, loads, C.byref(ms), C.byref(steps)))
        print(f"waves/SIMD {blocks_per_cu} {name}: {steps.value/ms.value/1e6:8.1f} G steps/s, {ms.value:.2f} ms")
