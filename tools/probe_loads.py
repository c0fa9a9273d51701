#!/usr/bin/env python3
"""Gather probe (rt_debug_probe_walk) on the bench scene: step rate by loads per step (2 ... 4 dwordx4 per lane and step,
5 = quad-cooperative: one 64-byte access per record) and resident waves per SIMD — is the per-lane gather bound by the
number of L1 accesses (rate ~ 1 / loads) or by latency (rate ~ waves)?"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.render(cam); r.wait()
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for loads in (2, 3, 4, 5):
    row = []
    for waves in (2, 4, 6, 8):
        best = 0.0
        for _ in range(3):
            ms, steps = C.c_double(), C.c_uint64()
            _ffi._check(L.rt_debug_probe_walk(ctx, 256 * waves, 2000, loads, C.byref(ms), C.byref(steps)))
            best = max(best, steps.value / (ms.value * 1e-3))
        row.append(f"{waves} waves {best/1e9:6.1f}")
    print(f"loads {loads}: " + "  ".join(row) + "  G steps/s", flush=True)
