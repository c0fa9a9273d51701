#!/usr/bin/env python3
"""The reference's own live configuration (PathTracer.cpp:68-69, 390-395; Renderer.cpp:21;
RayTracing.slang:45): built-in Cornell scene, 1280x720, 1 sample per pixel per frame, depth 3,
progressive accumulation.  Reports frames/s and Mrays/s of back-to-back frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from raytracing_amd import _ffi, renderer as R
from tests.test_host_parity import _cornell_via_product

W, H, FRAMES = 1280, 720, 300
r = _cornell_via_product(device=0, w=W, h=H)
cam = R.Camera((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0)
for variant in (_ffi.KERNEL_VARIANT_AUTO, 8, 0, 9):
    r.set_option(R.OPT_KERNEL_VARIANT, variant)
    for _ in range(20):
        r.render(cam)
    r.wait()
    ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
    _ffi._check(L.rt_reset_stats(ctx))
    t0 = time.perf_counter()
    for _ in range(FRAMES):
        r.render(cam)
    r.wait()
    dt = time.perf_counter() - t0
    st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
    print(f"variant {'auto' if variant == _ffi.KERNEL_VARIANT_AUTO else variant}: {FRAMES/dt:8.1f} frames/s wall, kernel {st.total_kernel_ms/FRAMES*1e3:7.1f} us/frame, "
          f"{st.rays_traced/dt/1e6:8.1f} Mrays/s wall, {st.rays_traced/st.total_kernel_ms/1e3:8.1f} Mrays/s in-kernel")
