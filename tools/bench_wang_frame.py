#!/usr/bin/env python3
"""One frame of C3 at RT_SPP (default 256) samples per pixel with the reference's RNG (a pixel's samples in sequence):
kernel 8 against kernel 9 (its kind-2 persistent kernel)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
spp = int(os.environ.get("RT_SPP", "256"))
r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_WANG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for variant in (8, 9):
    r.set_option(R.OPT_KERNEL_VARIANT, variant)
    best = None
    for rep in range(3):
        _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter(); r.render(cam); r.wait(); dt = time.perf_counter() - t0
        st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
        if rep and (best is None or dt < best[0]): best = (dt, st.rays_traced)
    print(f"wang {spp} spp one frame, kernel {variant}: {best[0]*1e3:.2f} ms {best[1]/best[0]/1e6:.0f} Mrays/s", flush=True)
