#!/usr/bin/env python3
"""Persistent mode of kernel 9 with fewer resident trace waves (the path budget limits the windows): how much the
launch gains from its last waves per SIMD.  C3, 1920x1080, 256 spp, PCG."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, 256); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
r.set_option(R.OPT_KERNEL_VARIANT, 9)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for waves_per_simd in (6, 5, 4, 3, 2):
    _ffi._check(L.rt_set_wavefront_budget(ctx, 256 * 4 * waves_per_simd * 512, 40 << 30))
    best = None
    for rep in range(3):
        _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter(); r.render(cam); r.wait(); dt = time.perf_counter() - t0
        st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
        if rep and (best is None or dt < best[0]): best = (dt, st.rays_traced)
    print(f"{waves_per_simd} waves per SIMD ({256 * 4 * waves_per_simd} windows): {best[0]*1e3:.2f} ms {best[1]/best[0]/1e6:.0f} Mrays/s", flush=True)
