#!/usr/bin/env python3
"""The per-launch intercept of kernel 9 (persistent mode) on C3: small launches (RT_SPPS, default 1,4,16) against the
suspension threshold (RT_SUSP lanes, default 32,0,8,64).  experiments/r03_persistent/README.md, "Windows that shrink"."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_MAX_DEPTH, int(os.environ.get("RT_DEPTH", "8"))); r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
r.set_option(R.OPT_KERNEL_VARIANT, 9)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for susp in [int(x) for x in os.environ.get("RT_SUSP", "32,0,8,64").split(",")]:
    _ffi._check(L.rt_debug_wavefront_tuning(ctx, susp, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF))
    row = []
    for spp in [int(x) for x in os.environ.get("RT_SPPS", "1,4,16").split(",")]:
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
        best = None
        for rep in range(5):
            t0 = time.perf_counter(); r.render(cam); r.wait(); dt = time.perf_counter() - t0
            if rep and (best is None or dt < best): best = dt
        row.append(f"{spp} spp {best*1e3:.3f} ms")
    print(f"suspend at {susp} lanes:", " | ".join(row), flush=True)
