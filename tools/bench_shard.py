#!/usr/bin/env python3
"""What one rank of an N-GPU run renders per step (interleaved 8-row groups, row_group_stride = N) on ONE GPU:
kernel 8 against kernel 9 on the bench scene, 1920x1080, 256 spp (RT_SPP), keyed PCG.  RT_STRIDES, RT_KERNELS select, RT_BANDS = rt_debug_wavefront_bands' mode;
RT_WF_TUNING = suspend_lanes,list_percent,share,finish_paths for kernel 9 (-1 = default)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, int(os.environ.get('RT_SPP', '256'))); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
_ffi._check(L.rt_debug_wavefront_bands(ctx, int(os.environ.get('RT_BANDS', '2'))))  # 0 interleaved, 1 bands, 2 by launch size
if os.environ.get("RT_WF_TUNING"):  # suspend_lanes,list_percent,share,finish_paths (-1 = default)
    _ffi._check(L.rt_debug_wavefront_tuning(ctx, *[int(x) & 0xFFFFFFFF for x in os.environ["RT_WF_TUNING"].split(",")]))
for stride in [int(x) for x in os.environ.get('RT_STRIDES', '8,4,2,1').split(',')]:
    r.set_option(R.OPT_ROW_BEGIN, 0); r.set_option(R.OPT_ROW_END, 1080); r.set_option(R.OPT_ROW_GROUP_STRIDE, stride)
    for variant in [int(x) for x in os.environ.get('RT_KERNELS', '8,9').split(',')]:
        r.set_option(R.OPT_KERNEL_VARIANT, variant)
        best = None
        for rep in range(3):
            _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter(); r.render(cam); r.wait(); dt = time.perf_counter() - t0
            st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
            if rep and (best is None or dt < best[0]): best = (dt, st.rays_traced)
        print(f"stride {stride} kernel {variant}: {best[0]*1e3:.2f} ms {best[1]/best[0]/1e6:.0f} Mrays/s", flush=True)
