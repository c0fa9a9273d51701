#!/usr/bin/env python3
"""Where the time of one rank's launch goes (kernel 9, interleaved 8-row groups with stride RT_STRIDES on ONE GPU):
per-kernel GPU time from events around every dispatch, rounds, wall time.  RT_SPP (256), RT_WF_TUNING as bench_shard.py."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, int(os.environ.get('RT_SPP', '256'))); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
if os.environ.get("RT_WF_TUNING"):
    _ffi._check(L.rt_debug_wavefront_tuning(ctx, *[int(x) & 0xFFFFFFFF for x in os.environ["RT_WF_TUNING"].split(",")]))
r.set_option(R.OPT_KERNEL_VARIANT, 9)
for stride in [int(x) for x in os.environ.get('RT_STRIDES', '8,1').split(',')]:
    r.set_option(R.OPT_ROW_BEGIN, 0); r.set_option(R.OPT_ROW_END, 1080); r.set_option(R.OPT_ROW_GROUP_STRIDE, stride)
    r.render(cam); r.wait()
    t0 = time.perf_counter(); r.render(cam); r.wait(); wall = time.perf_counter() - t0
    _ffi._check(L.rt_debug_wavefront_timing(ctx, 1, None, None))
    _ffi._check(L.rt_reset_stats(ctx))
    r.render(cam); r.wait()
    ms5, nl = (C.c_double * 5)(), C.c_uint64()
    _ffi._check(L.rt_debug_wavefront_timing(ctx, 0, ms5, C.byref(nl)))
    rounds = (C.c_uint64 * 4)()
    _ffi._check(L.rt_debug_wavefront_counts(ctx, *[C.byref(C.c_uint64.from_buffer(rounds, 8 * k)) for k in range(4)]))
    print(f"stride {stride}: wall {wall*1e3:.2f} ms; trace {ms5[0]:.2f} shade {ms5[1]:.2f} finish {ms5[2]:.2f} resolve {ms5[3]:.2f} fill {ms5[4]:.2f} "
          f"sum {sum(ms5):.2f}; rounds {rounds[0]} list rounds {rounds[1]} suspended {rounds[2]} finished paths {rounds[3]}", flush=True)
