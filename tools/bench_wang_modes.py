import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_WANG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
r.set_option(R.OPT_KERNEL_VARIANT, 9)
for mode in (0, 1):
    _ffi._check(L.rt_debug_wavefront_persistent(ctx, mode))
    best = None
    for rep in range(3):
        _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter(); r.render_frames(cam, 256); r.wait(); dt = time.perf_counter() - t0
        st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
        if rep and (best is None or dt < best[0]): best = (dt, st.rays_traced)
    print(f"wang 256 frames persistent={mode}: {best[0]*1e3:.2f} ms {best[1]/best[0]/1e6:.0f} Mrays/s", flush=True)
