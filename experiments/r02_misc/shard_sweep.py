import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
r = R.Renderer(1920, 1080, device=0)
cam, n = bench.build_dragon_scene(R, r)
r.set_option(R.OPT_SAMPLES_PER_PIXEL, 256); r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
r.set_option(R.OPT_ROW_BEGIN, 0); r.set_option(R.OPT_ROW_END, 1080); r.set_option(R.OPT_ROW_GROUP_STRIDE, 8)
for regen in (12, 16, 20, 24, 28, 32):
    for slots in (1 << 22, 1 << 23, 1 << 24):
        r.set_option(R.OPT_KERNEL_VARIANT, 9 | (regen << 8))
        _ffi._check(L.rt_set_wavefront_budget(ctx, slots, 10 << 30))
        best = None
        for rep in range(3):
            _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter(); r.render(cam); r.wait(); dt = time.perf_counter() - t0
            if rep and (best is None or dt < best): best = dt
        print(f"regen {regen} slots {slots>>20} Mi: {best*1e3:.2f} ms", flush=True)
