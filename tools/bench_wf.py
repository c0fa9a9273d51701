#!/usr/bin/env python3
"""Kernel 9 (wavefront pipeline) against kernel 8 in PCG mode on the bench scene: Mrays/s over knobs and budgets.
Usage: bench_wf.py [spp] [WxH] [variant:slots:sr_GiB ...]  (defaults 64, 1920x1080); RT_SCENE=dragon|cornell|teapot"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _ffi, _types as T, renderer as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
cases = [tuple(int(x, 0) for x in a.split(":")) for a in sys.argv[3:]] or \
    [(8, 0, 0), (9, 1 << 23, 10), (9, 1 << 20, 10), (9, 1 << 25, 10), (9 | (4 << 8), 1 << 23, 10), (9 | (32 << 8), 1 << 23, 10)]
r = R.Renderer(W, H, device=0)
scene = os.environ.get("RT_SCENE", "dragon")
if scene == "dragon":
    cam, n = bench.build_dragon_scene(R, r)
else:
    from tests import _scenes as S
    cam, _ = r.load_scene_file(os.path.join(S.ASSETS, "scenes", scene + ".json"))
r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
r.set_option(R.OPT_MAX_DEPTH, 8)
r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for variant, slots, sr_gib in cases:
    r.set_option(R.OPT_KERNEL_VARIANT, variant)
    if (variant & 0xFF) == 9:
        _ffi._check(L.rt_set_wavefront_budget(ctx, slots, sr_gib << 30))
    best = None
    for rep in range(3):
        _ffi._check(L.rt_reset_stats(ctx))
        t0 = time.perf_counter()
        r.render(cam); r.wait()
        dt = time.perf_counter() - t0
        st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
        if rep and (best is None or dt < best[0]):
            best = (dt, st.rays_traced)
    print(f"variant {variant & 0xFF} knobs {variant >> 8:#x} slots {slots} sr {sr_gib} GiB: {best[0]*1e3:8.2f} ms wall, "
          f"{best[1]/best[0]/1e6:8.1f} Mrays/s", flush=True)
