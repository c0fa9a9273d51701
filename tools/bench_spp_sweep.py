#!/usr/bin/env python3
"""Kernel 8 against kernel 9 over samples per launch (keyed PCG, 1920x1080, depth 8): where the automatic choice's
eligibility threshold (paths per launch) belongs.  RT_SCENE=dragon|cornell|teapot, RT_SPPS=1,4,9,...; RT_RNG=pcg|wang
(wang: RT_SPPS counts FRAMES of one sample per batch)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import _ffi, _types as T, renderer as R
W, H = 1920, 1080
r = R.Renderer(W, H, device=0)
scene = os.environ.get("RT_SCENE", "dragon")
if scene == "dragon":
    cam, n = bench.build_dragon_scene(R, r)
else:
    from tests import _scenes as S
    cam, _ = r.load_scene_file(os.path.join(S.ASSETS, "scenes", scene + ".json"))
wang = os.environ.get("RT_RNG", "pcg") == "wang"
r.set_option(R.OPT_MAX_DEPTH, 8); r.set_option(R.OPT_RNG_MODE, T.RNG_WANG if wang else T.RNG_PCG)
ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
for spp in [int(x) for x in os.environ.get("RT_SPPS", "1,4,9,16,36,64").split(",")]:
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1 if wang else spp)
    row = []
    for variant in (8, 9):
        r.set_option(R.OPT_KERNEL_VARIANT, variant)
        best = None
        for rep in range(4):
            _ffi._check(L.rt_reset_stats(ctx)); t0 = time.perf_counter()
            r.render_frames(cam, spp) if wang else r.render(cam)
            r.wait(); dt = time.perf_counter() - t0
            st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
            if rep and (best is None or dt < best[0]): best = (dt, st.rays_traced)
        row.append(f"kernel {variant}: {best[0]*1e3:8.3f} ms {best[1]/best[0]/1e6:7.0f} Mrays/s")
    print(f"{scene} {'frames' if wang else 'spp'} {spp:4d} ({W*H*spp/1e6:.1f} M paths): " + " | ".join(row), flush=True)
