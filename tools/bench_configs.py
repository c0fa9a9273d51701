#!/usr/bin/env python3
"""The BASELINE configs beside C3 (which bench.py measures), on one GPU, default kernel:
C1 Cornell OBJ scene file, C2 teapot, C4 2025 transformed instances, C5 camera (defocus) on C3.
Environment: RT_RNG=wang|pcg (default wang), RT_SPP (default 64), RT_VARIANT (a kernel variant word; default automatic)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import bench
from raytracing_amd import _ffi, renderer as R
from tests import _scenes as S

VARIANT = int(os.environ.get("RT_VARIANT", "-1"))  # optional: a kernel variant word with schedule knobs (sweeps)
RNG = os.environ.get("RT_RNG", "wang")
SPP = int(os.environ.get("RT_SPP", "64"))


def measure(r, cam, spp, depth, frames=3):
    if VARIANT >= 0:
        r.set_option(R.OPT_KERNEL_VARIANT, VARIANT)
    r.set_option(R.OPT_RNG_MODE, 1 if RNG == "pcg" else 0)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    for _ in range(4):  # warm-up; with RT_RNG=pcg the first three launches are the library's kernel trials
        r.render(cam)
    r.wait()
    ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
    _ffi._check(L.rt_reset_stats(ctx))
    for _ in range(frames):
        r.render(cam)
    r.wait()
    st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
    return st.total_kernel_ms / frames, st.rays_traced / st.total_kernel_ms / 1e3, r.total_triangle_count, st.last_kernel

W, H = 1920, 1080
rows = []
# C1: the shipped scene file (Cornell box OBJ, scale 3), sky lit
r = R.Renderer(W, H, device=0)
cam, info = r.load_scene_file(os.path.join(S.ASSETS, "scenes", "cornell.json"))
rows.append((f"C1 CornellBox-Original.obj via scene file, {SPP} spp, depth 8",) + measure(r, cam, SPP, 8))
r.shutdown()
# C2: teapot
r = R.Renderer(W, H, device=0)
cam, info = r.load_scene_file(os.path.join(S.ASSETS, "scenes", "teapot.json"))
rows.append((f"C2 teapot.obj, {SPP} spp, depth 8",) + measure(r, cam, SPP, 8))
r.shutdown()
# C4: 45 x 45 instances of teapot / sphere / cube, mixed materials (the parity test's scene class)
r = R.Renderer(W, H, device=0)
cam, _ = bench.build_instances_scene(R, r)
rows.append((f"C4 2025 instances (teapot/sphere/cube, 4 material types), {SPP} spp, depth 8",) + measure(r, cam, SPP, 8))
r.shutdown()
# C5 camera on the C3 scene: defocus blur, 3840 wide
r = R.Renderer(3840, 2160, device=0)
cam, _ = bench.build_dragon_scene(R, r)
cam.c.defocus_angle, cam.c.focus_dist = 0.8, 4.2
rows.append((f"C5 dragon stand-in 3840x2160 with defocus blur, {SPP} spp, depth 8 (one GPU)",) + measure(r, cam, SPP, 8, frames=2))
r.shutdown()
# C3b: the non-uniformly tessellated second dragon-class mesh (bench.build_teapot_dragon_scene)
r = R.Renderer(W, H, device=0)
cam, _ = bench.build_teapot_dragon_scene(R, r)
rows.append((f"C3b subdivided + displaced teapot (1,048,576 triangles, non-uniform), {SPP} spp, depth 8",) + measure(r, cam, SPP, 8))
r.shutdown()
# C3 itself, for the same settings (bench.py is the authoritative number)
r = R.Renderer(W, H, device=0)
cam, _ = bench.build_dragon_scene(R, r)
rows.append((f"C3 dragon stand-in (regular tube), {SPP} spp, depth 8",) + measure(r, cam, SPP, 8))
r.shutdown()
for name, ms, mrays, tris, kern in rows:
    print(f"[{RNG}] {name}: {tris} triangles (incl. primitives), kernel {kern}, {ms:.1f} ms per launch, {mrays:.0f} Mrays/s")
