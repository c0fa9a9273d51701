#!/usr/bin/env python3
"""Low-spp launches of the persistent megakernel (kernel 8): effect of the regeneration threshold (kernel 0 beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import bench
from raytracing_amd import _ffi, renderer as R
from tests.test_host_parity import _cornell_via_product

def run(r, cam, variant, spp, depth, frames):
    r.set_option(R.OPT_KERNEL_VARIANT, variant)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    for _ in range(2):
        r.render(cam)
    r.wait()
    ctx = C.c_void_p(r.hip_context_handle()); L = _ffi.lib()
    _ffi._check(L.rt_reset_stats(ctx))
    for _ in range(frames):
        r.render(cam)
    r.wait()
    st = _ffi.Stats(); _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
    return st.total_kernel_ms / frames, st.rays_traced / st.total_kernel_ms / 1e3

W, H = 1280, 720
from tests import _scenes as S
rd = R.Renderer(W, H, device=0)
camd, _ = bench.build_dragon_scene(R, rd)
rc = _cornell_via_product(device=0, w=W, h=H)
camc = R.Camera((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0)
rt = R.Renderer(W, H, device=0)
(mesh, kd, has), = R.Mesh.load_obj(os.path.join(S.ASSETS, "teapot.obj"))
rt.add_blas_instance(rt.add_blas(mesh), R.trs(), rt.default_material)
camt = R.Camera((0.0, 0.6, 2.2), (0, 0.15, 0), (0, 1, 0), 35.0)
for name, r, cam in (("cornell", rc, camc), ("teapot", rt, camt), ("dragon", rd, camd)):
    for spp, depth in ((1, 3), (4, 3), (16, 3), (1, 8), (4, 8), (16, 8), (64, 8)):
        row = []
        ms0, _ = run(r, cam, 0, spp, depth, max(2, 32 // spp))
        for idle in (16, 32, 48, 56):
            ms, mr = run(r, cam, 8 | (idle << 8), spp, depth, max(2, 32 // spp))
            row.append(f"idle{idle}: {ms:7.3f}")
        print(f"{name:8s} spp {spp:2d} depth {depth}  v0 {ms0:7.3f} | v8 " + " ".join(row))
