#!/usr/bin/env python3
"""Stress of kernel 9's round scheduling on the GPU alone: generated scenes (tests/test_gpu_random_scenes.py) at sizes
the oracle would take minutes for, kernel 9 with random budgets and scheduling settings against kernel 0 (one lane
per pixel, no scheduling at all) — images must be equal bit for bit, ray counts equal.  Round 3: either RNG and batches
of 1 ... 9 progressive frames (rt_render_frames).  Usage: stress_wavefront.py [cases]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from raytracing_amd import _ffi, _types as T
from tests import _oracle as O
from tests.test_gpu_random_scenes import _random_scene

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(500, 500 + cases):
    r, (eye, look, up, fov), rng = _random_scene(seed)
    sc = r.build()
    c = _ffi.HipContext(0)
    try:
        c.upload_scene(sc)
        w, h = int(rng.integers(200, 900)), int(rng.integers(100, 500))
        cam = O.camera_uniforms(eye, look, up, fov, w, h)
        c.set_camera(cam)
        c.resize(w, h)
        sqrt_spp, depth = int(rng.integers(2, 7)), int(rng.choice([2, 3, 5, 8, 17]))
        aabb = int(rng.choice([T.AABB_DIV, T.AABB_DIV, T.AABB_RCP]))
        rng_mode, nf = int(rng.choice([T.RNG_PCG, T.RNG_WANG])), int(rng.choice([1, 1, 2, 5, 9]))
        if nf > 1:
            sqrt_spp = min(sqrt_spp, 3)
        p = O.render_params(w, h, int(rng.integers(0, 50)), sqrt_spp, depth, rng_mode=rng_mode, aabb_mode=aabb)
        c.set_kernel_variant(0)
        c.reset_stats()
        c.render_frames(p, nf)
        want, rays0 = c.read_accum(w, h).copy(), c.stats().rays_traced
        paths = w * h * sqrt_spp * sqrt_spp
        for rep in range(3):
            c.set_kernel_variant(9)
            budget = int(rng.choice([4096, 4096 + 64, (1 << 16) + 192, 1 << 18, 1 << 20, 1 << 26]))  # (+64 / +192: a partly filled last window)
            tuning = (int(rng.choice([0, 4, 32, 64])), int(rng.choice([0, 50, 75, 100])), int(rng.choice([64, 256, 1024])),
                      int(rng.choice([0, 1000, 200000, 1 << 30])))
            if rep == 0:
                tuning = (0xFFFFFFFF,) * 4
            c.set_wavefront_budget(budget, 10 << 30)
            c.wavefront_tuning(*tuning)
            c.wavefront_compaction(rep != 2 or bool(rng.integers(0, 2)))  # ordered windows (default) / paths stay in place
            c.wavefront_record_cache(rep != 1 or bool(rng.integers(0, 2)))  # top records in LDS (default) / all through the L1
            persistent = rep == 0 or bool(rng.integers(0, 2))  # no rounds (default) / trace and shade rounds
            c.wavefront_persistent(persistent)
            c.wavefront_bands(2 if rep == 0 else int(rng.integers(0, 2)))  # persistent mode: by launch size / interleaved / bands
            c.clear_accum()
            c.reset_stats()
            c.render_frames(p, nf)
            got, rays9 = c.read_accum(w, h), c.stats().rays_traced
            diff = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
            counts = c.wavefront_counts()
            ok = diff == 0 and rays9 == rays0
            bad += 0 if ok else 1
            print(f"seed {seed} {w}x{h} rng {rng_mode} frames {nf} spp {sqrt_spp**2} depth {depth} paths {paths} budget {budget} tuning {tuning} {'persistent' if persistent else 'rounds'}: "
                  f"rounds/list/suspended/finished {counts} differing {diff} rays {rays9} vs {rays0} {'ok' if ok else 'MISMATCH'}", flush=True)
    finally:
        c.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
