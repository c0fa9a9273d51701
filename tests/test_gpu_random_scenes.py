"""Randomised differential test of the render path: 72 generated scenes (random instance counts, non-uniform and
mirrored scales, nested overlaps, all material types, textures, axis-parallel cameras, camera inside geometry),
random launch parameters (samples, depth, RNG mode, slab arithmetic, row ranges, progressive frames) and random
kernel variants — the HIP image must equal the oracle's bit for bit, and so must closest hits of rays shot from
inside the scene.  Seeds are fixed: the cases are the same on every run."""
import math

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from . import _oracle as O
from . import _scenes as S

pytestmark = pytest.mark.gpu
VARIANTS = [0, 8, 8 | (1 << 28), 8 | (1 << 31), _ffi.KERNEL_VARIANT_AUTO]


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.HipContext(0)
    yield c
    c.close()


def _random_scene(seed):
    rng = np.random.default_rng(1000 + seed)
    r = O.OracleRenderer()
    blases = [r.add_blas(*S.plane_mesh()), r.add_blas(*S.cube_mesh()),
              r.add_blas(*S.sphere_mesh(0.5, int(rng.integers(4, 14)), int(rng.integers(3, 9))))]
    if seed % 3 == 0:  # a mesh whose triangles are random: long slivers, degenerate and duplicate triangles
        n = int(rng.integers(4, 40))
        pos = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        pos[0] = pos[1]  # duplicate vertex -> zero-area triangles
        idx = rng.integers(0, n, (int(rng.integers(3, 60)), 3)).astype(np.uint32)
        nrm = rng.normal(size=(n, 3)).astype(np.float32)
        blases.append(r.add_blas(pos, nrm, rng.uniform(0, 1, (n, 2)).astype(np.float32), idx.reshape(-1)))
    tex = rng.integers(0, 256, (int(rng.integers(1, 6)), int(rng.integers(1, 6)), 4), dtype=np.uint8)
    mats = [r.default_material, r.add_lambert_material(rng.uniform(0, 1, 3)),
            r.add_metal_material(rng.uniform(0.3, 1, 3), float(rng.uniform(0, 0.6))),
            r.add_metal_material((0.9, 0.9, 0.9), 0.0), r.add_dielectric_material(float(rng.uniform(1.1, 2.4))),
            r.add_dielectric_material(1.0), r.add_emissive_material(rng.uniform(0, 6, 3)),
            r.add_lambert_texture(tex.shape[1], tex.shape[0], tex)]
    for k in range(int(rng.integers(1, 28))):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        q = O.angle_axis(float(rng.uniform(0, 2 * math.pi)), axis) if rng.random() < 0.8 else (1, 0, 0, 0)
        scale = rng.uniform(0.2, 3.0, 3) * rng.choice([1, 1, 1, -1], 3)  # non-uniform, sometimes mirrored
        if rng.random() < 0.15:
            scale[int(rng.integers(0, 3))] *= 1e-3  # nearly flat
        pos = rng.uniform(-3, 3, 3) if rng.random() < 0.85 else np.zeros(3)
        r.add_blas_instance(int(rng.choice(blases)), O.trs(pos, q, scale), mats[int(rng.integers(0, len(mats)))])
    eye = rng.uniform(-5, 5, 3)
    if seed % 5 == 0:
        eye = np.array([0.0, 0.0, 6.0])  # axis-parallel view directions: zero components in d
        look, up = np.array([0.0, 0.0, 0.0]), (0, 1, 0)
    else:
        look, up = rng.uniform(-1, 1, 3), (0, 1, 0)
    return r, (eye, look, up, float(rng.uniform(15, 100))), rng


@pytest.mark.parametrize("seed", range(72))
def test_random_scene_images_and_hits_are_bit_exact(ctx, seed):
    r, (eye, look, up, fov), rng = _random_scene(seed)
    sc = r.build()
    ctx.upload_scene(sc)
    w, h = int(rng.integers(9, 90)), int(rng.integers(9, 60))
    cam = O.camera_uniforms(eye, look, up, fov, w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    sqrt_spp, depth = int(rng.integers(1, 4)), int(rng.choice([0, 1, 2, 3, 5, 8, 17]))
    rng_mode = int(rng.choice([T.RNG_WANG, T.RNG_WANG, T.RNG_PCG]))
    aabb = int(rng.choice([T.AABB_DIV, T.AABB_DIV, T.AABB_RCP]))
    variant = VARIANTS[seed % len(VARIANTS)]
    ctx.set_kernel_variant(variant)
    rows = sorted(int(v) for v in rng.integers(0, h + 1, 2))
    if rows[0] == rows[1] or seed % 2:
        rows = [0, h]
    stride = int(rng.choice([0, 0, 2, 3]))
    want = np.zeros((h, w, 4), np.float32)
    try:
        for frame in range(int(rng.integers(1, 4))):
            p = O.render_params(w, h, frame, sqrt_spp, depth, rng_mode=rng_mode, aabb_mode=aabb,
                                row_begin=rows[0], row_end=rows[1], row_group_stride=stride)
            ctx.render(p)
            O.render(sc, cam, p, want, nthreads=8)
        got = ctx.read_accum(w, h)
    finally:
        ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, variant, sqrt_spp, depth, rng_mode, aabb)
    # closest hits of rays starting inside the scene (inside boxes, on surfaces, axis-parallel directions)
    n = 4000
    rays = np.zeros(n, T.ray_dtype)
    rays["origin"] = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 10, int(rng.integers(0, 3))] = 0.0  # a zero component: divisions by zero in the slab test
    d[n // 10: n // 8] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 8 - n // 10)]
    rays["direction"] = d
    rays["t_min"], rays["t_max"] = 0.0, np.float32(rng.choice([1e30, 5.0, 0.5]))
    for mode in (T.AABB_DIV, T.AABB_RCP):
        a, b = ctx.trace_closest(rays, mode), O.trace_closest(sc, rays, mode)
        for f in ("t", "u", "v"):
            assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), (seed, mode, f)
        assert np.array_equal(a["tri_id"], b["tri_id"]) and np.array_equal(a["inst_id"], b["inst_id"]), (seed, mode)


@pytest.mark.parametrize("seed", range(100, 136))
def test_random_scene_kernel9_round_scheduling_is_invisible(seed):
    """The same generated scenes through kernel 9 (keyed PCG RNG) with random path-slot budgets and random settings of
    its round scheduling (suspended rays, live-slot list, share size, finishing kernel): the image equals the
    oracle's bit for bit and the rays counter equals the oracle's count, whatever the schedule."""
    r, (eye, look, up, fov), rng = _random_scene(seed)
    sc = r.build()
    c = _ffi.HipContext(0)
    try:
        c.upload_scene(sc)
        w, h = int(rng.integers(40, 160)), int(rng.integers(24, 100))
        cam = O.camera_uniforms(eye, look, up, fov, w, h)
        c.set_camera(cam)
        c.resize(w, h)
        sqrt_spp, depth = int(rng.integers(1, 4)), int(rng.choice([1, 2, 3, 5, 8, 17]))
        aabb = int(rng.choice([T.AABB_DIV, T.AABB_DIV, T.AABB_RCP]))
        c.set_kernel_variant(9)
        c.set_wavefront_budget(int(rng.choice([256, 1024, 4096, 1 << 14, 1 << 20])), int(rng.choice([1 << 20, 1 << 30])))
        tuning = (int(rng.choice([0, 4, 32, 64])), int(rng.choice([0, 50, 100])), int(rng.choice([64, 128, 256, 1024])),
                  int(rng.choice([0, 100, 5000, 1 << 30])))
        c.wavefront_tuning(*tuning)
        persistent = bool(rng.integers(0, 2))  # no rounds (the default; budgets below one window of 512 slots fall back) / rounds
        c.wavefront_persistent(persistent)
        bands = int(rng.integers(0, 2))  # the persistent mode's paths interleaved over the XCDs / a band of the frame per XCD
        c.wavefront_bands(bands)
        c.reset_stats()
        cn = O.Counters()
        want = np.zeros((h, w, 4), np.float32)
        for frame in range(int(rng.integers(1, 3))):
            p = O.render_params(w, h, frame, sqrt_spp, depth, rng_mode=T.RNG_PCG, aabb_mode=aabb)
            c.render(p)
            O.render(sc, cam, p, want, counters=cn, nthreads=8)
        got = c.read_accum(w, h)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (seed, tuning, persistent, bands, sqrt_spp, depth, aabb)
        assert c.stats().rays_traced == cn.rays, (seed, tuning, persistent)
    finally:
        c.close()
