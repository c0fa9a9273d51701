"""rt_render_frames: `n` iterations of the reference's frame loop (PathTracer.cpp:396-411 -> Renderer::render ->
one dispatch per frame, Renderer.cpp:488-652, re-seeded by frame_index, RayTracing.slang:71-72, folded into the
running mean, :176-179) in one call.  The bar is that of everything else here: the image equals the oracle's after the
same frames rendered one by one, bit for bit — for every kernel, for the reference's sequential Wang RNG (whose
(pixel, frame) pairs the wavefront pipeline runs side by side, the samples of a frame in sequence inside a slot) and
for the keyed PCG RNG, across sample chunks, suspension, the live-slot list and the finishing kernel."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from raytracing_amd import renderer as R
from raytracing_amd import sharding
from . import _oracle as O
from . import _scenes as S

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _oracle_frames(sc, cam, w, h, first, count, sqrt_spp, depth, rng_mode, img=None, **rows):
    img = np.zeros((h, w, 4), np.float32) if img is None else img
    for f in range(first, first + count):
        O.render(sc, cam, O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng_mode, **rows), img, nthreads=16)
    return img


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.HipContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("rng_mode", [T.RNG_WANG, T.RNG_PCG])
@pytest.mark.parametrize("sqrt_spp", [1, 2, 3])
def test_frame_batch_equals_frames_one_by_one(ctx, rng_mode, sqrt_spp):
    """All four materials, instances, textures (the `mixed` scene): 3 frames by rt_render, then 7 by one
    rt_render_frames, on every kernel, against the oracle's ten frames; kernel 9 also with budgets that force chunks
    of frames / samples and many rounds, and with every round-scheduling mechanism forced on."""
    sc = S.mixed_scene().build()
    w, h, depth = 160, 90, 6
    cam = S.mixed_camera(w, h)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    want3 = _oracle_frames(sc, cam, w, h, 0, 3, sqrt_spp, depth, rng_mode)
    want10 = _oracle_frames(sc, cam, w, h, 3, 7, sqrt_spp, depth, rng_mode, img=want3.copy())
    assert not _same(want3, want10)
    # (persistent: kernel 9 without rounds, the default; False = trace / shade rounds, to which the last three tuning values belong)
    settings = [(0, None, None, True), (8, None, None, True), (9, None, None, True), (9, None, None, False),
                (9, (4096, 1 << 20), None, True), (9, (4096, 1 << 20), None, False),
                (9, (1 << 12, 5 << 18), (64, 100, 64, 0), False), (9, (1 << 12, 5 << 18), (64, 100, 64, 0), True),
                (9, (1 << 16, 1 << 24), (64, 100, 64, 1 << 30), False)]
    for variant, budget, tuning, persistent in settings:
        ctx.set_kernel_variant(variant)
        ctx.set_wavefront_budget(*(budget or (1 << 26, 40 << 30)))
        ctx.wavefront_tuning(*(tuning or (0xFFFFFFFF,) * 4))
        ctx.wavefront_persistent(persistent)
        ctx.clear_accum()
        for f in range(3):
            ctx.render(O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng_mode))
        assert _same(ctx.read_accum(w, h), want3), (variant, budget, tuning, persistent)
        ctx.render_frames(O.render_params(w, h, 3, sqrt_spp, depth, rng_mode=rng_mode), 7)
        assert ctx.stats().last_kernel == variant
        assert _same(ctx.read_accum(w, h), want10), (variant, budget, tuning, persistent)
    ctx.wavefront_persistent(True)
    ctx.wavefront_tuning()
    ctx.set_wavefront_budget(1 << 26, 40 << 30)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)


def test_frame_batch_with_row_ranges_defocus_and_flags(ctx):
    """The launch parameters a batch shares with a single launch: interleaved row groups (the multi-GPU partition),
    the thin-lens camera (extra RNG draws per camera ray — in the Wang stream they shift every later draw), the
    dielectric flag."""
    sc = S.mixed_scene().build()
    w, h, depth = 128, 96, 5
    cam = S.mixed_camera(w, h)
    cam.defocus_disk_u[:] = [0.02, 0.0, 0.0, 1.0]
    cam.defocus_disk_v[:] = [0.0, 0.02, 0.0, 1.0]
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    rows = sharding.row_interleave(1, 3, h)
    for rng_mode in (T.RNG_WANG, T.RNG_PCG):
        want = _oracle_frames(sc, cam, w, h, 0, 5, 2, depth, rng_mode, flags=T.RENDER_DIELECTRIC_ALWAYS_DRAWS, **rows)
        for variant in (8, 9):
            ctx.set_kernel_variant(variant)
            ctx.clear_accum()
            ctx.render_frames(O.render_params(w, h, 0, 2, depth, rng_mode=rng_mode, flags=T.RENDER_DIELECTRIC_ALWAYS_DRAWS, **rows), 5)
            assert _same(ctx.read_accum(w, h), want), (rng_mode, variant)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)


def test_frame_batch_argument_checks(ctx):
    sc = S.cornell_builtin().build()
    ctx.upload_scene(sc)
    ctx.set_camera(S.cornell_camera(64, 36))
    p = O.render_params(64, 36, 0, 1, 3)
    with pytest.raises(_ffi.RtError):
        ctx.render_frames(p, 0)
    p.frame_index = 0xFFFFFFF0
    with pytest.raises(_ffi.RtError):
        ctx.render_frames(p, 32)


def test_c3_reference_rng_frame_batch_takes_the_wavefront_pipeline():
    """The reference's own configuration of the RNG — Wang hash, ONE sample per pixel per frame (Renderer.cpp:21, 625),
    high sample counts by progressive frames — on the C3 scene at 1920x1080: 24 frames in one call.  The automatic
    choice warms kernel 9 up, times it and the frame-by-frame megakernel once each and keeps the faster; every image equals the
    others and, on three 8-row groups, the oracle's 24 sequential frames."""
    import bench
    w, h, depth, frames = 1920, 1080, 8, 24
    r = R.Renderer(w, h, device=0)
    cam, _ = bench.build_dragon_scene(R, r)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_RNG_MODE, T.RNG_WANG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1)
    r.set_option(R.OPT_KERNEL_VARIANT, _ffi.KERNEL_VARIANT_AUTO)
    st = _ffi.Stats()
    imgs, kernels = [], []
    for _ in range(4):
        r.reset_accumulation()
        r.render_frames(cam, frames)
        assert r.frame_index == frames
        imgs.append(r.read_image())
        _ffi._check(_ffi.lib().rt_get_stats(C.c_void_p(r.hip_context_handle()), C.byref(st)))
        kernels.append(st.last_kernel)
    assert kernels[:3] == [9, 9, 8] and kernels[3] in (8, 9), kernels
    assert all(_same(imgs[0], im) for im in imgs[1:])
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    want = _oracle_frames(sa, camu, w, h, 0, frames, 1, depth, T.RNG_WANG, row_begin=0, row_end=h, row_group_stride=67)
    rows = sharding.interleaved_rows(0, 67, h)
    assert len(rows) == 24 and _same(imgs[0][rows], want[rows])
    # a single frame of the reference's RNG (2.07e6 independent units: just eligible since the end of round 3) is
    # another launch shape: measured again, starting with kernel 9
    r.reset_accumulation()
    r.render(cam)
    _ffi._check(_ffi.lib().rt_get_stats(C.c_void_p(r.hip_context_handle()), C.byref(st)))
    assert st.last_kernel == 9
    r.shutdown()
