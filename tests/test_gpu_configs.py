"""Every BASELINE.json config as a GPU test on the kernel that ships (RT_KERNEL_VARIANT_AUTO, which must resolve
to kernel 8 for these launches), against the oracle:

  C1  Scenes/CornellBox through assets/scenes/cornell.json, 256x144, 16 spp, depth 8 — whole image vs the oracle
  C2  assets/scenes/teapot.json at its stated 1920x1080, 64 spp, depth 8 — 40 rows vs the oracle + properties
  C3  dragon-class stand-in (870,402 triangles), 1920x1080, depth 8 — properties + 40 rows vs the oracle
  C4  many instances: tests/test_gpu_parity.py::test_many_instances_c4_scale (2025 instances, whole image)
  C5  the C3 scene at 3840x2160 with defocus blur and the keyed PCG RNG — properties, three interleaved shares
      stitched (the multi-GPU partition), 40 rows vs the oracle

Where the oracle is too slow for the whole image inside the suite, size-independent properties carry the rest:
the same launch twice gives the same bits, the reference-structured kernel 0 and the default kernel agree on
every pixel, interleaved shares compose to the whole launch, alpha is 1 and everything is finite."""
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

SCENES = os.path.join(S.ASSETS, "scenes")


def _last_kernel(r):
    st = _ffi.Stats()
    _ffi._check(_ffi.lib().rt_get_stats(C.c_void_p(r.hip_context_handle()), C.byref(st)))
    return st.last_kernel


def _launch(r, cam, variant, rows=None):
    r.set_option(R.OPT_KERNEL_VARIANT, variant)
    for k, v in (rows or {"row_begin": 0, "row_end": 0, "row_group_stride": 0}).items():
        r.set_option({"row_begin": R.OPT_ROW_BEGIN, "row_end": R.OPT_ROW_END,
                      "row_group_stride": R.OPT_ROW_GROUP_STRIDE}[k], v)
    r.reset_accumulation()
    r.render(cam)
    return r.read_image()


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _properties_and_oracle_rows(r, cam, w, h, sqrt_spp, depth, rng_mode, shards=3, oracle_stride=33):
    """The shared body of the full-size tests.  Returns the whole image of the default kernel."""
    a = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
    # the shipping kernel is what ran (a launch of >= 2e6 independent paths starts with the trial of kernel 9 — with the
    # reference's RNG too since the end of round 3: a single frame's pixels are 2e6 independent units at 1080p)
    assert _last_kernel(r) in (8, 9)
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and (a[..., :3] >= 0).all()
    assert _same(a, _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO))    # deterministic
    b = _launch(r, cam, 0)
    assert _last_kernel(r) == 0
    assert _same(a, b)                                            # reference-structured kernel == default kernel
    stitched = np.zeros_like(a)
    for rank in range(shards):                                    # the multi-GPU partition, on one GPU
        img = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO, sharding.row_interleave(rank, shards, h))
        mine = sharding.interleaved_rows(rank, shards, h)
        stitched[mine] = img[mine]
    assert _same(a, stitched)
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    want = np.zeros_like(a)
    p = O.render_params(w, h, 0, sqrt_spp, depth, rng_mode=rng_mode, row_begin=0, row_end=h,
                        row_group_stride=oracle_stride)
    O.render(sa, camu, p, want, nthreads=16)
    rows = sharding.interleaved_rows(0, oracle_stride, h)
    assert len(rows) >= 40 and _same(a[rows], want[rows])         # >= 40 rows spread over the image, bit for bit
    return a


def test_c1_cornell_scene_file_whole_image():
    """C1 at the size BASELINE.md states: CornellBox-Original.obj through the scene file, 256x144, 16 spp, depth 8,
    the whole image against the oracle, both kernels."""
    w, h, sqrt_spp, depth = 256, 144, 4, 8
    r = R.Renderer(w, h, device=0)
    cam, info = r.load_scene_file(os.path.join(SCENES, "cornell.json"))
    assert info["triangles"] == 36
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, sqrt_spp * sqrt_spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    want = O.render(sa, camu, O.render_params(w, h, 0, sqrt_spp, depth), nthreads=16)
    got = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
    assert _last_kernel(r) == 8
    assert _same(got, want)
    assert _same(_launch(r, cam, 0), want)
    # second progressive frame on top of the first (RayTracing.slang:176-179)
    r.set_option(R.OPT_KERNEL_VARIANT, _ffi.KERNEL_VARIANT_AUTO)
    r.reset_accumulation()
    r.render(cam)
    r.render(cam)
    O.render(sa, camu, O.render_params(w, h, 1, sqrt_spp, depth), want, nthreads=16)
    assert _same(r.read_image(), want)
    r.shutdown()


def test_c2_teapot_scene_file_full_size():
    """C2 at its stated size: teapot.obj through assets/scenes/teapot.json, 1920x1080, 64 spp, depth 8."""
    w, h, sqrt_spp, depth = 1920, 1080, 8, 8
    r = R.Renderer(w, h, device=0)
    cam, info = r.load_scene_file(os.path.join(SCENES, "teapot.json"))
    assert info["screen_width"] == w and info["screen_height"] == h and info["triangles"] == 1024
    assert info["samples_per_pixel"] == 64 and info["max_depth"] == depth
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, sqrt_spp * sqrt_spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    _properties_and_oracle_rows(r, cam, w, h, sqrt_spp, depth, T.RNG_WANG)
    r.shutdown()


def test_c3_dragon_full_size():
    """C3 at its full size (870,402-triangle dragon stand-in, 1920x1080, depth 8; 4 spp per launch so that kernel 0
    and the oracle rows stay within seconds — bench.py runs the same scene at 256 spp per launch and checks a whole
    4-spp frame against the oracle on every default run)."""
    import bench
    w, h, sqrt_spp, depth = 1920, 1080, 2, 8
    r = R.Renderer(w, h, device=0)
    cam, n_tris = bench.build_dragon_scene(R, r)
    assert n_tris == 870402
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, sqrt_spp * sqrt_spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    _properties_and_oracle_rows(r, cam, w, h, sqrt_spp, depth, T.RNG_WANG)
    r.shutdown()


def test_spp_is_truncated_to_a_square_like_the_reference():
    """sqrt_spp = u32(sqrt(spp)) (Renderer.cpp:625) — 10 samples per pixel render 3 x 3."""
    r = R.Renderer(96, 54, device=0)
    cam, _ = r.load_scene_file(os.path.join(SCENES, "teapot.json"))
    r.set_option(R.OPT_MAX_DEPTH, 4)
    imgs = {}
    for spp in (9, 10, 15, 16):
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
        imgs[spp] = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
    assert _same(imgs[9], imgs[10]) and _same(imgs[9], imgs[15]) and not _same(imgs[9], imgs[16])
    r.shutdown()


def test_automatic_choice_times_both_kernels_once_per_scene_and_shape():
    """Keyed PCG, 1920x1080, 64 spp: a context that was given no kernel runs kernel 9 on the first two launches of a
    shape (warm-up, timed), kernel 8 on the third and the faster of the two afterwards; a new scene or shape starts over;
    all the same bits.
    On the closed Cornell box the megakernel and kernel 9 are within a few per cent of each other at this size (with
    kernel 9's rounds as without them: either may win, the choice must only be kept); under the open sky of the teapot
    scene the pipeline wins by a factor of two."""
    w, h = 1920, 1080
    for scene, persistent, winner in (("cornell.json", False, None), ("cornell.json", True, None), ("teapot.json", True, 9)):
        r = R.Renderer(w, h, device=0)
        _ffi._check(_ffi.lib().rt_debug_wavefront_persistent(C.c_void_p(r.hip_context_handle()), 1 if persistent else 0))
        cam, _ = r.load_scene_file(os.path.join(SCENES, scene))
        r.set_option(R.OPT_MAX_DEPTH, 8)
        r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, 64)
        imgs, kernels = [], []
        for _ in range(5):
            imgs.append(_launch(r, cam, _ffi.KERNEL_VARIANT_AUTO))
            kernels.append(_last_kernel(r))
        assert kernels[:3] == [9, 9, 8] and kernels[3] == kernels[4] and kernels[3] in ((8, 9) if winner is None else (winner,)), (scene, kernels)
        assert all(_same(imgs[0], im) for im in imgs[1:])
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, 100)           # another shape: measured again
        _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
        assert _last_kernel(r) == 9
        r.shutdown()


def test_kernel9_launch_returns_before_the_image_is_done():
    """The reference's dispatch is asynchronous (Renderer.cpp:644-651).  Kernel 9 in its default, persistent mode reads
    nothing back, so rt_render returns at once as it does for kernels 0 and 8 (it blocked for the length
    of the launch while the host read path counters between rounds — which the rounds mode still does)."""
    import bench, time
    w, h = 1920, 1080
    r = R.Renderer(w, h, device=0)
    cam, _ = bench.build_dragon_scene(R, r)
    r.set_option(R.OPT_MAX_DEPTH, 8)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 64)
    r.set_option(R.OPT_KERNEL_VARIANT, 9)
    ctx = C.c_void_p(r.hip_context_handle())
    times = {}
    for persistent in (1, 0):
        _ffi._check(_ffi.lib().rt_debug_wavefront_persistent(ctx, persistent))
        r.render(cam)  # buffers of this mode allocated, code loaded
        r.wait()
        t0 = time.perf_counter()
        r.render(cam)
        t_call = time.perf_counter() - t0
        r.wait()
        times[persistent] = (t_call, time.perf_counter() - t0)
    call, whole = times[1]
    assert whole > 0.03 and call < 0.25 * whole, times  # a 64-spp launch takes about 58 ms; the call a fraction of one
    call0, whole0 = times[0]
    assert call0 > 0.8 * whole0, times                  # the rounds: the call IS the launch
    r.shutdown()


def test_c3_pcg_takes_the_wavefront_pipeline():
    """C3 with the keyed PCG RNG at 64 spp per launch: the automatic choice is kernel 9 (the sample-parallel
    wavefront pipeline, rt_wavefront.hip), and its image equals kernel 8's and — on the same 40 rows as above, at
    4 spp — the oracle's, bit for bit; also with the quad-cooperative record fetch and with a budget that forces
    sample chunks."""
    import bench
    w, h, depth = 1920, 1080, 8
    r = R.Renderer(w, h, device=0)
    cam, _ = bench.build_dragon_scene(R, r)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 64)
    a = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
    assert _last_kernel(r) == 9
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()
    assert _same(a, _launch(r, cam, 8))
    assert _same(a, _launch(r, cam, 9 | (1 << 15)))
    _ffi._check(_ffi.lib().rt_set_wavefront_budget(C.c_void_p(r.hip_context_handle()), 1 << 22, 1 << 30))  # 32 chunks
    assert _same(a, _launch(r, cam, 9))
    _ffi._check(_ffi.lib().rt_set_wavefront_budget(C.c_void_p(r.hip_context_handle()), 1 << 25, 10 << 30))
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 4)
    b = _launch(r, cam, 9)
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    want = np.zeros_like(b)
    O.render(sa, camu, O.render_params(w, h, 0, 2, depth, rng_mode=T.RNG_PCG, row_begin=0, row_end=h, row_group_stride=33),
             want, nthreads=16)
    rows = sharding.interleaved_rows(0, 33, h)
    assert _same(b[rows], want[rows])
    r.shutdown()


def test_c5_dragon_4k_defocus_pcg_sharded():
    """C5: the dragon scene at 3840x2160 with defocus blur (Random.slang:57-70 semantics) and the keyed PCG RNG,
    4 spp per launch; interleaved shares of three ranks stitch to the one-GPU image bit for bit, and 40 rows match
    the oracle."""
    import bench
    w, h, sqrt_spp, depth = 3840, 2160, 2, 8
    r = R.Renderer(w, h, device=0)
    cam, n_tris = bench.build_dragon_scene(R, r, defocus=True)
    assert n_tris == 870402 and cam.c.defocus_angle > 0
    camu = r.camera_uniforms(cam)
    assert camu.defocus_disk_u[3] == 1.0   # the kernels' "defocus on" flag
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, sqrt_spp * sqrt_spp)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    a = _properties_and_oracle_rows(r, cam, w, h, sqrt_spp, depth, T.RNG_PCG, oracle_stride=54)
    # the blur does something: the same launch with a pinhole camera differs
    pin = R.Camera(tuple(cam.c.position), tuple(cam.c.look_at), tuple(cam.c.v_up), cam.c.fov)
    assert not _same(a, _launch(r, pin, _ffi.KERNEL_VARIANT_AUTO))
    r.shutdown()


def test_c3_fixed_ray_sets_a_million_camera_and_a_million_secondary_rays():
    """SURVEY.md 8(d), correctness gate (1): (t bits, tri_id, inst_id) — and u, v — identical CPU <-> HIP on fixed ray
    sets on the C3 scene: the 1.05 M camera rays of a 1366x768 frame and the 1.1 M cosine-scattered rays of their first
    three bounces (frame 7 seeds the RNG), closest hits through the C ABI's rt_trace_closest against the oracle's."""
    import bench
    w, h = 1366, 768
    r = R.Renderer(w, h, device=0)
    cam, n_tris = bench.build_dragon_scene(R, r)
    assert n_tris == 870402
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1)
    r.render(cam)  # uploads the scene
    r.wait()
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    ctx = C.c_void_p(r.hip_context_handle())
    L = _ffi.lib()
    pr = O.render_params(w, h, 7, 1, 8)
    total = 0
    for bounce in (0, 1, 2, 3):
        rays = O.gen_rays(sa, camu, pr, bounce)
        if bounce == 0:
            assert len(rays) == w * h
        total += len(rays)
        want = O.trace_closest(sa, rays, T.AABB_DIV)
        got = np.zeros(len(rays), T.hit_dtype)
        _ffi._check(L.rt_trace_closest(ctx, T.ptr(rays), len(rays), T.ptr(got), T.AABB_DIV))
        for f in ("t", "u", "v"):
            assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), (bounce, f)
        assert np.array_equal(got["tri_id"], want["tri_id"]) and np.array_equal(got["inst_id"], want["inst_id"]), bounce
        assert (want["tri_id"] != T.INVALID_ID).mean() > 0.2, bounce
    assert total - w * h > 1_000_000  # secondary rays
    r.shutdown()


def _oracle_rows(r, cam, w, h, sqrt_spp, depth, rng_mode, stride):
    """Rows of the 8-row groups 0, stride, 2 * stride, ... rendered by the oracle: (row indices, image)."""
    sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
    want = np.zeros((h, w, 4), np.float32)
    O.render(sa, camu, O.render_params(w, h, 0, sqrt_spp, depth, rng_mode=rng_mode, row_begin=0, row_end=h,
                                       row_group_stride=stride), want, nthreads=16)
    return sharding.interleaved_rows(0, stride, h), want


def test_c3_kernel9_at_the_bench_shape_equals_the_oracle():
    """The kernel and launch shape bench.py times — C3, keyed PCG, kernel 9, 1920x1080, 256 spp in ONE launch, default
    wavefront budget (64 Mi slots, suspension, live-slot list, finishing kernel) — against the oracle at the same
    256 spp on three 8-row groups (top, middle, bottom); then the north star's 1024-spp frame as one launch (one
    sample chunk) against the oracle on one 8-row group.  Sample order and sum order per pixel are what
    RayTracing.slang:76-90, 170-179 prescribe and k_wf_resolve re-creates."""
    import bench
    w, h, depth = 1920, 1080, 8
    r = R.Renderer(w, h, device=0)
    cam, _ = bench.build_dragon_scene(R, r)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    for spp, stride in ((256, 67), (1024, 200)):
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, spp)
        got = _launch(r, cam, 9)
        assert _last_kernel(r) == 9
        rows, want = _oracle_rows(r, cam, w, h, int(spp ** 0.5), depth, T.RNG_PCG, stride)
        assert len(rows) == (24 if spp == 256 else 8)
        assert _same(got[rows], want[rows]), spp
    rounds, list_rounds, suspended, finished = _ffi_counts(r)
    assert (rounds, list_rounds, finished) == (2, 0, 0) and suspended > 0  # two launches without rounds; windows' last rays suspended
    # the same 256-spp launch with the paths dealt to the XCDs the other way: launches of this size take bands of the
    # frame per XCD (rt_debug_wavefront_bands), here interleaved
    ctx = C.c_void_p(r.hip_context_handle())
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 256)
    rows, want = _oracle_rows(r, cam, w, h, 16, depth, T.RNG_PCG, 67)
    _ffi._check(_ffi.lib().rt_debug_wavefront_bands(ctx, 0))
    got = _launch(r, cam, 9)
    assert _same(got[rows], want[rows])
    _ffi._check(_ffi.lib().rt_debug_wavefront_bands(ctx, 2))
    # and through the trace / shade rounds of rounds 2 and 3 (suspension, live-slot list, finishing kernel)
    _ffi._check(_ffi.lib().rt_debug_wavefront_persistent(ctx, 0))
    _ffi._check(_ffi.lib().rt_reset_stats(ctx))
    got = _launch(r, cam, 9)
    assert _same(got[rows], want[rows])
    rounds, list_rounds, suspended, finished = _ffi_counts(r)
    assert rounds > 2 and suspended > 0 and finished > 0
    r.shutdown()


def _ffi_counts(r):
    a, b, c, d = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    _ffi._check(_ffi.lib().rt_debug_wavefront_counts(C.c_void_p(r.hip_context_handle()), C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
    return a.value, b.value, c.value, d.value


def test_automatic_choice_survives_instance_updates():
    """The reference's per-frame path when objects move: set_blas_instance_transform -> TLAS rebuild -> render
    (SceneGraph.cpp:544-566, Renderer.cpp:901-916, 994-1015).  Seven launches with an instance moved before each: the
    first three are the trials (kernel 9 warm-up, kernel 9 timed, kernel 8 timed), the others run the winner — rt_update_instances does not
    restart the measurement — and every image equals the oracle's of the scene as it then stands."""
    w, h, depth, sqrt_spp = 1920, 1080, 8, 4
    r = R.Renderer(w, h, device=0)
    cam, _ = r.load_scene_file(os.path.join(SCENES, "teapot.json"))
    mirror = r.add_metal_material((0.8, 0.85, 0.9), 0.05)
    cube = r.add_blas_instance(r.cube_blas_index, R.trs((2.0, 0.5, 0.0)), mirror)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, sqrt_spp * sqrt_spp)  # 3.3e7 paths: eligible for kernel 9
    r.set_option(R.OPT_KERNEL_VARIANT, _ffi.KERNEL_VARIANT_AUTO)
    kernels = []
    for k in range(7):
        r.set_blas_instance_transform(cube, R.trs((2.0 - 0.5 * k, 0.5 + 0.1 * k, 0.3 * k), 0.2 * k, (0, 1, 0), (1, 1 + 0.1 * k, 1)))
        r.render(cam)
        got = r.read_image()
        kernels.append(_last_kernel(r))
        assert r.frame_index == 1  # the moved instance reset the accumulation (Renderer.cpp:491-494, 1014)
        rows, want = _oracle_rows(r, cam, w, h, sqrt_spp, depth, T.RNG_PCG, 67)
        assert _same(got[rows], want[rows]), k
    # (a moved object may change the launch time by more than the drift threshold and so repeat the two timed trials —
    #  what must NOT happen is a trial on every update: at most one re-trial pair in these four launches)
    assert kernels[:3] == [9, 9, 8] and all(k in (8, 9) for k in kernels[3:]), kernels
    assert kernels[3:] == [kernels[3]] * 4 or kernels[3:].count(kernels[3]) >= 2, kernels
    r.shutdown()


def test_kernel9_without_memory_falls_back_to_kernel8_in_automatic_mode():
    """kernel 9 wants gigabytes; when the device cannot give them (simulated: the hook makes its
    allocations fail) the automatic choice renders with kernel 8 instead of failing, and stays there; a caller that
    NAMED kernel 9 gets RT_ERR_OUT_OF_MEMORY; with memory back, kernel 9 renders the same bits."""
    w, h = 1920, 1080
    r = R.Renderer(w, h, device=0)
    cam, _ = r.load_scene_file(os.path.join(SCENES, "teapot.json"))
    r.set_option(R.OPT_MAX_DEPTH, 8)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 16)
    ctx = C.c_void_p(r.hip_context_handle())
    L = _ffi.lib()
    want = _launch(r, cam, 8)
    _ffi._check(L.rt_debug_fail_wavefront_allocs(ctx, 1000))
    a = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)
    assert _last_kernel(r) == 8 and _same(a, want)
    b = _launch(r, cam, _ffi.KERNEL_VARIANT_AUTO)          # settled: no second trial of kernel 9
    assert _last_kernel(r) == 8 and _same(b, want)
    r.set_option(R.OPT_KERNEL_VARIANT, 9)
    r.reset_accumulation()
    with pytest.raises(R.RendererError, match="memory"):
        r.render(cam)
    _ffi._check(L.rt_debug_fail_wavefront_allocs(ctx, 0))
    _ffi._check(L.rt_set_wavefront_budget(ctx, 1 << 22, 256 << 20))  # (the failures had halved the budget away)
    c = _launch(r, cam, 9)
    assert _last_kernel(r) == 9 and _same(c, want)
    r.shutdown()


def test_c3b_nonuniformly_tessellated_mesh_full_size():
    """A second dragon-class mesh whose triangles are NOT all of one size (teapot.obj subdivided to 1,048,576 triangles
    and displaced, bench.build_teapot_dragon_scene; the C3 tube is a regular grid): the same checks as C3 — kernel 0 ==
    default kernel, shares stitch, 40 oracle rows (Wang, 4 spp) — and the wavefront pipeline against the megakernel at
    64 spp and against the oracle at 4 spp (keyed PCG).  The BVH is deeper (leaf depth up to 41, tools/leaf_histogram.py)."""
    import bench
    w, h, depth = 1920, 1080, 8
    r = R.Renderer(w, h, device=0)
    cam, n_tris = bench.build_teapot_dragon_scene(R, r)
    assert n_tris == 1048578
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 4)
    _properties_and_oracle_rows(r, cam, w, h, 2, depth, T.RNG_WANG)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 64)
    a = _launch(r, cam, 9)
    assert _last_kernel(r) == 9 and _same(a, _launch(r, cam, 8))
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 4)
    b = _launch(r, cam, 9)
    rows, want = _oracle_rows(r, cam, w, h, 2, depth, T.RNG_PCG, 33)
    assert _same(b[rows], want[rows])
    r.shutdown()


def test_c4_many_instances_at_full_size():
    """C4 at 1920x1080 and 64 spp (the reduced-size whole-image check is test_gpu_parity.py::test_many_instances_c4_scale):
    2025 transformed instances, four material types — the property checks and 40 oracle rows of the other full-size
    configs."""
    import bench
    w, h, depth = 1920, 1080, 8
    r = R.Renderer(w, h, device=0)
    cam, _ = bench.build_instances_scene(R, r)
    r.set_option(R.OPT_MAX_DEPTH, depth)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 64)
    _properties_and_oracle_rows(r, cam, w, h, 8, depth, T.RNG_WANG)
    r.shutdown()


def test_automatic_choice_is_measured_again_when_the_launch_time_drifts():
    """the launch shape does not say what a launch costs — the camera does.  The Cornell box at
    1920x1080 / 16 spp (keyed PCG): trials, winner; then the camera turns away to the empty sky (every path ends at its
    first ray: the launch takes a fraction of the time).  The first launch after the turn still runs the old winner —
    its time is the evidence — and the next two are the new trials."""
    w, h = 1920, 1080
    r = R.Renderer(w, h, device=0)
    cam, _ = r.load_scene_file(os.path.join(SCENES, "cornell.json"))
    r.set_option(R.OPT_MAX_DEPTH, 8)
    r.set_option(R.OPT_RNG_MODE, T.RNG_PCG)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 16)
    r.set_option(R.OPT_KERNEL_VARIANT, _ffi.KERNEL_VARIANT_AUTO)
    kernels = []
    for _ in range(5):
        r.render(cam)
        r.wait()
        kernels.append(_last_kernel(r))
    assert kernels[:3] == [9, 9, 8] and kernels[3] == kernels[4], kernels
    sky = R.Camera(tuple(cam.c.position), (0.0, 60.0, 16.0), (0, 0, 1), cam.c.fov)  # straight up, away from the box
    for _ in range(5):
        r.render(sky)
        r.wait()
        kernels.append(_last_kernel(r))
    after = kernels[5:]
    assert after[0] == kernels[4], kernels                       # the evidence: the old winner, now much faster
    assert after[1:4] == ([9, 9, 8] if kernels[4] == 8 else [9, 8, after[3]]), kernels  # (kernel 8 had won: kernel 9's buffers
    assert all(k in (8, 9) for k in after), kernels              #  were released, so it warms up again before it is timed)
    r.shutdown()
