"""GPU parity: librt_hip.so (through its C ABI) against the CPU oracle.

Bar: (t, tri_id, inst_id, u, v) bit-exact on identical ray sets; rendered
rgba32f images bit-exact (the kernels and the oracle share every rounding, so
the PSNR >= 40 dB gate of BASELINE.md is met with equality)."""
import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from . import _oracle as O
from . import _scenes as S

pytestmark = pytest.mark.gpu

SCENES = {"teapot": (S.teapot_scene, S.teapot_camera), "mixed": (S.mixed_scene, S.mixed_camera),
          "cornell": (S.cornell_builtin, S.cornell_camera)}


@pytest.fixture(scope="module")
def ctx():
    c = _ffi.HipContext(0)
    yield c
    c.close()


def _same_hits(a, b):
    for f in ("t", "u", "v"):
        assert np.array_equal(a[f].view(np.uint32), b[f].view(np.uint32)), f
    assert np.array_equal(a["tri_id"], b["tri_id"]) and np.array_equal(a["inst_id"], b["inst_id"])


def _same_image(got, want):
    bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(-1))
    if len(bad):
        y, x = bad[0]
        raise AssertionError(f"{len(bad)} of {got.shape[0] * got.shape[1]} pixels differ; first at "
                             f"(x={x}, y={y}): got {got[y, x].tolist()} want {want[y, x].tolist()}")


@pytest.mark.parametrize("name", list(SCENES))
@pytest.mark.parametrize("mode", [T.AABB_DIV, T.AABB_RCP])
def test_hits_bit_exact_random_rays(ctx, name, mode):
    sc = SCENES[name][0]().build()
    ctx.upload_scene(sc)
    rays = S.random_rays(20000, seed=11, center=(0, 0.5, 0), radius=3.0)
    want = O.trace_closest(sc, rays, mode)
    got = ctx.trace_closest(rays, mode)
    assert (want["tri_id"] != T.INVALID_ID).sum() > 1000
    _same_hits(got, want)


@pytest.mark.parametrize("name", list(SCENES))
def test_hits_bit_exact_camera_and_bounce_rays(ctx, name):
    make, camf = SCENES[name]
    sc = make().build()
    ctx.upload_scene(sc)
    w, h = 96, 54
    cam = camf(w, h)
    pr = O.render_params(w, h, 3, 1, 4)
    for bounce in (0, 1, 2):
        rays = O.gen_rays(sc, cam, pr, bounce)
        if len(rays) == 0:
            continue
        _same_hits(ctx.trace_closest(rays), O.trace_closest(sc, rays))


def test_ray_edge_cases(ctx):
    sc = S.mixed_scene().build()
    ctx.upload_scene(sc)
    assert len(ctx.trace_closest(np.zeros(0, T.ray_dtype))) == 0  # empty batch
    rays = S.random_rays(257, seed=5)                             # ragged: not a multiple of the block
    rays["direction"][0] = (0, 0, 0)                              # degenerate direction: NaN slabs
    rays["direction"][1] = (0, -1, 0)
    rays["origin"][2] = (np.inf, 0, 0)
    rays["direction"][3] = (np.nan, 1, 0)
    rays["t_max"][4] = 1e-3                                       # range too short to reach anything
    rays["t_max"][5] = np.inf
    _same_hits(ctx.trace_closest(rays), O.trace_closest(sc, rays))


def test_empty_scene_misses(ctx):
    r = O.OracleRenderer()
    r.add_blas(*S.plane_mesh())
    sc = r.build()  # no instances: tlas_node_count == 0
    assert sc.tlas_node_count == 0
    ctx.upload_scene(sc)
    rays = S.random_rays(100, seed=1)
    got = ctx.trace_closest(rays)
    assert (got["tri_id"] == T.INVALID_ID).all() and np.array_equal(got["t"], rays["t_max"])


# kernel variants and their schedule knobs (rt_kernels.hip): all must give the same bits
# knobs of kernel 8: bits 8-15 regeneration threshold, 16-27 vote weights, 28-30 join threshold, 31 no fast slab path
VARIANTS = [0, 8, 8 | (1 << 28), 8 | (7 << 28), 8 | (1 << 31),
            8 | (4 << 8) | (9 << 16) | (1 << 20) | (1 << 24) | (2 << 28),
            8 | (60 << 8) | (1 << 16) | (15 << 20) | (15 << 24) | (5 << 28), _ffi.KERNEL_VARIANT_AUTO]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name,sqrt_spp,depth,rng", [
    ("cornell", 1, 3, T.RNG_WANG),   # the reference's own launch shape
    ("cornell", 3, 8, T.RNG_WANG),
    ("mixed", 2, 8, T.RNG_WANG),
    ("mixed", 2, 8, T.RNG_PCG),
    ("teapot", 2, 8, T.RNG_WANG),
])
def test_image_bit_exact(ctx, name, sqrt_spp, depth, rng, variant):
    make, camf = SCENES[name]
    sc = make().build()
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(variant)
    w, h = 100, 52  # not a multiple of the 16x16 tile
    cam = camf(w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    want = np.zeros((h, w, 4), np.float32)
    for frame in range(2):  # progressive accumulation across launches
        p = O.render_params(w, h, frame, sqrt_spp, depth, rng_mode=rng)
        ctx.render(p)
        O.render(sc, cam, p, want, nthreads=8)
    got = ctx.read_accum(w, h)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)  # back to the default
    assert np.isfinite(got).all()
    _same_image(got, want)


@pytest.mark.parametrize("variant", [8, 8 | (1 << 28)])  # the kernel with the fast slab path
@pytest.mark.parametrize("mode", [T.AABB_DIV, T.AABB_RCP])
@pytest.mark.parametrize("kind,cam_pos", [
    ("tiny_vertex", (0, 0.75, 1.5)), ("huge_vertex", (0, 0.75, 1.5)),
    ("squashed", (0.1, 0.75, 1.5)), ("stretched", (0.1, 0.75, 1.5)),
    ("squashed", (0, 0.75, 1.5)),        # origin.x exactly 0 (allowed on the fast path)
    ("squashed", (1e-30, 0.75, 1.5)),    # origin.x tiny but not 0 (not allowed)
    ("stretched", (0, 1e25, 1.5)),       # origin beyond 2^59
])
def test_image_bit_exact_outside_fast_slab_preconditions(ctx, kind, cam_pos, mode, variant):
    """slab_pair_fast (rt_device_inl.h) may only be used where it is provably the IEEE
    division; lanes and scenes outside its preconditions must take the literal slab test."""
    sc = S.edge_scene(kind).build()
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(variant)
    w, h = 72, 40
    cam = O.camera_uniforms(cam_pos, (0, 0.75, -1), (0, 1, 0), 90.0, w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    p = O.render_params(w, h, 0, 2, 4, aabb_mode=mode)
    ctx.render(p)
    want = O.render(sc, cam, p, nthreads=8)
    got = ctx.read_accum(w, h)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)  # back to the default
    _same_image(got, want)


@pytest.mark.parametrize("variant", VARIANTS)
def test_rays_counted_and_rows(ctx, variant):
    sc = S.cornell_builtin().build()
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(variant)
    w, h = 64, 40
    cam = S.cornell_camera(w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    ctx.reset_stats()
    cn = O.Counters()
    p = O.render_params(w, h, 0, 2, 5)
    want = O.render(sc, cam, p, counters=cn, nthreads=4)
    # two row bands == whole frame
    ctx.render(O.render_params(w, h, 0, 2, 5, row_begin=0, row_end=17))
    ctx.render(O.render_params(w, h, 0, 2, 5, row_begin=17, row_end=h))
    got = ctx.read_accum(w, h)
    _same_image(got, want)
    st = ctx.stats()
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)  # back to the default
    assert st.rays_traced == cn.rays and st.launches == 2 and st.total_kernel_ms > 0


def test_update_instances_moves_object(ctx):
    r = S.cornell_builtin()
    sc = r.build()
    ctx.upload_scene(sc)
    rays = S.random_rays(5000, seed=3, center=(0, 1, 0), radius=1.5)
    before = ctx.trace_closest(rays)
    r.set_blas_instance_transform(6, O.trs((0.0, 0.9, 0.2), (1, 0, 0, 0), (0.5, 0.5, 0.5)))
    sc2 = r.build()
    ctx.update_instances(sc2.instances, sc2.tlas_nodes, sc2.tlas_node_count)
    after = ctx.trace_closest(rays)
    _same_hits(after, O.trace_closest(sc2, rays))
    assert not np.array_equal(before["t"], after["t"])


def test_bad_input_is_rejected_not_launched(ctx):
    sc = S.teapot_scene().build()
    bad = sc.bvh_nodes.copy()
    bad["local_left_first"][0] = 10 ** 6  # child index outside the BLAS
    sc_bad = T.SceneArrays(sc.tri_geom, sc.tri_shading, sc.tri_ids, bad, sc.blas, sc.instances,
                           sc.tlas_nodes, sc.tlas_node_count, sc.lamberts, None, None, None,
                           sc.textures, sc.texels)
    with pytest.raises(_ffi.RtError):
        ctx.upload_scene(sc_bad)
    with pytest.raises(_ffi.RtError):
        ctx.trace_closest(S.random_rays(4, 1))  # failed upload leaves no scene


def test_bad_launch_parameters_are_rejected(ctx):
    """A row_group_stride whose 8-fold wraps to 0 (the host would divide by it) or beyond 2^16 groups,
    an unknown kernel number and an empty row range are refused with RT_ERR_INVALID_ARGUMENT, nothing is launched."""
    sc = S.cornell_builtin().build()
    ctx.upload_scene(sc)
    w, h = 64, 40
    ctx.set_camera(S.cornell_camera(w, h))
    ctx.resize(w, h)
    ctx.reset_stats()
    for stride in (1 << 29, 1 << 30, (1 << 16) + 1):
        with pytest.raises(_ffi.RtError):
            ctx.render(O.render_params(w, h, 0, 1, 3, row_begin=0, row_end=h, row_group_stride=stride))
    with pytest.raises(_ffi.RtError):
        ctx.render(O.render_params(w, h, 0, 1, 3, row_begin=9, row_end=9))
    for v in (1, 2, 3, 4, 5, 6, 7, 10, 200):  # round 1's other kernels are no longer in the library
        with pytest.raises(_ffi.RtError):
            ctx.set_kernel_variant(v)
    assert ctx.stats().launches == 0
    ctx.render(O.render_params(w, h, 0, 1, 3, row_begin=0, row_end=h, row_group_stride=1 << 16))  # the largest legal stride
    assert ctx.stats().launches == 1


@pytest.mark.gpu
def test_fast_division_is_ieee_division(ctx):
    """slab_pair_fast's division (rt_device_inl.h: refined reciprocal + two residual
    corrections, packed fp32) must give the correctly rounded quotient wherever its
    preconditions hold: |d| in [2^-30, 2^30], n == 0 or |n| in [2^-90, 2^60]."""
    rs = np.random.RandomState(77)
    n_rand = 1 << 22

    def rand_floats(count, emin, emax):
        mant = rs.randint(0, 1 << 23, size=count).astype(np.uint32)
        expo = rs.randint(emin + 127, emax + 127, size=count).astype(np.uint32)  # emax exclusive
        sign = rs.randint(0, 2, size=count).astype(np.uint32)
        return ((sign << 31) | (expo << 23) | mant).view(np.float32)

    d = rand_floats(n_rand, -30, 30)
    n = rand_floats(n_rand, -90, 60)
    # adversarial: mantissas of all ones / all zeros, quotients next to rounding boundaries
    edge_m = np.array([0, 1, 2, 0x7FFFFF, 0x7FFFFE, 0x400000, 0x3FFFFF, 0x555555, 0x2AAAAA], np.uint32)
    ee = []
    for ma in edge_m:
        for mb in edge_m:
            for ea in (-90, -40, -1, 0, 1, 59):
                for eb in (-30, -1, 0, 1, 29):
                    ee.append((((ea + 127) << 23) | int(ma), ((eb + 127) << 23) | int(mb)))
    ee = np.array(ee, np.uint32)
    n = np.concatenate([n, ee[:, 0].view(np.float32), -ee[:, 0].view(np.float32), np.zeros(8, np.float32)])
    d = np.concatenate([d, ee[:, 1].view(np.float32), ee[:, 1].view(np.float32),
                        np.array([1, -1, 3, -3, 2.0 ** -30, 2.0 ** 29, 0.1, -7e8], np.float32)])
    # products n = q*d rounded: quotients that sit on or next to representable values
    q = rand_floats(1 << 20, -20, 20)
    d2 = rand_floats(1 << 20, -20, 20)
    n = np.ascontiguousarray(np.concatenate([n, (q * d2).astype(np.float32)]))
    d = np.ascontiguousarray(np.concatenate([d, d2]))
    q_fast, q_ieee = ctx.debug_fast_div(n, d)
    want = (n.astype(np.float32) / d.astype(np.float32)).astype(np.float32)  # numpy fp32 division is IEEE
    assert np.array_equal(q_ieee.view(np.uint32), want.view(np.uint32))
    nz = want != 0
    assert np.array_equal(q_fast.view(np.uint32)[nz], want.view(np.uint32)[nz])
    assert np.all(q_fast[~nz] == 0)  # the sign of a zero quotient is not part of the contract


def test_many_instances_c4_scale(ctx):
    """BASELINE config C4 in its size class: 2025 transformed instances (TLAS of 4049 nodes) of
    teapots, spheres and cubes with all four material types — hits and image bit-exact."""
    sc = S.mixed_scene(n_side=45, seed=7).build()
    assert sc.tlas_node_count == 2 * 2026 - 1
    ctx.upload_scene(sc)
    rays = S.random_rays(20000, seed=11, center=(0, 0.3, 0), radius=45.0)
    _same_hits(ctx.trace_closest(rays), O.trace_closest(sc, rays))
    w, h = 240, 135
    cam = O.camera_uniforms((30, 22, 40), (0, 0, 0), (0, 1, 0), 45.0, w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    for rng in (T.RNG_WANG, T.RNG_PCG):
        ctx.clear_accum()
        p = O.render_params(w, h, 0, 2, 8, rng_mode=rng)
        ctx.render(p)
        _same_image(ctx.read_accum(w, h), O.render(sc, cam, p, nthreads=8))


def test_defocus_blur_c5_camera(ctx):
    """BASELINE config C5's camera feature: defocus blur (Random.slang:57-70 semantics, an extension over
    the reference's pinhole kernel) with the keyed PCG RNG, at a 3840-wide image rendered in two row bands
    the way a tile-sharded multi-GPU run would."""
    from raytracing_amd import renderer as R
    sc = S.mixed_scene().build()
    ctx.upload_scene(sc)
    w, h = 3840, 24
    host = R.Renderer(w, h, device=None)
    cam = host.camera_uniforms(R.Camera((3.5, 2.5, 4.5), (0, 0, 0), (0, 1, 0), 12.0, defocus_angle=1.5, focus_dist=6.0))
    assert cam.defocus_disk_u[3] == 1.0
    ctx.set_camera(cam)
    ctx.resize(w, h)
    for rng in (T.RNG_WANG, T.RNG_PCG):
        ctx.clear_accum()
        p = O.render_params(w, h, 0, 2, 5, rng_mode=rng)
        for rows in ((0, 11), (11, 24)):
            pr = O.render_params(w, h, 0, 2, 5, rng_mode=rng)
            pr.row_begin, pr.row_end = rows
            ctx.render(pr)
        _same_image(ctx.read_accum(w, h), O.render(sc, cam, p, nthreads=8))


@pytest.mark.parametrize("sqrt_spp,depth", [(1, 3), (2, 3), (1, 8), (4, 3)])
def test_automatic_kernel_choice_is_bit_exact(sqrt_spp, depth):
    """A new context chooses the kernel itself (RT_KERNEL_VARIANT_AUTO): the one-lane-per-pixel kernel for the
    reference's live configuration (tiny scene, <= 4 samples, depth <= 3), the vote-scheduled one otherwise."""
    c = _ffi.HipContext(0)
    try:
        make, camf = SCENES["cornell"]
        sc = make().build()
        c.upload_scene(sc)
        w, h = 100, 52
        cam = camf(w, h)
        c.set_camera(cam)
        c.resize(w, h)
        want = np.zeros((h, w, 4), np.float32)
        for frame in range(2):
            p = O.render_params(w, h, frame, sqrt_spp, depth)
            c.render(p)
            O.render(sc, cam, p, want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
    finally:
        c.close()


def _chain_scene(n=38, e0=-28):
    """n cubes at x = 2^(i + e0), each stretched to an eighth of its distance from the origin: the reference's
    agglomerative TLAS builder (TLAS.cpp:38-58) chains them — ((((0 1) 2) 3) ...) — so a ray along +x pushes one far
    instance per TLAS level before it enters its first BLAS: 37 stack entries, beyond the 24 the kernels keep in LDS."""
    r = O.OracleRenderer()
    cube = r.add_blas(*S.cube_mesh())
    mats = [r.add_lambert_material((0.6, 0.5, 0.4)), r.add_metal_material((0.8, 0.8, 0.9), 0.05)]
    for i in range(n):
        x = float(2.0 ** (i + e0))
        r.add_blas_instance(cube, O.trs((x, 0, 0), (1, 0, 0, 0), (x / 8, 1, 1)), mats[i & 1])
    return r


@pytest.mark.parametrize("variant", [0, 8, 8 | (1 << 28), 8 | (7 << 28)])
def test_stack_deeper_than_the_lds_part(ctx, variant):
    """Traversal stacks that reach into the per-lane spill arrays (more than RT_LDS_STACK = 24 entries): hits and a
    rendered image stay bit-exact for every kernel, the wave-uniform spill tests of the default kernel included."""
    sc = _chain_scene().build()
    rng = np.random.default_rng(7)
    n = 4096
    rays = np.zeros(n, T.ray_dtype)
    rays["origin"] = np.stack([np.full(n, -1.0), rng.uniform(-0.6, 0.6, n), rng.uniform(-0.6, 0.6, n)], -1).astype(np.float32)
    rays["direction"] = np.stack([np.ones(n), rng.normal(scale=2e-4, size=n), rng.normal(scale=2e-4, size=n)], -1).astype(np.float32)
    rays["origin"][::7] = (600.0, 0.1, -0.2)
    rays["direction"][::7, 0] = -1.0
    rays["t_min"], rays["t_max"] = np.float32(1e-4), np.float32(1000.0)
    cn = O.Counters()
    want = O.trace_closest(sc, rays, counters=cn)
    assert cn.max_stack > 30  # the premise of this test
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(variant)
    got = ctx.trace_closest(rays)
    for k in ("t", "tri_id", "inst_id"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    w, h = 96, 40
    cam = O.camera_uniforms((-1.0, 0.2, 0.1), (600.0, 0.0, 0.0), (0, 1, 0), 30.0, w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    img = np.zeros((h, w, 4), np.float32)
    cn2 = O.Counters()
    p = O.render_params(w, h, 0, 2, 6)
    ctx.render(p)
    O.render(sc, cam, p, img, counters=cn2, nthreads=8)
    assert cn2.max_stack > 30
    got_img = ctx.read_accum(w, h)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
    _same_image(got_img, img)


@pytest.mark.parametrize("variant", [0, 8])
@pytest.mark.parametrize("rng", [T.RNG_WANG, T.RNG_PCG])
def test_dielectric_condition_both_readings(ctx, variant, rng):
    """Material.slang:91 `cannot_refract || reflectance(..) > rand(seed)` admits two readings (does a ray that
    cannot refract still draw a random number?) and the reference cannot be run to decide: both are selectable
    (rt_render_params.flags, RT_RENDER_DIELECTRIC_ALWAYS_DRAWS), both equal the oracle, and they differ from each
    other under the sequential Wang RNG, where a skipped draw shifts every later draw of the pixel."""
    sc = S.mixed_scene().build()
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(variant)
    w, h = 100, 52
    cam = S.mixed_camera(w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    imgs = []
    try:
        for flags in (0, T.RENDER_DIELECTRIC_ALWAYS_DRAWS):
            ctx.clear_accum()
            p = O.render_params(w, h, 0, 3, 8, rng_mode=rng, flags=flags)
            ctx.render(p)
            got = ctx.read_accum(w, h)
            _same_image(got, O.render(sc, cam, p, nthreads=8))
            imgs.append(got)
        with pytest.raises(_ffi.RtError):
            ctx.render(O.render_params(w, h, 0, 1, 3, flags=4))  # unknown flag bit
    finally:
        ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
    differ = not np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    # keyed PCG: the dielectric's draw is the only one of its bounce and the next bounce has its own key, so an
    # unused extra draw changes nothing — the two readings only matter for the reference's sequential RNG
    assert differ == (rng == T.RNG_WANG)


def test_caller_supplied_instance_order_flows_through(ctx):
    """The reference lists its TLAS leaves in std::unordered_set iteration order (Renderer.cpp:999-1000), which is
    implementation-defined; oracle and product default to ascending id.  A caller holding the reference's arrays
    hands over ITS order: the TLAS built from a permuted instance list (another tree shape) goes through
    rt_upload_scene unchanged — hits and image equal the oracle's on the same arrays, and t is the same as with the
    ascending order for every ray (only ids of equal-t ties may move)."""
    r = S.mixed_scene(n_side=5, seed=3)
    n = len(r.instances)
    perm = np.random.default_rng(99).permutation(n).astype(np.uint32)
    sc_asc, sc_perm = r.build(), r.build(instance_ids=perm)
    cnt = sc_asc.tlas_node_count
    assert sc_perm.tlas_node_count == cnt
    assert sc_perm.tlas_nodes[:cnt].tobytes() != sc_asc.tlas_nodes[:cnt].tobytes()
    rays = S.random_rays(20000, seed=5, center=(0, 0.5, 0), radius=5.0)
    ctx.upload_scene(sc_asc)
    hits_asc = ctx.trace_closest(rays)
    ctx.upload_scene(sc_perm)
    hits_perm = ctx.trace_closest(rays)
    _same_hits(hits_perm, O.trace_closest(sc_perm, rays))
    assert np.array_equal(hits_perm["t"].view(np.uint32), hits_asc["t"].view(np.uint32))
    moved = (hits_perm["tri_id"] != hits_asc["tri_id"]) | (hits_perm["inst_id"] != hits_asc["inst_id"])
    assert moved.mean() < 0.01
    w, h = 96, 54
    cam = S.mixed_camera(w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    p = O.render_params(w, h, 0, 2, 6)
    ctx.render(p)
    _same_image(ctx.read_accum(w, h), O.render(sc_perm, cam, p, nthreads=8))


def test_hits_against_brute_force_on_a_dragon_submesh(ctx):
    """Closest hits of the GPU traversal against orc_trace_brute — every triangle of every instance tested in turn,
    no BVH, no stack, no slab test: an oracle path that shares no traversal code with the kernels — on a coarse
    version of the bench mesh (the C3 dragon stand-in at 120 x 32 segments) in two instances."""
    from raytracing_amd import renderer as R
    mesh = R.Mesh.dragon(120, 32, 1234)
    r = O.OracleRenderer()
    b = r.add_blas(*mesh.arrays())
    plane = r.add_blas(*S.plane_mesh())
    r.add_blas_instance(b, O.trs(), r.default_material)
    r.add_blas_instance(b, O.trs((1.2, 0.1, -0.4), O.angle_axis(0.7, (0, 1, 0)), (0.6, 0.6, 0.6)), r.default_material)
    r.add_blas_instance(plane, O.trs((0, -0.28, 0), (1, 0, 0, 0), (14, 1, 14)), r.default_material)
    sc = r.build()
    assert mesh.triangle_count > 7000
    ctx.upload_scene(sc)
    rays = S.random_rays(6000, seed=13, center=(0.3, 0.4, 0), radius=2.5)
    want = O.trace_brute(sc, rays)
    for mode in (T.AABB_DIV, T.AABB_RCP):
        got = ctx.trace_closest(rays, mode)
        assert (got["tri_id"] != T.INVALID_ID).sum() > 1000
        assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32))
        # ids may differ only where two triangles give the same t (shared edges: the brute force keeps the
        # first such triangle in array order, the reference's traversal the last one it visits)
        diff = (got["tri_id"] != want["tri_id"]) | (got["inst_id"] != want["inst_id"])
        assert diff.mean() < 0.02


# ---- kernel 9: the sample-parallel wavefront pipeline (rt_wavefront.hip), keyed PCG RNG only ----
# knobs of kernel 9: bits 8-14 refill threshold, 15 quad-cooperative LDS-DMA record fetch, 16-30 as kernel 8, 31 no fast slab
@pytest.mark.parametrize("knobs", [0, (4 << 8), (60 << 8) | (1 << 16) | (15 << 20) | (15 << 24) | (5 << 28), 1 << 31,
                                   1 << 15, (1 << 15) | (1 << 31), (1 << 15) | (8 << 8) | (1 << 28)])
@pytest.mark.parametrize("name,sqrt_spp,depth", [("cornell", 1, 3), ("cornell", 3, 8), ("mixed", 2, 8), ("teapot", 4, 8)])
def test_wavefront_image_bit_exact(ctx, name, sqrt_spp, depth, knobs):
    make, camf = SCENES[name]
    sc = make().build()
    ctx.upload_scene(sc)
    ctx.set_kernel_variant(9 | knobs)
    w, h = 100, 52
    cam = camf(w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    want = np.zeros((h, w, 4), np.float32)
    try:
        for frame in range(2):
            p = O.render_params(w, h, frame, sqrt_spp, depth, rng_mode=T.RNG_PCG)
            ctx.render(p)
            O.render(sc, cam, p, want, nthreads=8)
        got = ctx.read_accum(w, h)
        assert ctx.stats().last_kernel == 9
    finally:
        ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
    _same_image(got, want)


@pytest.mark.parametrize("slots,result_bytes", [(256, 1 << 20), (1000, 3 << 20), (1 << 16, 1 << 20), (1 << 20, 1 << 30)])
def test_wavefront_small_budgets_many_rounds_and_chunks(slots, result_bytes):
    """Few path slots (many rounds, every slot reused often) and a small result buffer (the launch runs in several
    sample chunks whose partial sums continue each other) give the same bits; also rows, row interleaving, defocus,
    RCP slabs, depth 0 and the rays counter."""
    c = _ffi.HipContext(0)
    try:
        sc = S.mixed_scene().build()
        c.upload_scene(sc)
        c.set_kernel_variant(9)
        c.set_wavefront_budget(slots, result_bytes)
        from raytracing_amd import renderer as R
        w, h = 136, 76
        host = R.Renderer(w, h, device=None)
        cam = host.camera_uniforms(R.Camera((3.5, 2.5, 4.5), (0, 0, 0), (0, 1, 0), 45.0, defocus_angle=1.0, focus_dist=6.0))
        c.set_camera(cam)
        c.resize(w, h)
        c.reset_stats()
        cn = O.Counters()
        p = O.render_params(w, h, 0, 3, 6, rng_mode=T.RNG_PCG)
        want = O.render(sc, cam, p, counters=cn, nthreads=8)
        c.render(p)
        _same_image(c.read_accum(w, h), want)
        assert c.stats().rays_traced == cn.rays
        # row bands and interleaved groups compose to the same frame; RCP slabs; a second frame on top
        c.clear_accum()
        for rows in ((0, 24, 2), (8, 24, 2), (24, h, 0)):
            c.render(O.render_params(w, h, 0, 3, 6, rng_mode=T.RNG_PCG, row_begin=rows[0], row_end=rows[1], row_group_stride=rows[2]))
        _same_image(c.read_accum(w, h), want)
        p1 = O.render_params(w, h, 1, 2, 4, rng_mode=T.RNG_PCG, aabb_mode=T.AABB_RCP)
        c.render(p1)
        O.render(sc, cam, p1, want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
        p2 = O.render_params(w, h, 2, 2, 0, rng_mode=T.RNG_PCG)  # depth 0: the frame contributes black
        c.render(p2)
        O.render(sc, cam, p2, want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
        # the reference's sequential RNG on kernel 9 (round 3): a frame's samples run one after the other inside a slot
        # (only frame batches give it parallelism, tests/test_gpu_frames.py); a single frame is simply correct
        acc = c.read_accum(w, h)
        pw = O.render_params(w, h, 7, 2, 5, rng_mode=T.RNG_WANG)
        pw.accum_count = 3
        c.render(pw)
        O.render(sc, cam, pw, want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
        c.clear_accum()  # back to where the checks below expect the image: frames 0 .. 2
        want[:] = 0
        for q in (O.render_params(w, h, 0, 3, 6, rng_mode=T.RNG_PCG), p1, p2):
            c.render(q)
            O.render(sc, cam, q, want, nthreads=8)
        assert np.array_equal(c.read_accum(w, h).view(np.uint32), acc.view(np.uint32))
        with pytest.raises(_ffi.RtError, match="2\\^28"):  # the depth word's top bits are kernel 9's flags
            c.render(O.render_params(w, h, 3, 1, 1 << 28, rng_mode=T.RNG_PCG))
        c.render(O.render_params(w, h, 3, 1, (1 << 28) - 1, rng_mode=T.RNG_PCG))  # the largest depth it takes
        O.render(sc, cam, O.render_params(w, h, 3, 1, (1 << 28) - 1, rng_mode=T.RNG_PCG), want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
    finally:
        c.close()


@pytest.mark.parametrize("suspend,listp,share,finish", [(0, 0, 256, 0), (64, 0, 64, 0), (0, 100, 256, 0), (64, 100, 64, 0),
                                                         (32, 75, 256, 0), (8, 50, 512, 300), (64, 100, 64, 3000),
                                                         (0, 100, 256, 1 << 30)])
@pytest.mark.parametrize("scene", ["mixed", "chain"])
@pytest.mark.parametrize("persistent", [False, True])
def test_wavefront_suspended_rays_live_slot_list_and_finisher(scene, suspend, listp, share, finish, persistent):
    """Kernel 9's round scheduling — trace waves that give their last rays up (traversal state saved to HBM, resumed
    next round, also from inside an instance and with stack entries beyond the LDS part) and trace passes that walk the
    list of live slots, and the kernel that sees the last paths through (from a few hundred left to everything after
    the first thinning round, new paths of unfinished shares included) — changes neither the image nor the ray count,
    whatever the thresholds; the counters show that the paths under test were taken.  `persistent`: the same settings
    on the mode without rounds (the default), where only the suspension threshold applies — a wave gives the last rays
    of its window up and resumes them in its next trace phase."""
    c = _ffi.HipContext(0)
    try:
        if scene == "mixed":
            sc = S.mixed_scene().build()
            w, h = 136, 76
            from raytracing_amd import renderer as R
            cam = R.Renderer(w, h, device=None).camera_uniforms(R.Camera((3.5, 2.5, 4.5), (0, 0, 0), (0, 1, 0), 45.0))
        else:
            sc = _chain_scene().build()  # traversal stacks up to 37 entries: 21 of them beyond the 16 in LDS
            w, h = 96, 40
            cam = O.camera_uniforms((-1.0, 0.2, 0.1), (600.0, 0.0, 0.0), (0, 1, 0), 30.0, w, h)
        c.upload_scene(sc)
        c.set_kernel_variant(9)
        c.set_wavefront_budget(8192, 1 << 30)  # 8 Ki slots for 93 K / 35 K paths: full rounds, then the thinning ones
        c.wavefront_tuning(suspend, listp, share, finish)
        c.wavefront_persistent(persistent)
        c.wavefront_bands(1 if persistent and suspend != 8 else 2)  # (persistent: a band of the frame per XCD, also on row groups)
        c.set_camera(cam)
        c.resize(w, h)
        c.reset_stats()
        cn = O.Counters()
        p = O.render_params(w, h, 0, 3, 8, rng_mode=T.RNG_PCG)
        want = O.render(sc, cam, p, counters=cn, nthreads=8)
        c.render(p)
        _same_image(c.read_accum(w, h), want)
        assert c.stats().rays_traced == cn.rays  # a resumed ray is not counted twice
        rounds, list_rounds, suspended, finished = c.wavefront_counts()
        if persistent:
            assert (rounds, list_rounds, finished) == (1, 0, 0)  # one launch of the trace kernel per chunk, nothing else
        else:
            assert rounds >= (8 if finish < (1 << 30) else 2)
            if finish < (1 << 30):  # (with everything handed to the finisher no trace pass is left to walk the list)
                assert (list_rounds > 0) == (listp > 0), (rounds, list_rounds)
            assert (finished > 0) == (finish > 0), (rounds, finished)
        assert (suspended > 0) == (suspend > 0), (rounds, suspended)
        # a second frame on top through the same state buffers, in two interleaved shards
        for rb in (0, 8):
            c.render(O.render_params(w, h, 1, 2, 5, rng_mode=T.RNG_PCG, row_begin=rb, row_end=h, row_group_stride=2))
        O.render(sc, cam, O.render_params(w, h, 1, 2, 5, rng_mode=T.RNG_PCG), want, nthreads=8)
        _same_image(c.read_accum(w, h), want)
    finally:
        c.close()


@pytest.mark.parametrize("rng,spp_side,frames", [(T.RNG_PCG, 3, 1), (T.RNG_WANG, 1, 7), (T.RNG_WANG, 2, 3)])
@pytest.mark.parametrize("slots,suspend", [(8192, 0xFFFFFFFF), (8256, 64), (4160, 16)])
def test_wavefront_compaction_moves_paths_not_results(rng, spp_side, frames, slots, suspend):
    """The shade pass of kernel 9 re-orders the paths of every 256-slot window (those that go on, the new camera rays,
    the dead) so that trace waves hold coherent rays.  Which slot a path lives in must not matter: with the
    re-ordering and without it the image is the oracle's bit for bit and the ray count the same — also when slots are
    pinned (suspended rays: `suspend` 64 gives every wave's last rays up), when the last window is only partly inside
    the slot space (8256 = 32 windows + 64 slots) and for the reference's RNG, whose state and — at spp > 1, where the
    re-ordering is off by design — per-slot sample sum ride along.  Likewise for WHERE a pair record is read from: the
    trace kernel's LDS copy of the scene's top records (default) against every record through the L1."""
    c = _ffi.HipContext(0)
    try:
        sc = S.mixed_scene().build()
        w, h = 136, 76
        from raytracing_amd import renderer as R
        cam = R.Renderer(w, h, device=None).camera_uniforms(R.Camera((3.5, 2.5, 4.5), (0, 0, 0), (0, 1, 0), 45.0))
        c.upload_scene(sc)
        c.set_kernel_variant(9)
        c.set_wavefront_budget(slots, 1 << 30)
        c.wavefront_tuning(suspend, 75, 256, 300)
        c.set_camera(cam)
        c.resize(w, h)
        p = O.render_params(w, h, 2, spp_side, 8, rng_mode=rng, accum_count=0)
        want = np.zeros((h, w, 4), np.float32)
        cn = O.Counters()
        for f in range(frames):
            q = O.render_params(w, h, 2 + f, spp_side, 8, rng_mode=rng)
            q.accum_count = f
            O.render(sc, cam, q, want, counters=cn, nthreads=8)
        images, rays = [], []
        for on, cache, persistent, bands in ((True, True, False, 2), (False, False, False, 2), (True, False, False, 2), (True, True, True, 0),
                                             (True, True, True, 1)):
            c.wavefront_bands(bands)  # persistent mode: interleaved / a band of the frame per XCD (by default: large launches only)
            c.wavefront_compaction(on)
            c.wavefront_record_cache(cache)  # the trace kernel's LDS copy of the top records (and its 512-lane workgroups)
            c.wavefront_persistent(persistent)  # no rounds: every wave fills, traces and shades its own window (the default)
            c.clear_accum()
            c.reset_stats()
            c.render_frames(p, frames)
            images.append(c.read_accum(w, h))
            rays.append(c.stats().rays_traced)
            assert c.stats().last_kernel == 9
        assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
        assert np.array_equal(images[0].view(np.uint32), images[2].view(np.uint32))
        assert np.array_equal(images[0].view(np.uint32), images[3].view(np.uint32))
        assert np.array_equal(images[0].view(np.uint32), images[4].view(np.uint32))
        _same_image(images[0], want)
        assert rays[0] == rays[1] == rays[2] == rays[3] == rays[4] == cn.rays
        c.wavefront_bands(2)
    finally:
        c.close()


@pytest.mark.parametrize("kind,cam_pos", [("tiny_vertex", (0, 0.75, 1.5)), ("stretched", (0, 1e25, 1.5)), ("squashed", (1e-30, 0.75, 1.5))])
def test_edge_scenes_every_kernel_rng_and_mode(ctx, kind, cam_pos):
    """Scenes outside the fast slab test's preconditions, whose images are full of infinities and NaNs (a NaN's sign and
    payload depend on operand order, so this also pins code generation): every kernel, both RNGs (the reference's at one
    and at four samples per frame: kernel 9's kinds 1 and 2), kernel 9 with and without rounds, both slab tests —
    the oracle's bits."""
    sc = S.edge_scene(kind).build()
    ctx.upload_scene(sc)
    w, h = 72, 40
    cam = O.camera_uniforms(cam_pos, (0, 0.75, -1), (0, 1, 0), 90.0, w, h)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    try:
        for rng_mode in (T.RNG_PCG, T.RNG_WANG):
            for sqrt_spp in (1, 2):
                for mode in (T.AABB_DIV, T.AABB_RCP):
                    p = O.render_params(w, h, 0, sqrt_spp, 4, rng_mode=rng_mode, aabb_mode=mode)
                    want = O.render(sc, cam, p, nthreads=8)
                    for variant, persistent in ((0, True), (8, True), (9, True), (9, False)):
                        ctx.set_kernel_variant(variant)
                        ctx.wavefront_persistent(persistent)
                        ctx.clear_accum()
                        ctx.render(p)
                        got = ctx.read_accum(w, h)
                        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rng_mode, sqrt_spp, mode, variant, persistent)
    finally:
        ctx.wavefront_persistent(True)
        ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)


@pytest.mark.parametrize("coop", [0, 1])
def test_wavefront_deep_stack_and_edge_scenes(ctx, coop):
    """Kernel 9's trace kernel keeps 16 stack entries in LDS: the 38-instance chain drives it to 37; scenes outside the
    fast slab preconditions take its literal slab test."""
    ctx.set_kernel_variant(9 | (coop << 15))
    try:
        sc = _chain_scene().build()
        ctx.upload_scene(sc)
        w, h = 96, 40
        cam = O.camera_uniforms((-1.0, 0.2, 0.1), (600.0, 0.0, 0.0), (0, 1, 0), 30.0, w, h)
        ctx.set_camera(cam)
        ctx.resize(w, h)
        cn = O.Counters()
        p = O.render_params(w, h, 0, 2, 6, rng_mode=T.RNG_PCG)
        ctx.render(p)
        want = O.render(sc, cam, p, counters=cn, nthreads=8)
        assert cn.max_stack > 30
        _same_image(ctx.read_accum(w, h), want)
        for kind, cam_pos in (("tiny_vertex", (0, 0.75, 1.5)), ("stretched", (0, 1e25, 1.5)), ("squashed", (1e-30, 0.75, 1.5))):
            sc = S.edge_scene(kind).build()
            ctx.upload_scene(sc)
            w, h = 72, 40
            cam = O.camera_uniforms(cam_pos, (0, 0.75, -1), (0, 1, 0), 90.0, w, h)
            ctx.set_camera(cam)
            ctx.resize(w, h)
            for mode in (T.AABB_DIV, T.AABB_RCP):
                ctx.clear_accum()
                p = O.render_params(w, h, 0, 2, 4, rng_mode=T.RNG_PCG, aabb_mode=mode)
                ctx.render(p)
                _same_image(ctx.read_accum(w, h), O.render(sc, cam, p, nthreads=8))
    finally:
        ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
