"""The HIP path against the reference's own render (Screenshot3.png; see tests/test_reference_screenshot.py and
tests/_screenshot.py for what that image is and what the comparison can and cannot pin): assets/scenes/cornell.json
at the screenshot's own settings — 1920x1080, 961 spp (u32(sqrt(1000))^2), depth 50 — through the C ABI, with the
sky of the revision that rendered it (RT_RENDER_SKY_WHITE_HORIZON) and its PNG transform, in absolute 8-bit levels.
"""
import os

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from raytracing_amd import renderer as R
from . import _oracle as O
from . import _scenes as S
from . import _screenshot as SS

pytestmark = pytest.mark.gpu

SCENE = os.path.join(S.ASSETS, "scenes", "cornell.json")


@pytest.fixture(scope="module")
def cornell():
    """The scene file through the host library (loader, builders), then straight onto the C ABI of librt_hip: the
    camera uniforms handed to rt_set_camera carry the screenshot's pixel grid (tests/_screenshot.py, (c))."""
    host = R.Renderer(SS.W, SS.H, device=None)
    cam, info = host.load_scene_file(SCENE)
    assert (info["screen_width"], info["screen_height"]) == (SS.W, SS.H)
    sc = host.scene_arrays()
    camu = SS.screenshot_camera(host.camera_uniforms(cam))
    ctx = _ffi.HipContext(0)
    ctx.upload_scene(sc)
    ctx.set_camera(camu)
    ctx.resize(SS.W, SS.H)
    yield ctx, sc, camu
    ctx.close()


def test_hip_primary_hits_draw_the_screenshots_silhouettes(cornell):
    """rt_trace_closest through every pixel centre: hits bit-equal to the oracle's, and the boundaries of the surfaces
    they see lie on the screenshot's edges to the pixel."""
    ctx, sc, camu = cornell
    rays = SS.pixel_centre_rays(camu)
    hits = ctx.trace_closest(rays)
    want = O.trace_closest(sc, rays)
    for f in ("t", "tri_id", "inst_id", "u", "v"):
        assert np.array_equal(hits[f].view(np.uint32), want[f].view(np.uint32)), f
    _, edges = SS.load_fixture()
    precision, recall, within2 = SS.silhouette_agreement(SS.boundary_pixels(SS.surface_map(sc, hits)), edges)
    assert precision >= 0.995 and within2 >= 0.998 and recall >= 0.78, (precision, recall, within2)


@pytest.mark.parametrize("rng_mode", [T.RNG_WANG, T.RNG_PCG])
def test_hip_render_equals_the_screenshot(cornell, rng_mode):
    ctx, sc, camu = cornell
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)
    ctx.clear_accum()
    ctx.render(O.render_params(SS.W, SS.H, 0, SS.SPP_SQRT, SS.DEPTH, rng_mode=rng_mode, flags=T.RENDER_SKY_WHITE_HORIZON))
    img = ctx.read_accum(SS.W, SS.H)
    assert np.isfinite(img).all()
    shown = SS.display_transform(img[..., :3])
    want, edges = SS.load_fixture()
    got = SS.box4(shown)
    err = (got - want)[1:]  # (first row of blocks: the screenshot's artefact rows)
    bias = err.reshape(-1, 3).mean(0)
    psnr = SS.psnr(got[1:], want[1:])
    print(f"\nscreenshot pin, rng {rng_mode}: PSNR {psnr:.2f} dB over 4x4 boxes, bias {bias}, max |err| {np.abs(err).max():.2f}")
    # measured on MI355X: 52.2 dB, bias below 0.014 levels, largest box error 7 levels (edge pixels)
    assert np.abs(bias).max() < 0.1, bias
    assert psnr >= 50.0, psnr
    # per flat surface (walls, floor, box faces, sky): mean colour within one 8-bit level
    key = SS.surface_map(sc, O.trace_closest(sc, SS.pixel_centre_rays(camu)))
    means = SS.surface_means(key, got, want)
    assert len(means) == 11
    worst = max(float(np.abs(a - b).max()) for _, a, b in means.values())
    print("per-surface mean colour, worst channel difference:", worst)
    assert worst < 0.5, {k: (n, a.round(2), b.round(2)) for k, (n, a, b) in means.items()}
    # the render's own gradient edges (same detector as the fixture's) against the screenshot's
    mine = SS.gradient_edges(shown.astype(np.float32))
    precision, recall, within2 = SS.silhouette_agreement(mine, edges)
    print("edge map of the render vs the screenshot's:", precision, recall, within2)
    assert precision >= 0.95 and recall >= 0.97, (precision, recall, within2)  # measured 0.973 / 0.992
