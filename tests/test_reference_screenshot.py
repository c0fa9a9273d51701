"""The oracle against the ONE output of this path that the reference itself holds: Screenshot3.png, its render of
Scenes/TestScene.json (= assets/scenes/cornell.json: CornellBox-Original.obj at scale 3, camera (0, 6, 15.5) ->
(0, 2, 0), vfov 20, 1000 spp, depth 50).  tests/golden/screenshot3.npz holds the box-filtered image and its edge map
(tests/golden/make_screenshot_fixture.py); tests/_screenshot.py says what the screenshot is and how it is compared.

What this pins: the scene-file and OBJ/MTL loader (Kd per wall), the scale-3 transform, the camera math
(Renderer.cpp:506-537) and pixel mapping — silhouettes within one pixel at 1920x1080 — and, radiometrically, the sky
gradient, the Lambert scatter distribution and albedo, and the multi-bounce transport down to depth 50: absolute
8-bit levels, no fitted tone curve.  What it CANNOT pin: hit ids, the RNG sequence, traversal order, the other three
material types, instancing with rotations — those rest on the oracle's line-by-line restatement alone.

The GPU half (the HIP render at the screenshot's own 1920x1080 / 961 spp / depth 50) is tests/test_gpu_screenshot.py.
"""
import os

import numpy as np
import pytest

from raytracing_amd import _types as T
from raytracing_amd import renderer as R
from . import _oracle as O
from . import _scenes as S
from . import _screenshot as SS

SCENE = os.path.join(S.ASSETS, "scenes", "cornell.json")


@pytest.fixture(scope="module")
def cornell():
    r = R.Renderer(SS.W, SS.H, device=None)
    cam, info = r.load_scene_file(SCENE)
    assert (info["screen_width"], info["screen_height"]) == (SS.W, SS.H)
    return r, cam, r.scene_arrays()


def test_fixture_is_the_screenshot_the_script_describes():
    mean4, edges = SS.load_fixture()
    assert mean4.shape == (270, 480, 3) and edges.shape == (SS.H, SS.W)
    assert 5000 < edges.sum() < 20000
    # its sky: the numbers the white-horizon derivation in DESIGN.md §2 rests on
    assert np.allclose(mean4[1, 2], [224.2, 237.2, 255], atol=0.6) and np.allclose(mean4[268, 2], [235, 243.7, 255], atol=0.6)


def test_oracle_silhouettes_lie_on_the_screenshots_edges(cornell):
    """Primary hits of the oracle through every pixel centre at the screenshot's resolution: the boundaries of the
    surfaces they see against the screenshot's edges."""
    r, cam, sc = cornell
    hits = O.trace_closest(sc, SS.pixel_centre_rays(SS.screenshot_camera(r.camera_uniforms(cam))))
    key = SS.surface_map(sc, hits)
    assert len(np.unique(key)) == 11  # sky, floor, back wall, two side walls, 3 + 3 visible box faces (no ceiling or light in view)
    _, edges = SS.load_fixture()
    precision, recall, within2 = SS.silhouette_agreement(SS.boundary_pixels(key), edges)
    # measured: 0.9989 / 0.819 / 0.9992 — every edge the screenshot shows is a boundary of ours, to the pixel
    assert precision >= 0.995 and within2 >= 0.998, (precision, recall, within2)
    assert recall >= 0.78, recall


def test_oracle_radiometry_equals_the_screenshot(cornell):
    """The oracle's render at a quarter of the screenshot's resolution (480x270, 144 spp, depth 50) through the old PNG
    transform against the box-filtered screenshot, in absolute 8-bit levels."""
    r, cam, sc = cornell
    w, h = SS.W // 4, SS.H // 4
    small = R.Renderer(w, h, device=None)
    camu = SS.screenshot_camera(small.camera_uniforms(cam))  # (half a pixel of the 1920x1080 grid = 1/8 of this one:
    for k in range(3):                                          #  move back by 3/8)
        camu.pixel00_loc[k] -= np.float32(0.375) * (np.float32(camu.pixel_delta_u[k]) + np.float32(camu.pixel_delta_v[k]))
    p = O.render_params(w, h, 0, 12, SS.DEPTH, flags=T.RENDER_SKY_WHITE_HORIZON)
    img = O.render(sc, camu, p, nthreads=os.cpu_count() or 8)
    shown = SS.display_transform(img[..., :3])
    want, _ = SS.load_fixture()
    err = (shown - want)[1:]  # (first row of blocks: the screenshot's artefact rows)
    bias = err.reshape(-1, 3).mean(0)
    blocks = err[: 260].reshape(26, 10, 48, 10, 3).mean(axis=(1, 3))  # 40x40 screenshot pixels each: noise averaged out
    block_psnr = 20 * np.log10(255.0 / np.sqrt((blocks ** 2).mean()))
    # measured: bias (-0.06, -0.05, -0.04) levels, blocks 50.8 dB, pixels 34.6 dB (the noise of 144 spp against 961)
    assert np.abs(bias).max() < 0.4, bias
    assert block_psnr >= 47.0, block_psnr
    assert SS.psnr(shown[1:], want[1:]) >= 33.0
    # teeth: the sky of the shader in the tree (horizon 0.7) is NOT what the screenshot shows
    p0 = O.render_params(w, h, 0, 4, SS.DEPTH)
    shown0 = SS.display_transform(O.render(sc, camu, p0, nthreads=os.cpu_count() or 8)[..., :3])
    blocks0 = (shown0 - want)[1:261].reshape(26, 10, 48, 10, 3).mean(axis=(1, 3))
    assert 20 * np.log10(255.0 / np.sqrt((blocks0 ** 2).mean())) < 30.0
