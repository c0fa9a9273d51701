"""Comparison of a render of assets/scenes/cornell.json with the reference's own render of that scene
(tests/golden/screenshot3.npz, derived from the reference's Screenshot3.png by tests/golden/make_screenshot_fixture.py).

What the screenshot is (derived from its pixels): the output of an OLDER revision of the reference
than the snapshot's shader — same scene file, camera, OBJ loader, Lambert scatter and depth-50 / 1000-spp settings,
but (a) its sky blends float3(1.0) -> (0.5, 0.7, 1.0) where RayTracing.slang:162 now has float3(0.7) (selected here
by RT_RENDER_SKY_WHITE_HORIZON), and (b) it was written by the old PNG writer, 256 * clamp(sqrt(c), 0, 0.999)
truncated to 8 bits (the gamma-2 transform of the CPU renderer the README still documents), not by the ACES-style
curve of FullScreen.slang, and (c) its pixel (x, y) covers [x, x+1) x [y, y+1) of the viewport where the snapshot's
jitter `(s + rand) / sqrt_spp - 0.5` (RayTracing.slang:47-57) covers [x-0.5, x+0.5): measured on five strong edges of
the HIP render at 1920x1080, the screenshot's edges sit 0.43 ... 0.53 pixels to the upper left of ours
(`screenshot_camera`).  With (a), (b) and (c) a render can be compared with it in ABSOLUTE terms, no fitted curve.
"""
import os

import numpy as np
from scipy import ndimage as ndi

from raytracing_amd import _types as T
from .golden.make_screenshot_fixture import gradient_edges  # noqa: F401  (same detector for renders)

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "screenshot3.npz")
W, H = 1920, 1080
SPP_SQRT, DEPTH = 31, 50  # TestScene.json: 1000 spp -> u32(sqrt(1000)) = 31 (Renderer.cpp:625), max_depth 50
TOP_ROWS_IGNORED = 4      # the screenshot's first rows hold a dashed artefact of its renderer (not geometry)


def load_fixture():
    z = np.load(FIXTURE, allow_pickle=False)
    h, w = [int(v) for v in z["shape"]]
    assert (h, w) == (H, W)
    edges = np.unpackbits(z["edges"])[: h * w].reshape(h, w).astype(bool)
    mean4 = z["sum4x4"].astype(np.float32) / np.float32(16)  # box-filtered 8-bit levels, [270, 480, 3]
    return mean4, edges


def screenshot_camera(camu):
    """(c) above: the same camera with the pixel grid moved by half a pixel, pixel00 + (du + dv) / 2."""
    import copy
    c = copy.deepcopy(camu)
    for k in range(3):
        c.pixel00_loc[k] = np.float32(np.float32(c.pixel00_loc[k]) +
                                      np.float32(0.5) * (np.float32(c.pixel_delta_u[k]) + np.float32(c.pixel_delta_v[k])))
    return c


def display_transform(linear_rgb):
    """The old renderer's PNG write: int(256 * clamp(sqrt(c), 0, 0.999))."""
    c = np.sqrt(np.maximum(linear_rgb.astype(np.float32), np.float32(0)))
    return np.floor(np.float32(256) * np.clip(c, np.float32(0), np.float32(0.999)))


def box4(img):
    h, w = img.shape[:2]
    return img.reshape(h // 4, 4, w // 4, 4, -1).mean(axis=(1, 3))


def psnr(a, b):
    mse = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean())
    return float("inf") if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse))


def pixel_centre_rays(camu, w=W, h=H):
    """Camera rays through the pixel centres (RayTracing.slang:81-87 with zero jitter), fp32."""
    F = np.float32
    p00, du, dv, cc = (np.array(v[:3], F) for v in (camu.pixel00_loc, camu.pixel_delta_u, camu.pixel_delta_v,
                                                     camu.camera_center))
    xs, ys = np.arange(w, dtype=F), np.arange(h, dtype=F)
    p = p00[None, None, :] + xs[None, :, None] * du[None, None, :] + ys[:, None, None] * dv[None, None, :]
    d = p - cc
    d = d / np.sqrt((d * d).sum(-1, keepdims=True))
    rays = np.zeros(w * h, T.ray_dtype)
    rays["origin"] = cc
    rays["direction"] = d.reshape(-1, 3)
    rays["t_min"], rays["t_max"] = 1e-4, 1000.0
    return rays


def surface_map(scene, hits, w=W, h=H):
    """Per pixel an id of the flat surface the primary ray hit — (instance, face normal) — or -1 for the sky.  The
    two triangles of a quad share an id, so the map's boundaries are the silhouettes and creases of the scene."""
    g = scene.tri_geom
    v0 = g["v0"][:, :3].astype(np.float64)
    n = np.cross(g["v1"][:, :3] - v0, g["v2"][:, :3] - v0)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.where(ln > 0, ln, 1), 0)
    _, face = np.unique(np.round(n * 8).astype(np.int64), axis=0, return_inverse=True)
    hit = hits["tri_id"] != 0xFFFFFFFF
    tri = np.where(hit, hits["tri_id"], 0).astype(np.int64)
    key = np.where(hit, hits["inst_id"].astype(np.int64) * 4096 + face.reshape(-1)[tri], -1)
    return key.reshape(h, w)


def boundary_pixels(key):
    """Pixels on either side of a boundary of the surface map (4-neighbourhood)."""
    e = np.zeros(key.shape, bool)
    d = key[:, 1:] != key[:, :-1]
    e[:, 1:] |= d
    e[:, :-1] |= d
    d = key[1:, :] != key[:-1, :]
    e[1:, :] |= d
    e[:-1, :] |= d
    return e


def silhouette_agreement(geom_edges, shot_edges):
    """(precision, recall, within2): the share of the screenshot's edge pixels (below its artefact rows) that lie within
    1 pixel of a boundary of the surface map; the share of boundary pixels that have a screenshot edge within 1 pixel
    (boundaries between equally lit faces leave no edge in any render, so this one stays below 1); and the share of
    the screenshot's edge pixels within 2 pixels of a boundary (what is left are specks of its noise)."""
    s = shot_edges.copy()
    s[:TOP_ROWS_IGNORED] = False
    to_geom = ndi.distance_transform_edt(~geom_edges)
    to_shot = ndi.distance_transform_edt(~s)
    g = geom_edges.copy()
    g[:TOP_ROWS_IGNORED] = False
    return float((to_geom[s] <= 1.0).mean()), float((to_shot[g] <= 1.0).mean()), float((to_geom[s] <= 2.0).mean())


def surface_means(key, shown, shot_mean4):
    """Mean colour per surface over the 4x4 blocks that lie wholly inside it, at least 2 blocks from its boundary:
    {surface id: (blocks, mean of `shown` [levels, rgb], mean of the screenshot [levels, rgb])}."""
    kb = key.reshape(key.shape[0] // 4, 4, key.shape[1] // 4, 4)
    uniform = (kb == kb[:, :1, :, :1]).all(axis=(1, 3))
    kid = np.where(uniform, kb[:, 0, :, 0], -2)
    out = {}
    for k in np.unique(kid):
        if k == -2:
            continue
        m = ndi.binary_erosion(kid == k, iterations=2)
        m[:1] = False  # (the artefact rows)
        if m.sum() >= 50:
            out[int(k)] = (int(m.sum()), shown[m].mean(0), shot_mean4[m].mean(0))
    return out
