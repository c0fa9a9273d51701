"""Turns the reference's own render, /root/reference/Screenshot3.png, into tests/golden/screenshot3.npz.

Screenshot3.png (1920x1080, README.md:4 of the reference) is the reference's render of Scenes/TestScene.json:1-20 —
CornellBox-Original.obj at scale 3, camera (0, 6, 15.5) -> (0, 2, 0), vfov 20, 1000 spp, depth 50 — i.e. of this
repository's assets/scenes/cornell.json.  It is the only output of the path that the reference holds, so it is what
the oracle can be pinned against (tests/test_reference_screenshot.py).  The PNG is DATA; this script keeps two
compact derivatives of it, not the file:

  sum4x4  uint16 [270, 480, 3]   sums of the 8-bit values over 4x4 pixel blocks (the exact box-filtered image)
  edges   packed bits [1080*1920] non-maximum-suppressed gradient edges at full resolution (parameters below)

Run here (the reference tree does not exist on the GPU box):  python tests/golden/make_screenshot_fixture.py
"""
import hashlib
import os

import numpy as np
from scipy import ndimage as ndi

SRC = "/root/reference/Screenshot3.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "screenshot3.npz")
SIGMA, THRESHOLD = 1.0, 4.0  # Gaussian-derivative scale (pixels) and gradient threshold (8-bit levels per pixel)


def gradient_edges(img, sigma=SIGMA, threshold=THRESHOLD):
    """Canny-style edges of an [H, W, 3] float image: per pixel the channel with the largest Gaussian-derivative
    gradient, kept where it exceeds `threshold` and is a maximum along its dominant axis."""
    h, w = img.shape[:2]
    gm = np.zeros((h, w), np.float32)
    gx_best = np.zeros_like(gm)
    gy_best = np.zeros_like(gm)
    for c in range(3):
        gx = ndi.gaussian_filter(img[..., c], sigma, order=(0, 1))
        gy = ndi.gaussian_filter(img[..., c], sigma, order=(1, 0))
        m = np.hypot(gx, gy)
        sel = m > gm
        gm = np.where(sel, m, gm)
        gx_best = np.where(sel, gx, gx_best)
        gy_best = np.where(sel, gy, gy_best)
    p = np.pad(gm, 1, mode="edge")
    left, right, up, down = p[1:-1, :-2], p[1:-1, 2:], p[:-2, 1:-1], p[2:, 1:-1]
    along_x = np.abs(gx_best) >= np.abs(gy_best)
    is_max = np.where(along_x, (gm >= left) & (gm >= right), (gm >= up) & (gm >= down))
    return (gm > threshold) & is_max


def main():
    from PIL import Image
    raw = open(SRC, "rb").read()
    img = np.asarray(Image.open(SRC).convert("RGB"))
    assert img.shape == (1080, 1920, 3) and img.dtype == np.uint8
    sum4 = img.reshape(270, 4, 480, 4, 3).astype(np.uint16).sum(axis=(1, 3)).astype(np.uint16)
    edges = gradient_edges(img.astype(np.float32))
    np.savez_compressed(OUT, sum4x4=sum4, edges=np.packbits(edges.reshape(-1)), shape=np.array(img.shape[:2]),
                        sigma=np.float32(SIGMA), threshold=np.float32(THRESHOLD),
                        source=np.array("reference: Screenshot3.png (README.md:4), sha256 " + hashlib.sha256(raw).hexdigest()))
    print(OUT, os.path.getsize(OUT), "bytes;", int(edges.sum()), "edge pixels")


if __name__ == "__main__":
    main()
