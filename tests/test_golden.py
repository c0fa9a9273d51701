"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle still reproduces them bit for bit (the reference has no golden
vectors of its own — SURVEY.md §4 — so these are regression pins of the
restatement).  GPU (-m gpu): the HIP path reproduces them through the C ABI
with no oracle in the loop."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from . import _golden as G
from . import _oracle as O

HIT_SETS = ["teapot", "instanced", "cornell"]
IMAGES = ["image_cornell_ref_shape.npz", "image_instanced_wang.npz", "image_instanced_pcg.npz"]


def _hits_equal(got, want):
    want = want.view(T.hit_dtype).reshape(-1)
    for f in ("t", "u", "v"):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), f
    assert np.array_equal(got["tri_id"], want["tri_id"]) and np.array_equal(got["inst_id"], want["inst_id"])


def _image_case(name):
    z = G.load(name)
    sc = G.scene_from_npz(z)
    cam = T.CameraUniforms.from_buffer_copy(z["camera"].tobytes())
    w, h, frames, sqrt_spp, depth, rng = [int(v) for v in z["params"]]
    return sc, cam, (w, h, frames, sqrt_spp, depth, rng), z["image"]


@pytest.mark.parametrize("name", HIT_SETS)
def test_oracle_reproduces_golden_hits(name):
    z = G.load(f"hits_{name}.npz")
    sc = G.scene_from_npz(z)
    rays = z["rays"].view(T.ray_dtype).reshape(-1)
    _hits_equal(O.trace_closest(sc, rays, T.AABB_DIV), z["hits_div"])
    _hits_equal(O.trace_closest(sc, rays, T.AABB_RCP), z["hits_rcp"])
    assert (z["hits_div"].view(T.hit_dtype)["tri_id"] != T.INVALID_ID).sum() > 500


@pytest.mark.parametrize("name", IMAGES)
def test_oracle_reproduces_golden_images(name):
    sc, cam, (w, h, frames, sqrt_spp, depth, rng), want = _image_case(name)
    img = np.zeros((h, w, 4), np.float32)
    for f in range(frames):
        O.render(sc, cam, O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng), img, nthreads=4)
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))


def test_wang_rand_golden():
    z = G.load("wang_rand.npz")
    L = O.lib()
    for (x, y, f), want in zip(z["seeds"], z["rand"]):
        s = C.c_uint32((int(x) * 1973) ^ (int(y) * 9277) ^ (int(f) * 26699))
        got = [L.orc_rand(C.byref(s)) for _ in range(len(want))]
        assert np.array_equal(np.array(got, np.float32), want)


@pytest.fixture(scope="module")
def ctx():
    from raytracing_amd import _ffi
    c = _ffi.HipContext(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", HIT_SETS)
def test_hip_reproduces_golden_hits(ctx, name):
    z = G.load(f"hits_{name}.npz")
    ctx.upload_scene(G.scene_from_npz(z))
    rays = z["rays"].view(T.ray_dtype).reshape(-1)
    _hits_equal(ctx.trace_closest(rays, T.AABB_DIV), z["hits_div"])
    _hits_equal(ctx.trace_closest(rays, T.AABB_RCP), z["hits_rcp"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 8])
@pytest.mark.parametrize("name", IMAGES)
def test_hip_reproduces_golden_images(ctx, name, variant):
    sc, cam, (w, h, frames, sqrt_spp, depth, rng), want = _image_case(name)
    ctx.upload_scene(sc)
    ctx.set_camera(cam)
    ctx.resize(w, h)
    ctx.set_kernel_variant(variant)
    for f in range(frames):
        ctx.render(O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng))
    got = ctx.read_accum(w, h)
    ctx.set_kernel_variant(_ffi.KERNEL_VARIANT_AUTO)  # back to the default
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
