#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the oracle (run from the repo root:
`python -m tests.golden.make_golden`).  Inputs: assets/teapot.obj (data copied
from the reference's Scenes/), the reference's procedural plane/cube and its
hard-coded Cornell layout (src/PathTracer.cpp:93-220).  The reference itself
holds no golden vectors and cannot be run, so these pin the
ORACLE's behaviour (and, through the -m gpu tests, the HIP path's) against
regressions; the oracle in turn is pinned by tests/test_oracle_kat.py."""
import ctypes as C
import math
import os

import numpy as np

from raytracing_amd import _types as T
from tests import _golden as G
from tests import _oracle as O
from tests import _scenes as S

F = np.float32


def structured(a):
    """npz cannot hold structured dtypes without pickling: store raw bytes."""
    return np.ascontiguousarray(a).view(np.uint8) if a.dtype.fields else a


def save(name, **arrays):
    out = {k: structured(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(G.GOLDEN_DIR, name), **out)
    print(name, {k: v.shape for k, v in out.items() if not k.startswith("scene_")})


def cam_bytes(cam):
    return np.frombuffer(bytes(cam), np.uint8).copy()


def instanced_scene():
    """teapots + cubes + planes, all four material types, a texture (no sphere:
    its vertices depend on the machine's sin/cos)."""
    rng = np.random.default_rng(42)
    r = O.OracleRenderer()
    plane = r.add_blas(*S.plane_mesh())
    cube = r.add_blas(*S.cube_mesh())
    teapot = r.add_blas(*S.load_obj(os.path.join(S.ASSETS, "teapot.obj")))
    checker = np.zeros((4, 4, 4), np.uint8)
    checker[..., 3] = 255
    checker[::2, ::2, :3] = 230; checker[1::2, 1::2, :3] = 230
    checker[::2, 1::2, 0] = 200; checker[1::2, ::2, 2] = 200
    mats = [r.default_material, r.add_lambert_material((0.8, 0.3, 0.2)), r.add_metal_material((0.8, 0.8, 0.9), 0.05),
            r.add_metal_material((0.9, 0.6, 0.2), 0.4), r.add_dielectric_material(1.5),
            r.add_emissive_material((4.0, 3.5, 3.0)), r.add_lambert_texture(4, 4, checker)]
    r.add_blas_instance(plane, O.trs((0, -0.5, 0), (1, 0, 0, 0), (12, 1, 12)), mats[6])
    k = 0
    for iz in range(3):
        for ix in range(4):
            p = ((ix - 1.5) * 1.5 + rng.uniform(-0.2, 0.2), rng.uniform(0.0, 0.4), (iz - 1) * 1.6 + rng.uniform(-0.2, 0.2))
            axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
            q = O.angle_axis(rng.uniform(0, 2 * math.pi), axis)
            s = rng.uniform(0.6, 1.1)
            r.add_blas_instance([cube, teapot][k % 2], O.trs(p, q, (s, s, s)), mats[k % 6])
            k += 1
    return r


def cornell_no_sphere():
    r = O.OracleRenderer()
    plane, cube = r.add_blas(*S.plane_mesh()), r.add_blas(*S.cube_mesh())
    red, white = r.add_lambert_material((0.65, 0.05, 0.05)), r.add_lambert_material((0.73, 0.73, 0.73))
    green, light = r.add_lambert_material((0.12, 0.45, 0.15)), r.add_emissive_material((15.0, 15.0, 15.0))
    pi, hp = F(math.pi), F(math.pi) / F(2)
    aa, I = O.angle_axis, (1, 0, 0, 0)
    rad = lambda d: F(d) * F(0.01745329251994329576923690768489)
    for b, t, m in [(plane, O.trs((0, 0, -0.025), I, (2, 1, 2)), white),
                    (plane, O.trs((0, 2.0, -0.025), aa(pi, (1, 0, 0)), (2, 1, 2)), white),
                    (plane, O.trs((0, 1, -1.025), aa(-hp, (1, 0, 0)), (2, 1, 2)), white),
                    (plane, O.trs((-1, 1, -0.025), aa(hp, (0, 0, 1)), (2, 1, 2)), red),
                    (plane, O.trs((1, 1, -0.025), aa(-hp, (0, 0, 1)), (2, 1, 2)), green),
                    (plane, O.trs((0, 1.99, -0.03), aa(pi, (1, 0, 0)), (0.5, 1, 0.4)), light),
                    (cube, O.trs((0.3, 0.3, 0.35), aa(rad(-18.0), (0, 1, 0)), (0.6, 0.6, 0.6)), white),
                    (cube, O.trs((-0.4, 0.6, -0.3), aa(rad(15.0), (0, 1, 0)), (0.6, 1.2, 0.6)), white)]:
        r.add_blas_instance(b, t, m)
    return r


def main():
    os.makedirs(G.GOLDEN_DIR, exist_ok=True)
    # 1. hit records on fixed ray sets
    for name, make, center, radius in [("teapot", S.teapot_scene, (0, 0.15, 0), 2.0),
                                       ("instanced", instanced_scene, (0, 0.3, 0), 4.0),
                                       ("cornell", cornell_no_sphere, (0, 1, 0), 1.5)]:
        sc = make().build()
        rays = S.random_rays(4000, seed=7, center=center, radius=radius)
        save(f"hits_{name}.npz", rays=rays, hits_div=O.trace_closest(sc, rays, T.AABB_DIV),
             hits_rcp=O.trace_closest(sc, rays, T.AABB_RCP), **G.scene_to_dict(sc))
    # 2. images: the reference's launch shape (1 spp, depth 3) on its own scene, and a deeper one
    sc = cornell_no_sphere().build()
    w, h = 64, 36
    cam = S.cornell_camera(w, h)
    img = np.zeros((h, w, 4), F)
    for frame in range(3):
        O.render(sc, cam, O.render_params(w, h, frame, 1, 3), img)
    save("image_cornell_ref_shape.npz", image=img, camera=cam_bytes(cam), params=np.array([w, h, 3, 1, 3, T.RNG_WANG], np.uint32),
         **G.scene_to_dict(sc))
    sc = instanced_scene().build()
    w, h = 72, 40
    cam = S.mixed_camera(w, h)
    for rng_mode, tag in [(T.RNG_WANG, "wang"), (T.RNG_PCG, "pcg")]:
        img = np.zeros((h, w, 4), F)
        for frame in range(2):
            O.render(sc, cam, O.render_params(w, h, frame, 2, 8, rng_mode=rng_mode), img)
        save(f"image_instanced_{tag}.npz", image=img, camera=cam_bytes(cam),
             params=np.array([w, h, 2, 2, 8, rng_mode], np.uint32), **G.scene_to_dict(sc))
    # 3. Wang-hash / rand sequence (Random.slang:7-19) for the first pixels of frame 0
    L = O.lib()
    seq = []
    for (x, y, f) in [(0, 0, 0), (1, 0, 0), (0, 1, 0), (7, 3, 5), (1919, 1079, 255)]:
        s = C.c_uint32((x * 1973) ^ (y * 9277) ^ (f * 26699))
        seq.append([L.orc_rand(C.byref(s)) for _ in range(8)])
    save("wang_rand.npz", seeds=np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (7, 3, 5), (1919, 1079, 255)], np.uint32),
         rand=np.array(seq, F))


if __name__ == "__main__":
    main()
