"""Scene inputs for the tests, assembled through the oracle (tests/_oracle.py).

Meshes restate the reference's procedural primitives (src/Renderer.cpp:64-213)
and its hard-coded Cornell box (src/PathTracer.cpp:93-220, 390-395) in numpy
fp32; OBJ files come from assets/ (copies of the reference's Scenes/ data).
"""
import math
import os

import numpy as np

from raytracing_amd import _types as T
from . import _oracle as O

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
F = np.float32


def plane_mesh(width=1.0, depth=1.0, xs=1, zs=1):  # Renderer.cpp:109-145
    pos, nrm, uv, idx = [], [], [], []
    for z in range(zs + 1):
        vz = F(z) / F(zs)
        pz = (vz - F(0.5)) * F(depth)
        for x in range(xs + 1):
            ux = F(x) / F(xs)
            px = (ux - F(0.5)) * F(width)
            pos.append((px, 0.0, pz)); nrm.append((0, 1, 0)); uv.append((ux, vz))
    for z in range(zs):
        for x in range(xs):
            i0 = z * (xs + 1) + x
            i1 = i0 + xs + 1
            idx += [i0, i1, i0 + 1, i0 + 1, i1, i1 + 1]
    return np.array(pos, F), np.array(nrm, F), np.array(uv, F), np.array(idx, np.uint32)


def cube_mesh(w=1.0, h=1.0, d=1.0):  # Renderer.cpp:147-213
    hx, hy, hz = F(w) * F(0.5), F(h) * F(0.5), F(d) * F(0.5)
    pos, nrm, uv, idx = [], [], [], []

    def face(v0, v1, v2, v3, n):
        b = len(pos)
        pos.extend([v0, v1, v2, v3]); nrm.extend([n] * 4)
        uv.extend([(0, 0), (1, 0), (1, 1), (0, 1)])
        idx.extend([b, b + 1, b + 2, b, b + 2, b + 3])

    face((-hx, -hy, hz), (hx, -hy, hz), (hx, hy, hz), (-hx, hy, hz), (0, 0, 1))
    face((hx, -hy, -hz), (-hx, -hy, -hz), (-hx, hy, -hz), (hx, hy, -hz), (0, 0, -1))
    face((-hx, -hy, -hz), (-hx, -hy, hz), (-hx, hy, hz), (-hx, hy, -hz), (-1, 0, 0))
    face((hx, -hy, hz), (hx, -hy, -hz), (hx, hy, -hz), (hx, hy, hz), (1, 0, 0))
    face((-hx, hy, hz), (hx, hy, hz), (hx, hy, -hz), (-hx, hy, -hz), (0, 1, 0))
    face((-hx, -hy, -hz), (hx, -hy, -hz), (hx, -hy, hz), (-hx, -hy, hz), (0, -1, 0))
    return np.array(pos, F), np.array(nrm, F), np.array(uv, F), np.array(idx, np.uint32)


def sphere_mesh(radius=0.5, segments=64, rings=32):  # Renderer.cpp:64-107
    ys, xs = np.arange(rings + 1, dtype=F), np.arange(segments + 1, dtype=F)
    v = ys / F(rings)
    u = xs / F(segments)
    theta = (v * F(math.pi)).astype(F)
    phi = (u * F(2 * math.pi)).astype(F)
    st, ct = np.sin(theta).astype(F), np.cos(theta).astype(F)
    sp, cp = np.sin(phi).astype(F), np.cos(phi).astype(F)
    nrm = np.stack([np.outer(st, cp), np.repeat(ct[:, None], segments + 1, 1), np.outer(st, sp)], -1).astype(F)
    pos = (F(radius) * nrm).astype(F)
    uv = np.stack([np.repeat(u[None, :], rings + 1, 0), np.repeat(v[:, None], segments + 1, 1)], -1).astype(F)
    idx = []
    for y in range(rings):
        for x in range(segments):
            i0 = y * (segments + 1) + x
            i1 = i0 + segments + 1
            idx += [i0, i1, i0 + 1, i0 + 1, i1, i1 + 1]
    return pos.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), np.array(idx, np.uint32)


def load_obj(path):
    """Minimal OBJ reader: v / vn / vt / f (fan triangulation like tinyobj),
    negative indices, per-corner de-indexing (no vertex welding: welding does
    not change the triangle soup handed to add_blas).  Missing normals become
    the geometric face normal (SceneGraph.cpp:338-352 convention), missing uvs
    (0,0); v is flipped to 1-v (Scene.cpp:91-93)."""
    vs, vns, vts = [], [], []
    pos, nrm, uv = [], [], []
    with open(path) as fh:
        for line in fh:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if p[0] == "v":
                vs.append([float(x) for x in p[1:4]])
            elif p[0] == "vn":
                vns.append([float(x) for x in p[1:4]])
            elif p[0] == "vt":
                vts.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
            elif p[0] == "f":
                corners = []
                for tok in p[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0]); vi = vi - 1 if vi > 0 else len(vs) + vi
                    ti = ni = None
                    if len(parts) > 1 and parts[1]:
                        ti = int(parts[1]); ti = ti - 1 if ti > 0 else len(vts) + ti
                    if len(parts) > 2 and parts[2]:
                        ni = int(parts[2]); ni = ni - 1 if ni > 0 else len(vns) + ni
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):
                    tri = (corners[0], corners[k], corners[k + 1])
                    P = [np.array(vs[c[0]], F) for c in tri]
                    fn = np.cross(P[1] - P[0], P[2] - P[0]).astype(F)
                    ln = np.sqrt(np.dot(fn, fn))
                    fn = fn / ln if ln > 0 else np.array([0, 1, 0], F)
                    for c, pp in zip(tri, P):
                        pos.append(pp)
                        nrm.append(np.array(vns[c[2]], F) if c[2] is not None else fn)
                        uv.append((F(vts[c[1]][0]), F(1.0) - F(vts[c[1]][1])) if c[1] is not None else (0.0, 0.0))
    n = len(pos)
    return np.array(pos, F), np.array(nrm, F), np.array(uv, F), np.arange(n, dtype=np.uint32)


def cornell_builtin(r=None):
    """The scene the live reference binary renders."""
    r = r or O.OracleRenderer()
    plane = r.add_blas(*plane_mesh())
    cube = r.add_blas(*cube_mesh())
    r.add_blas(*sphere_mesh())
    red = r.add_lambert_material((0.65, 0.05, 0.05))
    white = r.add_lambert_material((0.73, 0.73, 0.73))
    green = r.add_lambert_material((0.12, 0.45, 0.15))
    light = r.add_emissive_material((15.0, 15.0, 15.0))
    pi, hp = F(math.pi), F(math.pi) / F(2)
    I = (1, 0, 0, 0)
    aa = O.angle_axis
    rad = lambda deg: F(deg) * F(0.01745329251994329576923690768489)  # glm::radians
    r.add_blas_instance(plane, O.trs((0, 0, -0.025), I, (2, 1, 2)), white)
    r.add_blas_instance(plane, O.trs((0, 2.0, -0.025), aa(pi, (1, 0, 0)), (2, 1, 2)), white)
    r.add_blas_instance(plane, O.trs((0, 1, -1.025), aa(-hp, (1, 0, 0)), (2, 1, 2)), white)
    r.add_blas_instance(plane, O.trs((-1, 1, -0.025), aa(hp, (0, 0, 1)), (2, 1, 2)), red)
    r.add_blas_instance(plane, O.trs((1, 1, -0.025), aa(-hp, (0, 0, 1)), (2, 1, 2)), green)
    r.add_blas_instance(plane, O.trs((0, 1.99, -0.03), aa(pi, (1, 0, 0)), (0.5, 1, 0.4)), light)
    r.add_blas_instance(cube, O.trs((0.3, 0.3, 0.35), aa(rad(-18.0), (0, 1, 0)), (0.6, 0.6, 0.6)), white)
    r.add_blas_instance(cube, O.trs((-0.4, 0.6, -0.3), aa(rad(15.0), (0, 1, 0)), (0.6, 1.2, 0.6)), white)
    return r


def cornell_camera(width, height):  # PathTracer.cpp:390-394
    return O.camera_uniforms((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0, width, height)


def teapot_scene(material=None):
    """BASELINE config C2: teapot.obj, one BLAS + one instance, default lambert."""
    r = O.OracleRenderer()
    b = r.add_blas(*load_obj(os.path.join(ASSETS, "teapot.obj")))
    r.add_blas_instance(b, O.trs(), material or r.default_material)
    return r


def teapot_camera(width, height):
    return O.camera_uniforms((0.0, 0.6, 2.2), (0, 0.15, 0), (0, 1, 0), 35.0, width, height)


def mixed_scene(n_side=3, seed=42):
    """Small multi-instance scene: teapots, spheres and cubes on a jittered grid
    with random rotations and all four material types (config C4 in miniature)."""
    rng = np.random.default_rng(seed)
    r = O.OracleRenderer()
    plane = r.add_blas(*plane_mesh())
    cube = r.add_blas(*cube_mesh())
    sphere = r.add_blas(*sphere_mesh(0.5, 16, 8))
    teapot = r.add_blas(*load_obj(os.path.join(ASSETS, "teapot.obj")))
    checker = np.zeros((4, 4, 4), np.uint8)
    checker[..., 3] = 255
    checker[::2, ::2, :3] = 230; checker[1::2, 1::2, :3] = 230
    checker[::2, 1::2, 0] = 200; checker[1::2, ::2, 2] = 200
    mats = [r.default_material, r.add_lambert_material((0.8, 0.3, 0.2)),
            r.add_metal_material((0.8, 0.8, 0.9), 0.05), r.add_metal_material((0.9, 0.6, 0.2), 0.4),
            r.add_dielectric_material(1.5), r.add_emissive_material((4.0, 3.5, 3.0)),
            r.add_lambert_texture(4, 4, checker)]
    r.add_blas_instance(plane, O.trs((0, -0.5, 0), (1, 0, 0, 0), (12, 1, 12)), mats[6])
    blases = [cube, sphere, teapot]
    k = 0
    for iz in range(n_side):
        for ix in range(n_side):
            p = ((ix - (n_side - 1) / 2) * 1.6 + rng.uniform(-0.2, 0.2), rng.uniform(0.0, 0.4),
                 (iz - (n_side - 1) / 2) * 1.6 + rng.uniform(-0.2, 0.2))
            axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
            q = O.angle_axis(rng.uniform(0, 2 * math.pi), axis)
            s = rng.uniform(0.6, 1.1)
            r.add_blas_instance(blases[k % 3], O.trs(p, q, (s, s, s)), mats[k % 6])
            k += 1
    return r


def mixed_camera(width, height):
    return O.camera_uniforms((3.5, 2.5, 4.5), (0, 0, 0), (0, 1, 0), 45.0, width, height)


def edge_scene(kind):
    """The Cornell scene with one ingredient that puts rays or the whole scene outside the
    preconditions of the kernels' exact fast slab division (rt_device_inl.h, slab_pair_fast):
      tiny_vertex  - one box coordinate of magnitude 1e-30: the scene flag RT_SCENE_BOXES_MODERATE is off
      huge_vertex  - one box coordinate of 1e19 (> 2^59): likewise
      squashed     - an instance scaled by 1e-10 in x: object-space directions exceed 2^30
      stretched    - an instance scaled by 1e10 in x: object-space direction components below 2^-30
    Everything else renders as usual, so the literal slab path runs on real work."""
    r = cornell_builtin()
    cube = 1  # add order of cornell_builtin: plane, cube, sphere
    white = r.add_lambert_material((0.6, 0.6, 0.7))
    I = (1, 0, 0, 0)
    if kind in ("tiny_vertex", "huge_vertex"):
        pos, nrm, uv, idx = plane_mesh()
        pos = np.array(pos, np.float32).copy()
        pos.reshape(-1, 3)[0, 1] = 1e-30 if kind == "tiny_vertex" else 1e19
        b = r.add_blas(pos, nrm, uv, idx)
        r.add_blas_instance(b, O.trs((0.2, 0.9, 0.2), I, (0.5, 1, 0.5)), white)
    elif kind == "squashed":
        r.add_blas_instance(cube, O.trs((0, 1.0, 0.1), I, (1e-10, 0.8, 0.8)), white)
    elif kind == "stretched":
        r.add_blas_instance(cube, O.trs((0, 1.2, -0.2), I, (1e10, 0.3, 0.3)), white)
    else:
        raise ValueError(kind)
    return r


def random_rays(n, seed, center=(0, 0.3, 0), radius=4.0):
    """Rays from a shell towards a jittered target around `center` (unnormalised
    and axis-aligned directions included: zero components exercise the inf/NaN
    paths of the slab test)."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, T.ray_dtype)
    o = rng.normal(size=(n, 3)); o /= np.linalg.norm(o, axis=1, keepdims=True)
    o = o * radius * rng.uniform(0.2, 1.2, size=(n, 1)) + np.array(center)
    tgt = np.array(center) + rng.normal(scale=0.6, size=(n, 3))
    d = tgt - o
    d *= rng.uniform(0.5, 2.0, size=(n, 1))
    axis_aligned = rng.random(n) < 0.05
    ax = rng.integers(0, 3, size=n)
    d[axis_aligned] = 0.0
    d[axis_aligned, ax[axis_aligned]] = rng.choice([-1.0, 1.0], size=int(axis_aligned.sum()))
    rays["origin"], rays["direction"] = o.astype(F), d.astype(F)
    rays["t_min"], rays["t_max"] = F(1e-4), F(1000.0)
    return rays
