"""SURVEY.md Appendix C, item by item: a SECOND restatement of the reference's builder and traversal, written in
numpy / plain Python from the reference sources and sharing no code with oracle/rt_oracle.cpp, run against the
oracle on small inputs.  It cannot pin the oracle to the reference binary (nothing can here: parity unpinned), but
it catches transcription slips in either restatement, and it is where each numbered quirk has a named check
(DESIGN.md §2 holds the item -> test map).

fp32 conventions are the oracle's (bit-exactness contract): every operation rounded to fp32, no
contraction, dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z, mat*vec = (c0*x + c1*y) + (c2*z + c3*w), min/max as
compare-select in the reference's operand order.
"""
import sys

import numpy as np
import pytest

from raytracing_amd import _types as T
from . import _oracle as O
from . import _scenes as S

F = np.float32
INF = F(np.inf)
BIN_COUNT = 200          # BVHNode.cpp:5
MISS = F(1e30)           # BVHNode.slang:25
EPSILON = F(1.192092896e-07)  # Ray.slang:4


# ---------------------------------------------------------------- BLAS::build (BVHNode.cpp:14-161), items 1-5

def half_area(bmin, bmax):  # AABB.hpp:21-24
    e = bmax - bmin
    return F(F(e[0] * e[1]) + F(e[1] * e[2])) + F(e[0] * e[2])


def py_build_blas(v0, v1, v2, tri_id_offset=0):
    """Returns (nodes as list of [aabb_min, aabb_max, local_left_first, tri_count], tri_ids)."""
    n = len(v0)
    cent = ((v0 + v1) + v2) * F(0.3333)                      # Renderer.cpp:780-783 (item 1)
    tri_ids = np.arange(n, dtype=np.int64)                   # BLAS-local here; the offset is added when leaves are written
    nodes = []

    def bounds(first, count):                                # BVHNode.cpp:89-105
        ids = tri_ids[first:first + count]
        pts = np.concatenate([v0[ids], v1[ids], v2[ids]])
        return pts.min(0), pts.max(0)

    def best_plane(first, count):                            # BVHNode.cpp:14-74
        best_cost, best_axis, best_pos = INF, -1, F(0)
        ids = tri_ids[first:first + count]
        for a in range(3):
            c = cent[ids, a]
            lo, hi = c.min(), c.max()
            if lo == hi:
                continue
            scale = F(BIN_COUNT) / F(hi - lo)
            b = np.minimum(BIN_COUNT - 1, ((c - lo) * scale).astype(np.uint32))   # truncation, clamp to 199
            # bins are fresh for every axis (item 2): counts and boxes start empty here
            cnt = np.bincount(b, minlength=BIN_COUNT)
            bmin = np.full((BIN_COUNT, 3), INF, F)
            bmax = np.full((BIN_COUNT, 3), -INF, F)
            for vv in (v0, v1, v2):                          # bin boxes grow by VERTICES, binning is by CENTROID
                np.minimum.at(bmin, b, vv[ids])
                np.maximum.at(bmax, b, vv[ids])
            lmin, lmax = np.minimum.accumulate(bmin, 0), np.maximum.accumulate(bmax, 0)
            rmin, rmax = np.minimum.accumulate(bmin[::-1], 0)[::-1], np.maximum.accumulate(bmax[::-1], 0)[::-1]
            lcnt, rcnt = np.cumsum(cnt), np.cumsum(cnt[::-1])[::-1]
            step = F(hi - lo) / F(BIN_COUNT)
            for i in range(BIN_COUNT - 1):                   # plane i separates bins 0..i from i+1..199
                cost = F(F(lcnt[i]) * half_area(lmin[i], lmax[i])) + F(F(rcnt[i + 1]) * half_area(rmin[i + 1], rmax[i + 1]))
                if cost < best_cost:                         # strict: the earliest axis / plane keeps a tie
                    best_cost, best_axis, best_pos = cost, a, F(lo + F(step * F(i + 1)))
        return best_cost, best_axis, best_pos

    def subdivide(idx):                                      # BVHNode.cpp:107-161
        bmin, bmax, first, count = nodes[idx]
        cost, axis, pos = best_plane(first, count)
        e = bmax - bmin
        parent_area = F(F(e[0] * e[1]) + F(e[1] * e[2])) + F(e[2] * e[0])
        if cost >= F(F(count) * parent_area):                # item 3: the only leaf criterion
            return
        l, r = first, first + count - 1
        while l <= r:                                        # item 4: swap from the right
            if cent[tri_ids[l], axis] < pos:
                l += 1
            else:
                tri_ids[l], tri_ids[r] = tri_ids[r], tri_ids[l]
                r -= 1
        left = l - first
        if left == 0 or left == count:
            return
        li = len(nodes)
        nodes.append([None, None, first, left])
        nodes.append([None, None, l, count - left])
        nodes[idx][2], nodes[idx][3] = li, 0                 # item 5: interior left_first is BLAS-local
        for k in (li, li + 1):
            nodes[k][0], nodes[k][1] = bounds(nodes[k][2], nodes[k][3])
        subdivide(li)
        subdivide(li + 1)

    with np.errstate(invalid="ignore", over="ignore"):
        nodes.append([*bounds(0, n), 0, n])
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(10000)
        try:
            subdivide(0)
        finally:
            sys.setrecursionlimit(old)
    return nodes, tri_ids + tri_id_offset, cent


def _soup(n, seed, spread=(4.0, 2.0, 3.0), size=0.3, snap=None):
    rng = np.random.default_rng(seed)
    c = (rng.uniform(-1, 1, (n, 3)) * np.array(spread)).astype(F)
    if snap is not None:
        c[:, snap] = F(0.5)                                 # one axis with identical centroids is skipped (BVHNode.cpp:28-29)
    tri = (c[:, None, :] + rng.uniform(-size, size, (n, 3, 3)).astype(F)).astype(F)
    if snap is not None:
        tri[:, :, snap] = F(0.5)
    pos = tri.reshape(-1, 3)
    return pos, np.tile(np.array([[0, 1, 0]], F), (3 * n, 1)), np.zeros((3 * n, 2), F), np.arange(3 * n, dtype=np.uint32)


MESHES = {
    "soup300": lambda: _soup(300, 1),
    "soup97_flat_y": lambda: _soup(97, 2, snap=1),
    "soup64_tiny": lambda: _soup(64, 3, spread=(1e-3, 1e-3, 1e-3), size=1e-4),
    "sphere": lambda: S.sphere_mesh(0.5, 16, 8),
    "cube": lambda: S.cube_mesh(),
    "plane4x4": lambda: S.plane_mesh(2.0, 2.0, 4, 4),
}


@pytest.mark.parametrize("name", sorted(MESHES))
def test_items_1_to_5_second_restatement_of_the_blas_builder(name):
    pos, nrm, uv, idx = MESHES[name]()
    r = O.OracleRenderer()
    pad = r.add_blas(*S.plane_mesh())          # a first BLAS, so that the global offsets of the second are non-zero
    b = r.add_blas(pos, nrm, uv, idx)
    rec = r.blas[b]
    tri0 = int(r.blas[pad]["tri_count"])
    pos = np.ascontiguousarray(pos, F).reshape(-1, 3)
    idx = np.asarray(idx).reshape(-1, 3)
    v0, v1, v2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    nodes, tri_ids, cent = py_build_blas(v0, v1, v2)
    want = r.nodes[rec["bvh_nodes_offset"]: rec["bvh_nodes_offset"] + rec["nodes_count"]]
    assert len(nodes) == len(want)
    assert np.array_equal(r.centroids[tri0:], cent)                                    # item 1
    assert np.array_equal(r.tri_ids[tri0:], tri_ids + tri0)                            # item 4: order inside leaves
    for k, (bmin, bmax, first, count) in enumerate(nodes):
        assert np.array_equal(want[k]["aabb_min"], bmin) and np.array_equal(want[k]["aabb_max"], bmax), k
        assert int(want[k]["tri_count"]) == count, k
        # item 5: interior -> BLAS-local child index; leaf -> GLOBAL index into tri_ids
        assert int(want[k]["local_left_first"]) == (first + tri0 if count else first), k
    assert int(rec["bvh_nodes_offset"]) == int(r.blas[pad]["nodes_count"])             # item 5: global root index


# ------------------------------------------------- traversal (BVHNode.slang, TLAS.slang, Triangle.slang), items 6-10

def fmin(a, b):  # GLSL.std.450 FMin as compare-select: min(a, b) = b < a ? b : a
    return b if b < a else a


def fmax(a, b):
    return b if a < b else a


def dot3(a, b):
    return F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2])


def cross3(a, b):
    return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], F)


def py_intersect_aabb(o, d, bmin, bmax, t):  # BVHNode.slang:8-26 — divides by the direction (item 7)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        tx1, tx2 = F(bmin[0] - o[0]) / d[0], F(bmax[0] - o[0]) / d[0]
        tmin, tmax = fmin(tx1, tx2), fmax(tx1, tx2)
        ty1, ty2 = F(bmin[1] - o[1]) / d[1], F(bmax[1] - o[1]) / d[1]
        tmin, tmax = fmax(tmin, fmin(ty1, ty2)), fmin(tmax, fmax(ty1, ty2))
        tz1, tz2 = F(bmin[2] - o[2]) / d[2], F(bmax[2] - o[2]) / d[2]
        tmin, tmax = fmax(tmin, fmin(tz1, tz2)), fmin(tmax, fmax(tz1, tz2))
    return tmin if (tmax >= tmin and tmin < t and tmax > 0) else MISS


def py_tri_hit(o, d, g, tmin, tmax):  # Triangle.slang:5-47
    v0 = g["v0"][:3]
    e1, e2 = g["v1"][:3] - v0, g["v2"][:3] - v0
    h = cross3(d, e2)
    a = dot3(e1, h)
    if -EPSILON < a < EPSILON:
        return None
    f = F(1) / a
    s = o - v0
    u = F(f * dot3(s, h))
    if u < 0 or u > 1:
        return None
    q = cross3(s, e1)
    v = F(f * dot3(d, q))
    if v < 0 or F(u + v) > 1:
        return None
    t = F(f * dot3(e2, q))
    if t < EPSILON or not (tmin <= t <= tmax):            # item 9: inclusive range and the extra epsilon reject
        return None
    return t, u, v


def py_blas_intersect(sc, blas, o, d, tmin, tmax):        # BVHNode.slang:41-109
    root = int(blas["bvh_nodes_offset"])
    stack, closest, hit = [root], tmax, None              # item 6: the root box is never tested
    visits = 0
    while True:
        node = sc.bvh_nodes[stack[-1]]
        visits += 1
        if node["tri_count"] > 0:
            for i in range(int(node["tri_count"])):
                tri = int(sc.tri_ids[int(node["local_left_first"]) + i])     # item 5: leaf index is global
                got = py_tri_hit(o, d, sc.tri_geom[tri], tmin, closest)
                if got:
                    hit, closest = (got[0], tri, got[1], got[2]), got[0]
            if len(stack) == 1:
                break
            stack.pop()
        else:
            c1 = root + int(node["local_left_first"])                       # item 5: interior index is BLAS-local
            c2 = c1 + 1
            d1 = py_intersect_aabb(o, d, sc.bvh_nodes[c1]["aabb_min"], sc.bvh_nodes[c1]["aabb_max"], closest)
            d2 = py_intersect_aabb(o, d, sc.bvh_nodes[c2]["aabb_min"], sc.bvh_nodes[c2]["aabb_max"], closest)
            if d1 > d2:                                                      # item 8: equal distances keep child1 first
                d1, d2, c1, c2 = d2, d1, c2, c1
            if d1 == MISS:                                                   # item 7: equality with the sentinel
                if len(stack) == 1:
                    break
                stack.pop()
            else:
                stack.pop()
                if d2 != MISS:
                    stack.append(c2)
                stack.append(c1)
    return hit, visits


def mat_vec(m16, v4):  # column-major mat4 * vec4 as glm computes it
    c = m16.reshape(4, 4)
    return (F(c[0] * v4[0]) + F(c[1] * v4[1])) + (F(c[2] * v4[2]) + F(c[3] * v4[3]))


def py_tlas_intersect(sc, o, d, tmin, tmax):              # TLAS.slang:14-67
    stack, closest, hit = [0], tmax, None
    t = sc.tlas_nodes
    while True:
        node = t[stack[-1]]
        if node["left_right"] == 0:
            inst = sc.instances[int(node["blas_instance_idx"])]
            oo = mat_vec(inst["inv_transform"], np.array([o[0], o[1], o[2], 1], F))[:3]
            dd = mat_vec(inst["inv_transform"], np.array([d[0], d[1], d[2], 0], F))[:3]   # item 10: not normalised
            got, _ = py_blas_intersect(sc, sc.blas[int(inst["blas_id"])], oo, dd, tmin, closest)
            if got:
                hit, closest = got + (int(node["blas_instance_idx"]),), got[0]
            if len(stack) == 1:
                break
            stack.pop()
        else:
            lr = int(node["left_right"])
            c1, c2 = lr & 0xFFFF, lr >> 16
            d1 = py_intersect_aabb(o, d, t[c1]["aabb_min"], t[c1]["aabb_max"], closest)
            d2 = py_intersect_aabb(o, d, t[c2]["aabb_min"], t[c2]["aabb_max"], closest)
            if d1 > d2:
                d1, d2, c1, c2 = d2, d1, c2, c1
            if d1 == MISS:
                if len(stack) == 1:
                    break
                stack.pop()
            else:
                stack.pop()
                if d2 != MISS:
                    stack.append(c2)
                stack.append(c1)
    return hit


@pytest.mark.parametrize("scene", ["cornell", "mixed"])
def test_items_5_to_10_second_restatement_of_the_traversal(scene):
    sc = {"cornell": S.cornell_builtin, "mixed": S.mixed_scene}[scene]().build()
    rays = S.random_rays(160, seed=11, center=(0, 0.6, 0), radius=3.0)
    # axis-parallel rays: 0/0 and x/0 in the slab test (IEEE inf/NaN through the compare-select min/max)
    rays["direction"][:12] = np.array([(0, -1, 0), (1, 0, 0), (0, 0, -1), (0, 1, 0)] * 3, F)
    want = O.trace_closest(sc, rays)
    hits = 0
    for k, ray in enumerate(rays):
        got = py_tlas_intersect(sc, ray["origin"][:3].astype(F), ray["direction"][:3].astype(F), F(ray["t_min"]), F(ray["t_max"]))
        if got is None:
            assert want["tri_id"][k] == T.INVALID_ID, k
            continue
        hits += 1
        t, tri, u, v, inst = got
        assert (F(t).view(np.uint32), tri, inst) == (want["t"][k].view(np.uint32), int(want["tri_id"][k]), int(want["inst_id"][k])), k
        assert F(u) == want["u"][k] and F(v) == want["v"][k], k
    assert hits > 30


def test_item_8_equal_distances_keep_child1_first_and_item_9_later_equal_t_wins():
    # two coplanar quads in ONE BLAS; the rays go through a quad's diagonal, where both of its triangles return the
    # same t and their leaf boxes are entered at the same distance: child1 is visited first, and the triangle tested
    # LATER replaces the earlier hit (inclusive range)
    pos = np.array([(0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 0, 1),  (2, 0, 0), (3, 0, 0), (2, 0, 1), (3, 0, 1)], F)
    idx = np.array([0, 2, 1, 1, 2, 3,  4, 6, 5, 5, 6, 7], np.uint32)
    r = O.OracleRenderer()
    b = r.add_blas(pos, np.tile(np.array([[0, 1, 0]], F), (8, 1)), np.zeros((8, 2), F), idx)
    r.add_blas_instance(b, O.trs(), r.default_material)
    sc = r.build()
    rays = np.zeros(2, T.ray_dtype)
    rays["origin"], rays["direction"] = [(0.5, 1, 0.5), (2.5, 1, 0.5)], [(0, -1, 0)] * 2
    rays["t_min"], rays["t_max"] = F(1e-4), F(1000)
    want = O.trace_closest(sc, rays)
    for k, ray in enumerate(rays):
        got = py_tlas_intersect(sc, ray["origin"][:3].astype(F), ray["direction"][:3].astype(F), F(1e-4), F(1000))
        assert got is not None and (got[1], got[4]) == (int(want["tri_id"][k]), int(want["inst_id"][k]))
        assert got[0] == F(1.0)
