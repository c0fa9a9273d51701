#!/usr/bin/env python3
"""Leaf sizes and depth of the reference's binned-SAH BVH (BVHNode.cpp:14-161, as built by the host library) over the
two dragon-class meshes: the regular tube of the bench (C3) and the non-uniformly tessellated subdivided teapot (C3b).
CPU only.  The question: is "99.6 % of the leaves hold ONE triangle"  an artefact of the tube's grid?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from raytracing_amd import renderer as R


def stats(name, build):
    r = R.Renderer(8, 8, device=None)
    build(R, r)
    sa = r.scene_arrays()
    big = int(np.argmax(sa.blas["tri_count"]))
    off, cnt = int(sa.blas["bvh_nodes_offset"][big]), int(sa.blas["nodes_count"][big])
    nodes = sa.bvh_nodes[off:off + cnt]
    leaf = nodes["tri_count"][nodes["tri_count"] > 0]
    hist = np.bincount(np.minimum(leaf, 9))
    # depth by walking (children are left, left + 1; interior local_left_first is BLAS-local)
    depth = np.zeros(cnt, np.int32)
    for i in range(cnt):
        if nodes["tri_count"][i] == 0:
            l = int(nodes["local_left_first"][i])
            depth[l] = depth[l + 1] = depth[i] + 1
    g = sa.tri_geom
    v0 = g["v0"][:, :3]; a = 0.5 * np.linalg.norm(np.cross(g["v1"][:, :3] - v0, g["v2"][:, :3] - v0), axis=1)
    a = a[a > 0]
    print(f"{name}: {int(sa.blas['tri_count'][big])} triangles, {cnt} nodes, {len(leaf)} leaves; leaves by size 1..8, 9+: "
          f"{hist[1:].tolist()}; single-triangle leaves {hist[1] / len(leaf):.4f}; leaf depth mean {depth[nodes['tri_count'] > 0].mean():.1f} "
          f"max {depth.max()}; triangle area p5 / p50 / p95: {np.percentile(a, 5):.3g} / {np.percentile(a, 50):.3g} / {np.percentile(a, 95):.3g}")


stats("C3  regular tube", lambda R_, r: bench.build_dragon_scene(R_, r))
stats("C3b subdivided teapot", lambda R_, r: bench.build_teapot_dragon_scene(R_, r))
