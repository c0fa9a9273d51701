"""Golden fixtures: self-contained (scene arrays + inputs + expected outputs), so
a check against them depends on no libm / numpy transcendental on the machine
that runs it.  Written by tests/golden/make_golden.py from the oracle."""
import os

import numpy as np

from raytracing_amd import _types as T

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_FIELDS = ["tri_geom", "tri_shading", "tri_ids", "bvh_nodes", "blas", "instances", "tlas_nodes",
           "lamberts", "metals", "dielectrics", "emissives", "textures", "texels"]


def scene_to_dict(sc, prefix="scene_"):
    d = {prefix + f: getattr(sc, f) for f in _FIELDS}
    d[prefix + "tlas_node_count"] = np.array([sc.tlas_node_count], np.uint32)
    return d


def scene_from_npz(z, prefix="scene_"):
    g = lambda f: np.ascontiguousarray(z[prefix + f])
    return T.SceneArrays(g("tri_geom").view(T.geom_dtype).reshape(-1), g("tri_shading").view(T.shading_dtype).reshape(-1),
                         g("tri_ids"), g("bvh_nodes").view(T.bvh_node_dtype).reshape(-1),
                         g("blas").view(T.blas_dtype).reshape(-1), g("instances").view(T.instance_dtype).reshape(-1),
                         g("tlas_nodes").view(T.tlas_node_dtype).reshape(-1), int(z[prefix + "tlas_node_count"][0]),
                         g("lamberts"), g("metals"), g("dielectrics"), g("emissives"),
                         g("textures").view(T.texture_dtype).reshape(-1), g("texels"))


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
