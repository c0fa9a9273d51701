import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from raytracing_amd import renderer as R, _types as T
from tests import _oracle as O
W, H = 480, 270
r = R.Renderer(W, H, device=None)
cam, n = bench.build_dragon_scene(R, r)
sa = r.scene_arrays(); camu = r.camera_uniforms(cam)
prev = None
for d in range(1, 9):
    cn = O.Counters()
    p = O.render_params(W, H, 0, 2, d, rng_mode=T.RNG_PCG)
    O.render(sa, camu, p, counters=cn, nthreads=8)
    c = cn.as_dict()
    if prev:
        dr = c['rays'] - prev['rays']
        vis = sum(c[k] - prev[k] for k in ('tlas_interior', 'tlas_leaf', 'blas_interior', 'blas_leaf'))
        print(f"depth {d-1}: rays {dr} node visits/ray {vis/max(dr,1):.1f} (blas interior {(c['blas_interior']-prev['blas_interior'])/max(dr,1):.1f}, blas leaf {(c['blas_leaf']-prev['blas_leaf'])/max(dr,1):.1f}) tris tested/ray {(c['tris_tested']-prev['tris_tested'])/max(dr,1):.2f}")
    else:
        dr = c['rays']; vis = sum(c[k] for k in ('tlas_interior', 'tlas_leaf', 'blas_interior', 'blas_leaf'))
        print(f"depth 0: rays {dr} node visits/ray {vis/dr:.1f} (blas interior {c['blas_interior']/dr:.1f}, blas leaf {c['blas_leaf']/dr:.1f}) tris tested/ray {c['tris_tested']/dr:.2f}")
    prev = c
