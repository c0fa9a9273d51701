// accel_build.h — host-side acceleration-structure builders of the renderer.
// They must reproduce the reference's trees node for node (hit IDs depend on
// node numbering and on the triangle order inside leaves):
//   build_blas  <- BLAS::build / subdivide / find_best_split_plane, src/BVHNode.cpp:14-161
//   refit_blas  <- BLAS::refit, src/BVHNode.cpp:163-177
//   build_tlas  <- TLAS::build / find_best_match, src/TLAS.cpp:6-80
#ifndef RTA_ACCEL_BUILD_H
#define RTA_ACCEL_BUILD_H

#include <stdint.h>

#include "../../../include/rt_types.h"

namespace rta {

// nodes: scratch for 2*tri_count-1 entries (BLAS-local numbering, root = 0).
// tris / centroids (xyz packed) / tri_ids are the renderer's global arrays;
// the BLAS owns tri_ids[first, first + tri_count).  Returns the node count.
uint32_t build_blas(rt_bvh_node *nodes, const rt_triangle_geom *tris, const float *centroids,
                    uint32_t *tri_ids, uint32_t tri_count, uint32_t first);

void refit_blas(rt_bvh_node *nodes, uint32_t nodes_count, const rt_triangle_geom *tris,
                const uint32_t *tri_ids);

// tlas_nodes: room for 2*n entries.  instance_ids gives the leaf order
// (leaf k+1 <- instance_ids[k]).  Returns the node count (2n-1; 0 for n == 0).
// The two halves of build_tlas: the leaf boxes (TLAS.cpp:15-36; leaves[k] <- instance_ids[k]) and the clustering
// loop (TLAS.cpp:38-58) over nodes[1..n] already holding those leaves.  The second half has a GPU counterpart,
// rt_build_tlas_cluster (include/rt_hip.h), which rta::Renderer uses for large instance counts.
void tlas_leaves(rt_tlas_node *leaves, const rt_blas_instance *instances, const uint32_t *instance_ids, uint32_t n,
                 const rt_blas *blas, const rt_bvh_node *bvh_nodes);
uint32_t tlas_cluster(rt_tlas_node *nodes, uint32_t n);
uint32_t build_tlas(rt_tlas_node *tlas_nodes, const rt_blas_instance *instances,
                    const uint32_t *instance_ids, uint32_t n, const rt_blas *blas,
                    const rt_bvh_node *bvh_nodes);

} // namespace rta
#endif
