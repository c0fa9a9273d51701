// accel_build.cpp — see accel_build.h.  Written as flat loops over an explicit
// work stack (no recursion: a degenerate mesh can make the tree as deep as it
// is large) with the three axes binned in one sweep over the node's
// triangles; every floating-point expression keeps the operand order of the
// reference so the resulting bits are identical.
#include "accel_build.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "hmath.h"

namespace rta {
namespace {

constexpr uint32_t kBins = 200; // BIN_COUNT, src/BVHNode.cpp:5

struct Box { // AABB, src/AABB.hpp:7-27
  vec3 lo{kInfinity}, hi{-kInfinity};
  void grow(vec3 p) { lo = min(lo, p), hi = max(hi, p); }
  void grow(const Box &b) { lo = min(lo, b.lo), hi = max(hi, b.hi); }
  float half_area() const {
    vec3 e = hi - lo;
    return e.x * e.y + e.y * e.z + e.x * e.z;
  }
};

struct AxisBins {
  Box bounds[kBins];
  uint32_t count[kBins];
};

struct Builder {
  rt_bvh_node *nodes;
  const rt_triangle_geom *tris;
  const float *centroids;
  uint32_t *ids;
  uint32_t nodes_count = 1;
  std::vector<AxisBins> bins = std::vector<AxisBins>(3);

  void fit(uint32_t n) { // update_node_bounds, BVHNode.cpp:89-105
    rt_bvh_node &node = nodes[n];
    vec3 lo{kInfinity}, hi{-kInfinity};
    const uint32_t *p = ids + node.local_left_first, *e = p + node.tri_count;
    for (; p != e; ++p) {
      const rt_triangle_geom &t = tris[*p];
      vec3 a(t.v0), b(t.v1), c(t.v2);
      lo = min(min(min(lo, a), b), c);
      hi = max(max(max(hi, a), b), c);
    }
    lo.store(node.aabb_min);
    hi.store(node.aabb_max);
  }

  // find_best_split_plane, BVHNode.cpp:14-74.  Returns the best cost; axis /
  // pos are untouched when no axis has any extent (cost stays +inf).
  float best_split(const rt_bvh_node &node, int &axis, float &pos) {
    const uint32_t *first = ids + node.local_left_first;
    const uint32_t n = node.tri_count;
    float cmin[3] = {kInfinity, kInfinity, kInfinity};
    float cmax[3] = {-kInfinity, -kInfinity, -kInfinity};
    for (uint32_t i = 0; i < n; ++i) {
      const float *c = centroids + 3 * (size_t)first[i];
      for (int a = 0; a < 3; ++a) {
        cmin[a] = c[a] < cmin[a] ? c[a] : cmin[a]; // std::min(bounds_min, c)
        cmax[a] = cmax[a] < c[a] ? c[a] : cmax[a]; // std::max(bounds_max, c)
      }
    }
    bool live[3];
    float scale[3];
    for (int a = 0; a < 3; ++a) {
      live[a] = !(cmin[a] == cmax[a]);
      if (!live[a])
        continue;
      scale[a] = kBins / (cmax[a] - cmin[a]);
      for (uint32_t b = 0; b < kBins; ++b) {
        bins[a].bounds[b] = Box();
        bins[a].count[b] = 0;
      }
    }
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t id = first[i];
      const rt_triangle_geom &t = tris[id];
      const float *c = centroids + 3 * (size_t)id;
      vec3 v0(t.v0), v1(t.v1), v2(t.v2);
      for (int a = 0; a < 3; ++a) {
        if (!live[a])
          continue;
        uint32_t raw = (uint32_t)((c[a] - cmin[a]) * scale[a]);
        uint32_t b = raw < kBins - 1 ? raw : kBins - 1;
        bins[a].count[b]++;
        Box &bb = bins[a].bounds[b];
        bb.grow(v0);
        bb.grow(v1);
        bb.grow(v2);
      }
    }
    float best = kInfinity;
    float left_area[kBins - 1], right_area[kBins - 1];
    int left_count[kBins - 1], right_count[kBins - 1];
    for (int a = 0; a < 3; ++a) {
      if (!live[a])
        continue;
      const AxisBins &ab = bins[a];
      Box lbox, rbox;
      int lsum = 0, rsum = 0;
      for (uint32_t i = 0; i < kBins - 1; ++i) {
        lsum += (int)ab.count[i];
        left_count[i] = lsum;
        lbox.grow(ab.bounds[i]);
        left_area[i] = lbox.half_area();
        rsum += (int)ab.count[kBins - 1 - i];
        right_count[kBins - 2 - i] = rsum;
        rbox.grow(ab.bounds[kBins - 1 - i]);
        right_area[kBins - 2 - i] = rbox.half_area();
      }
      const float step = (cmax[a] - cmin[a]) / kBins;
      for (uint32_t i = 0; i < kBins - 1; ++i) {
        float cost = left_count[i] * left_area[i] + right_count[i] * right_area[i];
        if (cost < best) {
          best = cost;
          pos = cmin[a] + step * (i + 1);
          axis = a;
        }
      }
    }
    return best;
  }

  // subdivide, BVHNode.cpp:107-161, over an explicit stack.  Children are
  // numbered when their parent splits and the left subtree is finished before
  // the right one starts, exactly as the recursion numbers them.
  // One subdivision step (BVHNode.cpp:107-161): splits node n into nodes_count, nodes_count + 1 and returns true, or
  // leaves it a leaf.
  bool split_once(uint32_t n) {
    rt_bvh_node &node = nodes[n];
    int axis = -1;
    float pos = 0.f;
    const float best = best_split(node, axis, pos);
    vec3 e = vec3(node.aabb_max) - vec3(node.aabb_min);
    const float parent_area = e.x * e.y + e.y * e.z + e.z * e.x;
    const float parent_cost = node.tri_count * parent_area;
    if (best >= parent_cost)
      return false;
    // partition (BVHNode.cpp:128-136); signed 64-bit so that first == 0 is safe
    int64_t l = node.local_left_first;
    int64_t r = l + node.tri_count - 1;
    while (l <= r) {
      if (centroids[3 * (size_t)ids[l] + axis] < pos) {
        ++l;
      } else {
        uint32_t t = ids[l];
        ids[l] = ids[r];
        ids[r] = t;
        --r;
      }
    }
    const uint32_t left_count = (uint32_t)(l - node.local_left_first);
    if (left_count == 0 || left_count == node.tri_count)
      return false;
    const uint32_t li = nodes_count++, ri = nodes_count++;
    nodes[li].local_left_first = node.local_left_first;
    nodes[li].tri_count = left_count;
    nodes[ri].local_left_first = (uint32_t)l;
    nodes[ri].tri_count = node.tri_count - left_count;
    node.local_left_first = li;
    node.tri_count = 0;
    fit(li);
    fit(ri);
    return true;
  }

  // subdivide over an explicit stack.  Children are numbered when their parent splits and the left subtree is
  // finished before the right one starts, exactly as the recursion numbers them.
  void run() {
    std::vector<uint32_t> todo{0};
    while (!todo.empty()) {
      const uint32_t n = todo.back();
      todo.pop_back();
      if (!split_once(n))
        continue;
      const uint32_t li = nodes[n].local_left_first;
      todo.push_back(li + 1);
      todo.push_back(li);
    }
  }
};

} // namespace

namespace {

// ---- the same build on several host threads -----------------------------------------------------------------------
// The reference's recursion (BVHNode.cpp:107-161) numbers the two children of a node when it splits and finishes the
// left subtree before the right one.  Subtrees work on disjoint ranges of the triangle-id array, so they can be built
// independently: the top of the tree is split on one thread down to subtrees of at most `grain` triangles, every such
// subtree is built by a Builder of its own into a private node array (numbered from its own root), and a final
// depth-first pass over the top hands out the node numbers in the reference's order and copies the subtrees to their
// places.  Every box, split and leaf is the one the single-threaded build produces.
struct TopNode {
  uint32_t first = 0, count = 0;
  float lo[3], hi[3];
  int left = -1, right = -1; // children in `top` when the node was split on the main thread
  int task = -1;             // subtree built by a worker
};

uint32_t build_blas_parallel(rt_bvh_node *nodes, const rt_triangle_geom *tris, const float *centroids,
                             uint32_t *tri_ids, uint32_t tri_count, uint32_t first, unsigned threads) {
  const uint32_t grain = std::max<uint32_t>(4096u, tri_count / (threads * 8u));
  std::vector<TopNode> top;
  std::vector<int> tasks; // indices into top
  {
    Builder b; // main-thread splitter: node 0 of a two-node scratch is the node being split
    rt_bvh_node scratch[3];
    b.nodes = scratch, b.tris = tris, b.centroids = centroids, b.ids = tri_ids;
    scratch[0].local_left_first = first, scratch[0].tri_count = tri_count;
    b.fit(0);
    TopNode root;
    root.first = first, root.count = tri_count;
    memcpy(root.lo, scratch[0].aabb_min, 12), memcpy(root.hi, scratch[0].aabb_max, 12);
    top.push_back(root);
    std::vector<int> todo{0};
    while (!todo.empty()) {
      const int t = todo.back();
      todo.pop_back();
      if (top[(size_t)t].count <= grain) {
        top[(size_t)t].task = (int)tasks.size();
        tasks.push_back(t);
        continue;
      }
      // one step of Builder::run on this node
      scratch[0].local_left_first = top[(size_t)t].first, scratch[0].tri_count = top[(size_t)t].count;
      memcpy(scratch[0].aabb_min, top[(size_t)t].lo, 12), memcpy(scratch[0].aabb_max, top[(size_t)t].hi, 12);
      b.nodes_count = 1;
      if (!b.split_once(0))
        continue; // stays a leaf
      TopNode l, r;
      l.first = scratch[1].local_left_first, l.count = scratch[1].tri_count;
      memcpy(l.lo, scratch[1].aabb_min, 12), memcpy(l.hi, scratch[1].aabb_max, 12);
      r.first = scratch[2].local_left_first, r.count = scratch[2].tri_count;
      memcpy(r.lo, scratch[2].aabb_min, 12), memcpy(r.hi, scratch[2].aabb_max, 12);
      const int li = (int)top.size();
      top.push_back(l), top.push_back(r);
      top[(size_t)t].left = li, top[(size_t)t].right = li + 1;
      todo.push_back(li + 1), todo.push_back(li);
    }
  }
  // workers: largest subtrees first
  std::vector<std::vector<rt_bvh_node>> built(tasks.size());
  std::vector<uint32_t> built_count(tasks.size(), 0);
  std::vector<size_t> order(tasks.size());
  for (size_t i = 0; i < order.size(); ++i)
    order[i] = i;
  std::sort(order.begin(), order.end(), [&](size_t a, size_t c) {
    return top[(size_t)tasks[a]].count > top[(size_t)tasks[c]].count;
  });
  std::atomic<size_t> next_task{0};
  auto work = [&]() {
    for (;;) {
      const size_t k = next_task.fetch_add(1);
      if (k >= order.size())
        return;
      const size_t ti = order[k];
      const TopNode &t = top[(size_t)tasks[ti]];
      std::vector<rt_bvh_node> &local = built[ti];
      local.assign((size_t)t.count * 2 - 1, rt_bvh_node{});
      Builder b;
      b.nodes = local.data(), b.tris = tris, b.centroids = centroids, b.ids = tri_ids;
      local[0].local_left_first = t.first, local[0].tri_count = t.count;
      memcpy(local[0].aabb_min, t.lo, 12), memcpy(local[0].aabb_max, t.hi, 12);
      b.run();
      built_count[ti] = b.nodes_count;
    }
  };
  std::vector<std::thread> pool;
  for (unsigned i = 1; i < threads; ++i)
    pool.emplace_back(work);
  work();
  for (std::thread &th : pool)
    th.join();
  // numbering pass: depth first, children numbered at the split, left subtree before right
  uint32_t next = 1;
  struct Item {
    int t;
    uint32_t id;
  };
  std::vector<Item> stack{{0, 0u}};
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const TopNode &t = top[(size_t)it.t];
    rt_bvh_node &out = nodes[it.id];
    memcpy(out.aabb_min, t.lo, 12), memcpy(out.aabb_max, t.hi, 12);
    if (t.task >= 0) {
      const std::vector<rt_bvh_node> &local = built[(size_t)t.task];
      const uint32_t n_local = built_count[(size_t)t.task], base = next;
      auto place = [&](uint32_t k) -> rt_bvh_node & { return k == 0 ? nodes[it.id] : nodes[base + k - 1]; };
      for (uint32_t k = 0; k < n_local; ++k) {
        rt_bvh_node v = local[k];
        if (v.tri_count == 0)
          v.local_left_first = base + v.local_left_first - 1; // child index: local -> global
        place(k) = v;
      }
      next += n_local - 1;
    } else if (t.left >= 0) {
      const uint32_t li = next++, ri = next++;
      out.local_left_first = li, out.tri_count = 0;
      stack.push_back({t.right, ri});
      stack.push_back({t.left, li});
    } else {
      out.local_left_first = t.first, out.tri_count = t.count;
    }
  }
  return next;
}

} // namespace

uint32_t build_blas(rt_bvh_node *nodes, const rt_triangle_geom *tris, const float *centroids,
                    uint32_t *tri_ids, uint32_t tri_count, uint32_t first) {
  // large meshes on several threads (same nodes, same triangle order); RTA_BUILD_THREADS=1 forces one
  unsigned threads = std::thread::hardware_concurrency();
  threads = threads > 16u ? 16u : (threads ? threads : 1u);
  if (const char *env = getenv("RTA_BUILD_THREADS")) {
    const long v = atol(env);
    if (v >= 1 && v <= 64)
      threads = (unsigned)v;
  }
  if (threads > 1 && tri_count >= 32768u)
    return build_blas_parallel(nodes, tris, centroids, tri_ids, tri_count, first, threads);
  Builder b;
  b.nodes = nodes, b.tris = tris, b.centroids = centroids, b.ids = tri_ids;
  nodes[0].local_left_first = first;
  nodes[0].tri_count = tri_count;
  b.fit(0);
  b.run();
  return b.nodes_count;
}

void refit_blas(rt_bvh_node *nodes, uint32_t nodes_count, const rt_triangle_geom *tris,
                const uint32_t *tri_ids) {
  Builder b;
  b.nodes = nodes, b.tris = tris, b.centroids = nullptr, b.ids = const_cast<uint32_t *>(tri_ids);
  for (uint32_t k = nodes_count; k-- > 0;) {
    rt_bvh_node &node = nodes[k];
    if (node.tri_count) {
      b.fit(k);
    } else {
      const rt_bvh_node &lc = nodes[node.local_left_first], &rc = nodes[node.local_left_first + 1];
      min(vec3(lc.aabb_min), vec3(rc.aabb_min)).store(node.aabb_min);
      max(vec3(lc.aabb_max), vec3(rc.aabb_max)).store(node.aabb_max);
    }
  }
}

namespace {
// find_best_match, TLAS.cpp:61-80
int best_match(const rt_tlas_node *nodes, const std::vector<int> &list, int count, int a) {
  float smallest = kInfinity;
  int best = -1;
  const vec3 alo(nodes[list[a]].aabb_min), ahi(nodes[list[a]].aabb_max);
  for (int b = 0; b < count; ++b) {
    if (b == a)
      continue;
    vec3 hi = max(ahi, vec3(nodes[list[b]].aabb_max));
    vec3 lo = min(alo, vec3(nodes[list[b]].aabb_min));
    vec3 e = hi - lo;
    float area = e.x * e.y + e.y * e.z + e.z * e.x;
    if (area < smallest) {
      smallest = area;
      best = b;
    }
  }
  return best;
}
} // namespace

void tlas_leaves(rt_tlas_node *leaves, const rt_blas_instance *instances, const uint32_t *instance_ids, uint32_t n,
                 const rt_blas *blas, const rt_bvh_node *bvh_nodes) {
  for (uint32_t k = 0; k < n; ++k) { // TLAS.cpp:15-36
    const uint32_t id = instance_ids[k];
    const rt_bvh_node &root = bvh_nodes[blas[instances[id].blas_id].bvh_nodes_offset];
    const vec3 lo(root.aabb_min), hi(root.aabb_max);
    const mat4 to_world = inverse(mat4(instances[id].inv_transform)); // sic, TLAS.cpp:22-23
    Box world;
    for (int j = 0; j < 8; ++j) {
      vec3 corner((j & 1) ? hi.x : lo.x, (j & 2) ? hi.y : lo.y, (j & 4) ? hi.z : lo.z);
      vec4 p = to_world * vec4(corner, 1.0f);
      world.grow(vec3(p.x, p.y, p.z));
    }
    rt_tlas_node &leaf = leaves[k];
    world.lo.store(leaf.aabb_min);
    world.hi.store(leaf.aabb_max);
    leaf.blas_instance_idx = id;
    leaf.left_right = 0;
  }
}

uint32_t tlas_cluster(rt_tlas_node *out, uint32_t n) {
  if (n == 0)
    return 0;
  uint32_t count = n + 1;
  std::vector<int> live(n);
  for (uint32_t k = 0; k < n; ++k)
    live[k] = (int)(k + 1);
  int remaining = (int)n;
  // find_best_match returns -1 when no union area is below +inf (boxes with inf / NaN extents or areas that
  // overflow); the reference then indexes its list with -1.  Any other cluster is taken instead, so that the
  // loop terminates with a valid tree.
  auto match = [&](int of) {
    const int m = best_match(out, live, remaining, of);
    return m >= 0 ? m : (of == 0 ? 1 : 0);
  };
  int a = 0, b = remaining > 1 ? match(a) : -1; // TLAS.cpp:39-55
  while (remaining > 1) {
    int c = match(b);
    if (a == c) {
      const int ia = live[a], ib = live[b];
      rt_tlas_node &merged = out[count];
      merged.left_right = (uint32_t)ia + ((uint32_t)ib << 16);
      merged.blas_instance_idx = 0;
      min(vec3(out[ia].aabb_min), vec3(out[ib].aabb_min)).store(merged.aabb_min);
      max(vec3(out[ia].aabb_max), vec3(out[ib].aabb_max)).store(merged.aabb_max);
      live[a] = (int)count++;
      live[b] = live[remaining - 1];
      --remaining;
      b = remaining > 1 ? match(a) : -1;
    } else {
      a = b;
      b = c;
    }
  }
  out[0] = out[live[a]]; // TLAS.cpp:57
  return count - 1;
}

uint32_t build_tlas(rt_tlas_node *out, const rt_blas_instance *instances, const uint32_t *instance_ids,
                    uint32_t n, const rt_blas *blas, const rt_bvh_node *bvh_nodes) {
  if (n == 0)
    return 0;
  tlas_leaves(out + 1, instances, instance_ids, n, blas, bvh_nodes);
  return tlas_cluster(out, n);
}

} // namespace rta
