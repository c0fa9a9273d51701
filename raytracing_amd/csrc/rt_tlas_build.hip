// rt_tlas_build.hip — the agglomerative clustering loop of TLAS::build (src/TLAS.cpp:38-58) and its
// find_best_match (:61-80) on one MI355X compute unit, bit-equal to the reference's sequential loop.
//
// The reference rebuilds the TLAS whenever an instance moves, is added or removed (Renderer.cpp:994-1015): the
// loop keeps a list of live clusters, repeatedly scans it for the cluster whose union with cluster A has the
// smallest surface area (first index wins ties), and merges mutual best matches.  The loop itself is sequential —
// every step depends on the previous one — so the kernel runs it in ONE 512-lane workgroup on one compute unit:
// lane t keeps the boxes of list slots t, t + 512, ... in registers (8 slots = 48 VGPRs), a scan is up to 8
// unrolled box unions per lane with an (area bits, index) minimum reduced by DPP row operations inside the wave
// and through 64 bytes of LDS plus one barrier across the eight waves (about 1 us per scan, 3.6 scans per
// instance).  A copy of the list in LDS (36 B per
// slot) serves the broadcast read of the query box and the bookkeeping of node ids.  All control state (a, b,
// remaining, node count) is recomputed redundantly by every lane from the reduced key.  Arithmetic as the
// reference: fp32, no contraction, area = ex*ey + ey*ez + ez*ex, strict '<' against +inf; the node boxes are
// merged with glm's compare-select min/max.  Measured (tools/bench_tlas.py): 4000 instances 15.6 ms against
// 37.4 ms on one host core, break-even near 1500 instances.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_hip.h"
#include "../../include/rt_types.h"
#include "rt_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int TB_THREADS = 512; // measured: 256 / 512 / 1024 lanes -> 21.7 / 15.6 / 16.1 ms at 4000 instances
constexpr int TB_WAVES = TB_THREADS / 64;
constexpr int TB_PER_LANE = RT_TLAS_GPU_MAX_LEAVES / TB_THREADS; // list slots owned by one lane (registers)
constexpr uint32_t TB_NONE32 = 0xFFFFFFFFu;

__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; } // glm::min
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; } // glm::max
// the plain machine instructions (the fmaxf/fminf builtins add a canonicalising v_max x,x per operand)
__device__ __forceinline__ float hw_max(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float hw_min(float x, float y) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// min over the lanes of a 16-lane row by DPP row shifts (min is idempotent, so overlapping windows are fine);
// lanes without a source keep their own value.  The row's minimum ends in its lane 15.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_min_step(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
  return o < v ? o : v;
}
__device__ __forceinline__ uint32_t row_min_u32(uint32_t v) {
  v = dpp_min_step<0x111, 0xF>(v); // row_shr:1
  v = dpp_min_step<0x112, 0xF>(v); // row_shr:2
  v = dpp_min_step<0x114, 0xF>(v); // row_shr:4
  v = dpp_min_step<0x118, 0xF>(v); // row_shr:8
  return v;
}
// min over the 64 lanes of the wave, returned uniformly (read from lane 63)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = row_min_u32(v);
  v = dpp_min_step<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
  v = dpp_min_step<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct LiveList { // copy of the list in LDS (structure of arrays, `cap` entries each): broadcast reads and bookkeeping
  float *lox, *loy, *loz, *hix, *hiy, *hiz;
  uint32_t *node, *left_right, *inst;
};
struct Box {
  float lx, ly, lz, hx, hy, hz;
};
__device__ __forceinline__ Box lds_box(const LiveList &L, int k) {
  return {L.lox[k], L.loy[k], L.loz[k], L.hix[k], L.hiy[k], L.hiz[k]};
}
__device__ __forceinline__ void lds_store_box(const LiveList &L, int k, const Box &b) {
  L.lox[k] = b.lx, L.loy[k] = b.ly, L.loz[k] = b.lz, L.hix[k] = b.hx, L.hiy[k] = b.hy, L.hiz[k] = b.hz;
}

// find_best_match (TLAS.cpp:61-80) over live[0, remaining): the index b != a that minimises the surface area of
// box(a) ∪ box(b), the first such index on ties, or -1 if no area is below +inf.  Lane t scans the slots
// t, t + 512, ... it owns (boxes in registers `mine`); the (area bits, index) minimum is taken as two 32-bit
// minima (areas of boxes are >= +0, so their bit patterns order like the floats).
__device__ __forceinline__ int best_match(const Box (&mine)[TB_PER_LANE], const Box &qa, int remaining, int a,
                                          uint32_t (*red)[TB_WAVES][2], int &phase) {
  uint32_t best_area = TB_NONE32, best_b = TB_NONE32;
#pragma unroll
  for (int j = TB_PER_LANE - 1; j >= 0; --j) { // descending, so that '<=' keeps the smaller index on ties
    const int b = (int)threadIdx.x + j * TB_THREADS;
    if (j * TB_THREADS < remaining) { // uniform: the list only shrinks
      // The boxes are finite (checked by the caller), so v_max/v_min equal glm's compare-select max/min except
      // for the sign of a zero; that can only change the sign of a zero area, which the reference's float
      // comparison does not see and which is canonicalised to +0 here.  Slots at or behind `remaining` hold
      // hx = +inf, which makes their area inf or NaN: the reference's own `area < smallest` test drops them.
      const float hx = hw_max(qa.hx, mine[j].hx), hy = hw_max(qa.hy, mine[j].hy), hz = hw_max(qa.hz, mine[j].hz);
      const float lx = hw_min(qa.lx, mine[j].lx), ly = hw_min(qa.ly, mine[j].ly), lz = hw_min(qa.lz, mine[j].lz);
      const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
      const float area = ex * ey + ey * ez + ez * ex;
      const uint32_t bits = area == 0.0f ? 0u : __float_as_uint(area);
      if (b != a && area < __builtin_huge_valf() && bits <= best_area)
        best_area = bits, best_b = (uint32_t)b;
    }
  }
  const uint32_t wave_area = wave_min_u32(best_area);
  const uint32_t wave_b = wave_min_u32(best_area == wave_area ? best_b : TB_NONE32);
  if ((threadIdx.x & 63u) == 0)
    red[phase][threadIdx.x >> 6][0] = wave_area, red[phase][threadIdx.x >> 6][1] = wave_b;
  __syncthreads();
  const uint32_t ra = red[phase][threadIdx.x & (TB_WAVES - 1)][0], rb = red[phase][threadIdx.x & (TB_WAVES - 1)][1];
  const uint32_t all_area = (uint32_t)__builtin_amdgcn_readlane((int)row_min_u32(ra), 15);
  const uint32_t all_b = (uint32_t)__builtin_amdgcn_readlane((int)row_min_u32(ra == all_area ? rb : TB_NONE32), 15);
  phase ^= 1; // the next scan writes the other row: no second barrier needed
  return all_area == TB_NONE32 ? -1 : (int)all_b;
}

// slot k of the list takes box v: its owner lane updates its registers (k is uniform: one case runs)
__device__ __forceinline__ void set_mine(Box (&mine)[TB_PER_LANE], int k, const Box &v) {
  const int kj = k / TB_THREADS;
  const bool owner = (int)threadIdx.x == k % TB_THREADS;
#pragma unroll
  for (int j = 0; j < TB_PER_LANE; ++j)
    if (j == kj && owner)
      mine[j] = v;
}

__global__ __launch_bounds__(TB_THREADS) void k_tlas_cluster(const rt_tlas_node *__restrict__ leaves, uint32_t n,
                                                             uint32_t cap, rt_tlas_node *__restrict__ out,
                                                             uint32_t *__restrict__ out_count) {
  extern __shared__ uint32_t tb_lds[];
  __shared__ uint32_t red[2][TB_WAVES][2];
  LiveList L;
  L.lox = reinterpret_cast<float *>(tb_lds);
  L.loy = L.lox + cap, L.loz = L.loy + cap, L.hix = L.loz + cap, L.hiy = L.hix + cap, L.hiz = L.hiy + cap;
  L.node = tb_lds + 6 * cap, L.left_right = L.node + cap, L.inst = L.left_right + cap;

  Box mine[TB_PER_LANE];
#pragma unroll
  for (int j = 0; j < TB_PER_LANE; ++j) { // leaves become nodes 1..n (TLAS.cpp:15-36)
    const uint32_t k = threadIdx.x + j * TB_THREADS;
    mine[j] = Box{0.f, 0.f, 0.f, __builtin_huge_valf(), 0.f, 0.f}; // not in the list
    if (k < n) {
      const rt_tlas_node leaf = leaves[k];
      out[1 + k] = leaf;
      mine[j] = Box{leaf.aabb_min[0], leaf.aabb_min[1], leaf.aabb_min[2],
                    leaf.aabb_max[0], leaf.aabb_max[1], leaf.aabb_max[2]};
      lds_store_box(L, (int)k, mine[j]);
      L.node[k] = 1 + k, L.left_right[k] = leaf.left_right, L.inst[k] = leaf.blas_instance_idx;
    }
  }
  __syncthreads();

  int remaining = (int)n, phase = 0;
  uint32_t count = n + 1;
  int a = 0, b = best_match(mine, lds_box(L, 0), remaining, a, red, phase); // TLAS.cpp:39-55
  bool bad = false;
  while (remaining > 1) {
    if (b < 0) { // every union area is inf/NaN: the reference would index with -1; stop instead
      bad = true;
      break;
    }
    const Box box_b = lds_box(L, b);
    const int c = best_match(mine, box_b, remaining, b, red, phase);
    if (a == c) {
      // every lane computes the merged box (uniform broadcast reads); the owners of slots a and b update their
      // registers, lane 0 the LDS copy and the node array.  All lanes are past the scan's barrier here.
      const Box box_a = lds_box(L, a);
      const Box merged = {gmin(box_a.lx, box_b.lx), gmin(box_a.ly, box_b.ly), gmin(box_a.lz, box_b.lz),
                          gmax(box_a.hx, box_b.hx), gmax(box_a.hy, box_b.hy), gmax(box_a.hz, box_b.hz)};
      const int last = remaining - 1; // live[a] = merged, then live[b] = live[remaining - 1] (a may be `last`)
      const Box moved = last == a ? merged : lds_box(L, last);
      __syncthreads(); // the broadcast reads above are done before lane 0 overwrites the slots
      set_mine(mine, a, merged);
      set_mine(mine, b, moved);
      set_mine(mine, last, Box{0.f, 0.f, 0.f, __builtin_huge_valf(), 0.f, 0.f}); // in this order: last may be a or b
      if (threadIdx.x == 0) {
        rt_tlas_node m;
        m.aabb_min[0] = merged.lx, m.aabb_min[1] = merged.ly, m.aabb_min[2] = merged.lz;
        m.aabb_max[0] = merged.hx, m.aabb_max[1] = merged.hy, m.aabb_max[2] = merged.hz;
        m.left_right = L.node[a] + (L.node[b] << 16);
        m.blas_instance_idx = 0;
        out[count] = m;
        lds_store_box(L, a, merged);
        L.node[a] = count, L.left_right[a] = m.left_right, L.inst[a] = 0;
        lds_store_box(L, b, moved);
        L.node[b] = L.node[last], L.left_right[b] = L.left_right[last], L.inst[b] = L.inst[last];
      }
      ++count;
      --remaining;
      __syncthreads();
      b = best_match(mine, merged, remaining, a, red, phase);
    } else {
      a = b;
      b = c;
    }
  }
  if (threadIdx.x == 0) {
    rt_tlas_node root; // nodes[0] = nodes[live[a]], TLAS.cpp:57
    root.aabb_min[0] = L.lox[a], root.aabb_min[1] = L.loy[a], root.aabb_min[2] = L.loz[a];
    root.aabb_max[0] = L.hix[a], root.aabb_max[1] = L.hiy[a], root.aabb_max[2] = L.hiz[a];
    root.left_right = L.left_right[a], root.blas_instance_idx = L.inst[a];
    out[0] = root;
    *out_count = bad ? 0u : count - 1;
  }
}

} // namespace

size_t rt_tlas_cluster_lds_bytes(uint32_t cap) { return (size_t)cap * 9 * sizeof(uint32_t); }

hipError_t rt_launch_tlas_cluster(const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out, uint32_t *out_count,
                                  hipStream_t stream) {
  // capacity in steps of 256 entries so that small scenes do not reserve the whole LDS
  uint32_t cap = (n + 255u) & ~255u;
  const size_t lds = rt_tlas_cluster_lds_bytes(cap);
  static bool attr_done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tlas_cluster), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)rt_tlas_cluster_lds_bytes(RT_TLAS_GPU_MAX_LEAVES));
    if (e != hipSuccess)
      return e;
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL(k_tlas_cluster, dim3(1), dim3(TB_THREADS), lds, stream, leaves, n, cap, out, out_count);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// rt_build_tlas_cluster_noalloc: the same kernel for callers that cannot allocate.  Its device buffers (the staged
// leaves, the 2n output nodes, the node count) live in this file-scope array instead of hipMalloc'd memory:
// RT_TLAS_GPU_MAX_LEAVES leaves in, twice as many nodes out, one counter = 384 KiB + 256 B, held by every GPU the
// library runs on until the process ends.  The array's name is only a host-side handle: each device has its own copy
// at its own address, looked up once per device with hipGetSymbolAddress (rt_tlas_scratch_init, called by rt_create
// with that device current) and used through that pointer everywhere below.  Every call on a GPU shares that GPU's
// array, so calls that overlap on one GPU (two contexts, two threads) race; one context's calls are ordered by its stream.
// ---------------------------------------------------------------------------
#define TB_SCRATCH_LEAVES_BYTES ((size_t)RT_TLAS_GPU_MAX_LEAVES * sizeof(rt_tlas_node))
#define TB_SCRATCH_OUT_BYTES (2 * (size_t)RT_TLAS_GPU_MAX_LEAVES * sizeof(rt_tlas_node))
#define TB_SCRATCH_BYTES (TB_SCRATCH_LEAVES_BYTES + TB_SCRATCH_OUT_BYTES + 256)
static __device__ __align__(256) unsigned char scratch[TB_SCRATCH_BYTES];
static unsigned char *g_scratch_ptr[64]; // per device: the address of that device's copy of `scratch`

hipError_t rt_tlas_scratch_init(int device) { // `device` is current
  if (device < 0 || device >= 64)
    return hipSuccess; // the no-allocation entry point refuses such a device; everything else works
  if (g_scratch_ptr[device])
    return hipSuccess;
  void *p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  if (e != hipSuccess)
    return e;
  if (!p)
    return hipErrorInvalidSymbol;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_tlas_cluster), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)rt_tlas_cluster_lds_bytes(RT_TLAS_GPU_MAX_LEAVES));
  if (e != hipSuccess)
    return e;
  g_scratch_ptr[device] = static_cast<unsigned char *>(p);
  return hipSuccess;
}

// Stages `leaves` (host) into the device's scratch, runs the clustering kernel there, copies 2n nodes and the count
// back and waits.  No hipMalloc / hipFree / event creation.  n <= RT_TLAS_GPU_MAX_LEAVES is the caller's check.
hipError_t rt_run_tlas_cluster_noalloc(int device, const rt_tlas_node *leaves, uint32_t n, rt_tlas_node *out_nodes,
                                       uint32_t *out_count, hipStream_t stream) {
  if (device < 0 || device >= 64 || !g_scratch_ptr[device])
    return hipErrorNotInitialized;
  if (n == 0 || n > RT_TLAS_GPU_MAX_LEAVES)
    return hipErrorInvalidValue;
  unsigned char *base = g_scratch_ptr[device];
  rt_tlas_node *d_leaves = reinterpret_cast<rt_tlas_node *>(base);
  rt_tlas_node *d_out = reinterpret_cast<rt_tlas_node *>(base + TB_SCRATCH_LEAVES_BYTES);
  uint32_t *d_count = reinterpret_cast<uint32_t *>(base + TB_SCRATCH_LEAVES_BYTES + TB_SCRATCH_OUT_BYTES);
  // the array keeps the previous call's contents: the count must start at zero (a kernel that did not run then
  // reads as a failure), and the output range of THIS call is cleared so that nothing of a larger, earlier call
  // can be mistaken for a node of this one
  hipError_t e = hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream);
  if (e == hipSuccess)
    e = hipMemsetAsync(d_out, 0, 2 * (size_t)n * sizeof(rt_tlas_node), stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_leaves, leaves, (size_t)n * sizeof(rt_tlas_node), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess)
    return e;
  const uint32_t cap = (n + 255u) & ~255u;
  hipLaunchKernelGGL(k_tlas_cluster, dim3(1), dim3(TB_THREADS), rt_tlas_cluster_lds_bytes(cap), stream, d_leaves, n, cap,
                     d_out, d_count);
  e = hipGetLastError();
  if (e == hipSuccess)
    e = hipMemcpyAsync(out_count, d_count, sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(out_nodes, d_out, 2 * (size_t)n * sizeof(rt_tlas_node), hipMemcpyDeviceToHost, stream);
  const hipError_t es = hipStreamSynchronize(stream); // also on an error path: nothing stays queued on the caller's arrays
  return e != hipSuccess ? e : es;
}
