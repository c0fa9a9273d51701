// div_exhaustive.hip — exhaustive check (all 2^23 x 2^23 mantissa pairs) of shortened exact-division sequences for
// the slab test, plus a throughput probe of plain vs packed fp32 VALU.  Standalone: hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off div_exhaustive.hip -o div_exhaustive.  Not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#pragma clang fp contract(off)

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float refined_rcp(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float refined_rcp2(float d) {
  const float r1 = refined_rcp(d);
  const float e = __builtin_fmaf(-d, r1, 1.0f);
  return __builtin_fmaf(e, r1, r1);
}
__device__ __forceinline__ float div3(float n, float d, float r) { // one correction round
  const float q0 = n * r;
  const float e0 = __builtin_fmaf(-d, q0, n);
  return __builtin_fmaf(e0, r, q0);
}
__device__ __forceinline__ float div5(float n, float d, float r) { // two rounds: what the library uses today
  const float q0 = n * r;
  const float e0 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(e0, r, q0);
  const float e1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(e1, r, q1);
}

// one lane = one denominator mantissa md; loops over numerator mantissas [mn_lo, mn_hi)
// out[0] A mismatches (div3, r1), out[1] B mismatches (div3, r2), out[2] C mismatches (div5, r1: must be 0)
// bad_md_A / bad_md_B: bitmaps of denominators with any mismatch
extern "C" __global__ void __launch_bounds__(256)
k_exhaustive(uint32_t mn_lo, uint32_t mn_hi, unsigned long long *out, uint32_t *bad_md_A, uint32_t *bad_md_B) {
  const uint32_t md = blockIdx.x * 256u + threadIdx.x; // < 2^23
  const float d = __uint_as_float(0x3F800000u | md);
  const float r1 = refined_rcp(d), r2 = refined_rcp2(d);
  uint32_t a = 0, b = 0, c = 0;
  for (uint32_t mn = mn_lo; mn < mn_hi; ++mn) {
    const float n = __uint_as_float(0x3F800000u | mn);
    const float want = n / d;
    a += __float_as_uint(div3(n, d, r1)) != __float_as_uint(want);
    b += __float_as_uint(div3(n, d, r2)) != __float_as_uint(want);
    c += __float_as_uint(div5(n, d, r1)) != __float_as_uint(want);
  }
  if (a) { atomicAdd(&out[0], (unsigned long long)a); atomicOr(&bad_md_A[md >> 5], 1u << (md & 31u)); }
  if (b) { atomicAdd(&out[1], (unsigned long long)b); atomicOr(&bad_md_B[md >> 5], 1u << (md & 31u)); }
  if (c) atomicAdd(&out[2], (unsigned long long)c);
}

// scale invariance of the refined reciprocal: r1(d * 2^k) == r1(d) * 2^-k for every mantissa and k in [-30, 30]
extern "C" __global__ void __launch_bounds__(256) k_scale(unsigned long long *out) {
  const uint32_t md = blockIdx.x * 256u + threadIdx.x;
  const float d = __uint_as_float(0x3F800000u | md);
  const float r1 = refined_rcp(d), r2 = refined_rcp2(d);
  uint32_t bad = 0;
  for (int k = -30; k <= 30; ++k) {
    const float s = __uint_as_float((uint32_t)(127 + k) << 23), si = __uint_as_float((uint32_t)(127 - k) << 23);
    bad += refined_rcp(d * s) != r1 * si;
    bad += refined_rcp2(d * s) != r2 * si;
    bad += refined_rcp(-d * s) != -(r1 * si);
  }
  if (bad) atomicAdd(out, (unsigned long long)bad);
}

// throughput probe: dependent-free streams of plain fma, packed fma, packed mul
typedef float f2v __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void __launch_bounds__(256) k_rate(float *out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float m = 1.0000001f, c = 1e-9f;
  const f2v mm = {m, m}, cc = {c, c};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a0 = __builtin_fmaf(a0, m, c), a1 = __builtin_fmaf(a1, m, c), a2 = __builtin_fmaf(a2, m, c), a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c), a5 = __builtin_fmaf(a5, m, c), a6 = __builtin_fmaf(a6, m, c), a7 = __builtin_fmaf(a7, m, c);
      }
    } else if (KIND == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p0 = __builtin_elementwise_fma(p0, mm, cc), p1 = __builtin_elementwise_fma(p1, mm, cc);
        p2 = __builtin_elementwise_fma(p2, mm, cc), p3 = __builtin_elementwise_fma(p3, mm, cc);
        p0 = __builtin_elementwise_fma(p0, mm, cc), p1 = __builtin_elementwise_fma(p1, mm, cc);
        p2 = __builtin_elementwise_fma(p2, mm, cc), p3 = __builtin_elementwise_fma(p3, mm, cc);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        p0 = p0 * mm, p1 = p1 * mm, p2 = p2 * mm, p3 = p3 * mm;
        p0 = p0 * mm, p1 = p1 * mm, p2 = p2 * mm, p3 = p3 * mm;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "all";
  unsigned long long *d_out;
  uint32_t *d_badA, *d_badB;
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMalloc(&d_badA, (1u << 23) / 8));
  CHECK(hipMalloc(&d_badB, (1u << 23) / 8));
  CHECK(hipMemset(d_out, 0, 64));
  CHECK(hipMemset(d_badA, 0, (1u << 23) / 8));
  CHECK(hipMemset(d_badB, 0, (1u << 23) / 8));
  if (!strcmp(mode, "rate") || !strcmp(mode, "all")) {
    float *d_f;
    CHECK(hipMalloc(&d_f, 4096 * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(4096), dim3(256), 0, 0, d_f, iters);
        if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(4096), dim3(256), 0, 0, d_f, iters);
        if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(4096), dim3(256), 0, 0, d_f, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double inst = 4096.0 * 4 * iters * 64.0; // wave-instructions
        printf("rate kind %d (%s): %.3f ms, %.2f G wave-instr/s, %.1f Gflop-lanes/s\n", kind,
               kind == 0 ? "v_fma_f32" : kind == 1 ? "v_pk_fma_f32" : "v_pk_mul_f32", ms, inst / ms / 1e6,
               inst * 64 * (kind == 0 ? 1 : 2) / ms / 1e6);
      }
    }
  }
  if (!strcmp(mode, "scale") || !strcmp(mode, "all")) {
    hipLaunchKernelGGL(k_scale, dim3((1u << 23) / 256), dim3(256), 0, 0, d_out + 4);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad;
    CHECK(hipMemcpy(&bad, d_out + 4, 8, hipMemcpyDeviceToHost));
    printf("scale invariance of refined reciprocals over 2^23 mantissas x 61 exponents x 3: %llu violations\n", bad);
  }
  if (!strcmp(mode, "exhaustive") || !strcmp(mode, "all")) {
    const uint32_t total = 1u << 23;
    const uint32_t slice = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096;   // numerators per launch
    const uint32_t upto = argc > 3 ? (uint32_t)atoi(argv[3]) : total;   // numerators covered (default: all)
    for (uint32_t lo = 0; lo < upto; lo += slice) {
      hipLaunchKernelGGL(k_exhaustive, dim3(total / 256), dim3(256), 0, 0, lo, lo + slice < upto ? lo + slice : upto, d_out, d_badA, d_badB);
      CHECK(hipDeviceSynchronize());
      if ((lo / slice) % 64 == 0) { printf("  numerators %u / %u\n", lo, upto); fflush(stdout); }
    }
    unsigned long long h[3];
    CHECK(hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost));
    uint32_t *bm = (uint32_t *)malloc(total / 8);
    for (int which = 0; which < 2; ++which) {
      CHECK(hipMemcpy(bm, which ? d_badB : d_badA, total / 8, hipMemcpyDeviceToHost));
      uint32_t cnt = 0, first[8], nf = 0;
      for (uint32_t i = 0; i < total; ++i)
        if (bm[i >> 5] >> (i & 31) & 1u) { if (nf < 8) first[nf++] = i; ++cnt; }
      printf("%s: %llu mismatching pairs of %.3e, %u denominators involved; first:", which ? "div3 with r2" : "div3 with r1",
             h[which], (double)upto * total, cnt);
      for (uint32_t k = 0; k < nf; ++k) printf(" 0x%06x", first[k]);
      printf("\n");
    }
    printf("div5 with r1 (the library's sequence): %llu mismatching pairs\n", h[2]);
  }
  return 0;
}
