// half_exec.hip — does a wave64 VALU instruction whose EXEC mask covers only one 32-lane half issue faster on gfx950?
// (If it did, a traversal loop could run its two half-waves out of phase — twice the loads in flight per wave for the
// same VALU time.)  Standalone: hipcc --offload-arch=gfx950 -O3 half_exec.hip -o half_exec && ./half_exec
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE> // 0 all 64 lanes, 1 lanes 0-31, 2 lanes 32-63, 3 even lanes, 4 one lane
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long *cycles) {
  const int lane = threadIdx.x & 63;
  bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? lane >= 32 : MODE == 3 ? (lane & 1) == 0 : lane == 5;
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (on) {
    for (int i = 0; i < iters; ++i) { // four independent chains of fma: issue bound
#pragma unroll 16
      for (int u = 0; u < 16; ++u) {
        a = __builtin_fmaf(a, b, c);
        c = __builtin_fmaf(c, b, d);
        d = __builtin_fmaf(d, b, a);
        b = __builtin_fmaf(b, 0.99999f, 1e-7f);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    *cycles = t1 - t0;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
  float *out;
  unsigned long long *cyc, h = 0;
  const int blocks = 256 * waves_per_simd; // 4 waves per block = one per SIMD; waves_per_simd blocks per CU
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int iters = 20000;
  k<MODE><<<blocks, 256>>>(out, 100, cyc);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double insts = (double)iters * 64; // wave-instructions per wave
  printf("%-14s %d waves/SIMD: %.3f ms, %.2f cycles per wave-instruction per wave (memtime), %.1f G wave-instr/s chip\n", name,
         waves_per_simd, ms, (double)h / insts, insts * blocks * 4 / (ms * 1e-3) / 1e9);
  hipFree(out), hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("all 64 lanes", w);
    run<1>("lanes 0-31", w);
    run<2>("lanes 32-63", w);
    run<3>("even lanes", w);
    run<4>("one lane", w);
  }
  return 0;
}
