#!/usr/bin/env python3
"""TLAS rebuild (Renderer.cpp:994-1015): clustering loop on one host core (rth_tlas_cluster) vs on the GPU
(rt_build_tlas_cluster, one workgroup).  Prints one line per instance count; results are compared bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import _ffi, _types as T, renderer as R

ctx = _ffi.HipContext(0)
rng = np.random.default_rng(42)
busy = None
if os.environ.get("TLAS_BENCH_BUSY"):  # keep the GPU clocks up with other work right before the timed call
    import torch
    m = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    def busy():
        for _ in range(60):
            m @ m
        torch.cuda.synchronize()
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 64, 128, 192, 256, 512, 1000, 2025, 4000, 4096]:
    lv = np.zeros(n, T.tlas_node_dtype)
    c = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    e = rng.uniform(0.2, 2.0, (n, 3)).astype(np.float32)
    lv["aabb_min"], lv["aabb_max"], lv["blas_instance_idx"] = c - e, c + e, np.arange(n, dtype=np.uint32)
    nodes = np.zeros(2 * n, T.tlas_node_dtype)
    nodes[1:1 + n] = lv
    t0 = time.perf_counter()
    cnt = R.lib().rth_tlas_cluster(T.ptr(nodes), n)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    ctx.build_tlas_cluster(lv)  # warm-up (module load, attribute)
    if busy:
        busy()
    t0 = time.perf_counter()
    got, gcnt, kernel_ms = ctx.build_tlas_cluster(lv)
    gpu_ms = (time.perf_counter() - t0) * 1e3
    same = gcnt == cnt and got[:cnt].tobytes() == nodes[:cnt].tobytes()
    print(f"n={n:5d}  host {cpu_ms:8.3f} ms   gpu call {gpu_ms:8.3f} ms (kernel {kernel_ms:8.3f} ms)   "
          f"speed-up {cpu_ms / gpu_ms:5.2f}x   identical={same}", flush=True)
