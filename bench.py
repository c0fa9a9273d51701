#!/usr/bin/env python3
"""bench.py — Mrays/s (incl. bounces) of the HIP render path on BASELINE config C3.

Workload (N=1): dragon-class mesh (procedural stand-in, 870,400 triangles, seed
1234 — the reference's dragon.obj is missing from its snapshot) on a ground
plane under the sky gradient, 1920x1080, 256 spp per step (one launch of the
path-tracing megakernel with a 16x16 stratified grid), max_depth 8, reference
RNG (Wang) and reference slab arithmetic (division).  A step is one full frame;
successive steps are successive progressive frames (RayTracing.slang:176-179).

N>1 (one process per GPU, RCCL): the scene is replicated, rank r renders the
progressive frames r, r+N, ... (the reference reseeds per frame_index, so frames
are independent sample sets); after every step the per-GPU running means are
summed onto rank 0 with one RCCL reduce inside the timed region.  Per-GPU work
is fixed => "scaling": "weak".

The CPU baseline is the oracle (oracle/liboracle.so, kind "port": the
reference's own CPU path does not exist) timed on rank 0 on a
bounded sample of the same workload; the same oracle run supplies the
algorithmic bytes per ray of the roofline and a bit-exact
parity check of a band of the GPU image.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_dragon_scene(R, renderer):
    """C3: dragon stand-in + ground plane, lambert, sky lit."""
    mesh = R.Mesh.dragon(1700, 256, 1234)
    dragon = renderer.add_blas(mesh)
    body = renderer.add_lambert_material((0.55, 0.62, 0.35))
    ground = renderer.add_lambert_material((0.6, 0.6, 0.6))
    renderer.add_blas_instance(dragon, R.trs(), body)
    renderer.add_blas_instance(renderer.plane_blas_index, R.trs((0, -0.28, 0), 0.0, (0, 1, 0), (14, 1, 14)), ground)
    cam = R.Camera((0.6, 1.7, 3.9), (0.15, 0.5, 0.0), (0, 1, 0), 40.0)
    return cam, mesh.triangle_count + 2


def algorithmic_bytes_per_ray(cn, spp_per_launch):
    """SURVEY.md §8(d), reference layouts: per interior visit two 32-B child
    nodes; per TLAS leaf a 144-B instance + 16-B BLAS record; per triangle test
    4-B tri id + 48-B TriangleGeom; per shaded hit 80-B TriangleShading + 144-B
    instance + 8-B material (lambert index + 1x1 texel); per pixel-sample the
    rgba32f read-modify-write amortised over the launch's samples."""
    rays = max(cn["rays"], 1)
    trav = 64 * cn["tlas_interior"] + 160 * cn["tlas_leaf"] + 64 * cn["blas_interior"] + 52 * cn["tris_tested"]
    shade = (80 + 144 + 8) * cn["shaded_hits"]
    return (trav + shade) / rays, 32.0 / spp_per_launch


KERNEL_NAMES = {0: "k_render_v0", 1: "k_render_v1", 2: "k_render_v2<SCHED=0>", 3: "k_render_v2<NCACHE=1536>",
                4: "k_wf_init+k_wf_extend+k_wf_shade", 5: "k_render_v2<SCHED=1>", 6: "k_render_v2<SCHED=2>",
                7: "k_render_v7", 8: "k_render_v2<SCHED=3>", 10: "k_render_v10"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step (perfect square)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--aabb", choices=["div", "rcp"], default="div")
    ap.add_argument("--variant", type=int, default=8)
    ap.add_argument("--shard", choices=["frames", "rows", "interleave"], default="frames",
                    help="N>1: progressive frames round-robin + reduce (default), contiguous row bands + all-gather, "
                         "or interleaved 8-row groups + all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on a box with fewer GPUs than ranks: ranks share the GPUs (local_rank %% device_count) and "
                         "talk over gloo instead of RCCL.  Exercises the multi-rank code path; its numbers mean nothing.")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from raytracing_amd import _ffi, _types as T
    from raytracing_amd import renderer as R
    from raytracing_amd import sharding

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    if args.rehearse:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    W, H = args.width, args.height
    sqrt_spp = int(round(args.spp ** 0.5))
    assert sqrt_spp * sqrt_spp == args.spp, "--spp must be a perfect square (Renderer.cpp:625)"

    t_setup = time.time()
    r = R.Renderer(W, H, device=local_rank)
    cam, n_tris = build_dragon_scene(R, r)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, args.spp)
    r.set_option(R.OPT_MAX_DEPTH, args.depth)
    r.set_option(R.OPT_AABB_MODE, T.AABB_RCP if args.aabb == "rcp" else T.AABB_DIV)
    bands, interleave = None, None
    if args.shard == "rows" and world > 1:  # every rank renders its row band of every frame
        bands = [sharding.row_shard(k, world, H) for k in range(world)]
        r.set_option(R.OPT_ROW_BEGIN, bands[rank][0])
        r.set_option(R.OPT_ROW_END, bands[rank][1])
    elif args.shard == "interleave" and world > 1:  # every rank renders row groups rank, rank+N, ... of every frame
        interleave = sharding.row_interleave(rank, world, H)
        r.set_option(R.OPT_ROW_BEGIN, interleave["row_begin"])
        r.set_option(R.OPT_ROW_END, interleave["row_end"])
        r.set_option(R.OPT_ROW_GROUP_STRIDE, interleave["row_group_stride"])
    else:
        shard = sharding.frame_shard(rank, world)
        r.set_option(R.OPT_FRAME_STRIDE, shard["frame_stride"])
        r.set_option(R.OPT_FRAME_OFFSET, shard["frame_offset"])
    r.set_option(R.OPT_KERNEL_VARIANT, args.variant)
    ctx = C.c_void_p(r.hip_context_handle())
    L = _ffi.lib()

    # the accumulation image lives in a torch tensor so RCCL can read it
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(accum.data_ptr()), W, H))
    reduced = torch.empty_like(accum) if world > 1 else None
    setup_s = time.time() - t_setup

    def step():
        r.render(cam)          # one launch: args.spp samples for every pixel
        if world > 1:
            r.wait()           # kernel runs on the library's stream
            if interleave is not None:
                sharding.gather_interleaved(accum)       # every rank: the complete image, bit-exact
            elif bands is not None:
                sharding.gather_rows(accum, bands)       # every rank: the complete image, bit-exact
            else:
                sharding.combine(accum, out=reduced, dst=0)  # rank 0: (1/N) * sum of the N running means
            # the collective reads `accum` on torch's stream; the next launch writes it on the library's
            torch.cuda.current_stream().synchronize()

    def fence():
        r.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _ffi._check(L.rt_reset_stats(ctx))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0

    st = _ffi.Stats()
    _ffi._check(L.rt_get_stats(ctx, C.byref(st)))
    rays_local = float(st.rays_traced)
    vals = torch.tensor([elapsed, rays_local, st.total_kernel_ms], dtype=torch.float64,
                        device="cpu" if args.rehearse else "cuda")
    if world > 1:
        tmax = vals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = vals.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, rays_total = float(tmax[0]), float(tsum[1])
    else:
        elapsed_max, rays_total = elapsed, rays_local

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    kernel_ms = st.total_kernel_ms / max(st.launches, 1)
    rays_per_launch = rays_local / max(st.launches, 1)
    value = rays_total / elapsed_max / 1e6

    cpu_baseline, roofline, parity = None, None, None
    bytes_per_ray = None
    if not args.no_cpu_baseline:
        # rank 0 only.  N = 1: the timed CPU baseline; N > 1: one oracle frame, untimed, for the algorithmic byte
        # counters the roofline needs (cpu_baseline stays null there, as the contract asks)
        from tests import _oracle as O
        sa = r.scene_arrays()                 # the product-built arrays, handed to the checker
        camu = r.camera_uniforms(cam)
        nthreads = args.cpu_threads
        cn = O.Counters()
        img = np.zeros((H, W, 4), np.float32)
        cpu_sqrt = 2
        frames, t_cpu = 0, 0.0
        while (t_cpu < args.cpu_seconds and frames < 64) if world == 1 else frames < 1:
            p = O.render_params(W, H, frames, cpu_sqrt, args.depth,
                                aabb_mode=T.AABB_RCP if args.aabb == "rcp" else T.AABB_DIV)
            t1 = time.perf_counter()
            O.render(sa, camu, p, img, counters=cn, nthreads=nthreads)
            t_cpu += time.perf_counter() - t1
            frames += 1
        cnd = cn.as_dict()
        if world == 1:
            cpu_baseline = {"value": cnd["rays"] / t_cpu / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "port",
                            "sample": f"{frames} progressive frame(s) of {cpu_sqrt * cpu_sqrt} spp, full {W}x{H}, "
                                      f"depth {args.depth}, {cnd['rays']} rays in {t_cpu:.2f} s (oracle/liboracle.so, "
                                      f"{nthreads} threads)"}
        trav_b, accum_b = algorithmic_bytes_per_ray(cnd, args.spp)
        rays_per_sample = cnd["rays"] / (frames * cpu_sqrt * cpu_sqrt * W * H)
        bytes_per_ray = trav_b + accum_b / rays_per_sample
    if not args.no_cpu_baseline and world == 1:
        # parity spot check: the first oracle frame vs the same launch on the GPU
        nthreads = args.cpu_threads
        chk = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(chk.data_ptr()), W, H))
        p0 = O.render_params(W, H, 0, cpu_sqrt, args.depth, aabb_mode=T.AABB_RCP if args.aabb == "rcp" else T.AABB_DIV)
        _ffi._check(L.rt_set_camera(ctx, C.byref(camu)))
        _ffi._check(L.rt_render(ctx, C.byref(p0)))
        r.wait()
        want = O.render(sa, camu, p0, nthreads=nthreads)
        got = chk.cpu().numpy()
        bad = int((got.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
        parity = {"pixels": W * H, "differing": bad, "bar": "bit-exact rgba32f vs oracle at equal spp/seed"}
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(accum.data_ptr()), W, H))
    if bytes_per_ray is not None:
        achieved = bytes_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        # the PMC passes were taken on the default workload; other shapes report no traffic figure
        if os.path.exists(tpath) and (W, H, args.spp, args.depth, args.aabb) == (1920, 1080, 256, 8, "div"):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "kernel": KERNEL_NAMES.get(args.variant & 0xFF, "?"), "kernel_ms": kernel_ms,
                    "algorithmic_bytes_per_ray": bytes_per_ray, "rays_per_launch": rays_per_launch}

    out = {
        "metric": "Mrays/s (incl. bounces) dragon scene", "value": value, "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if (bands is not None or interleave is not None) else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.rehearse else "synthetic (REHEARSAL: shared GPU, gloo)",
        "config": {"workload": "C3 dragon-class stand-in (procedural, seed 1234)", "triangles": n_tris,
                   "image": f"{W}x{H}", "spp_per_step": args.spp, "max_depth": args.depth,
                   "rng": "wang", "aabb": args.aabb, "kernel_variant": args.variant,
                   "sharding": ("none" if world == 1 else "interleaved 8-row groups per rank, RCCL all-gather per step"
                                if interleave is not None else "row bands per rank, RCCL all-gather per step" if bands is not None
                                else "progressive frames round-robin over ranks, RCCL reduce per step")},
        "s_per_frame_1024spp": elapsed_max / args.steps * (1024.0 / (args.spp * (1 if (bands is not None or interleave is not None) else world))),
        "rays_per_step": rays_total / args.steps, "kernel_ms_per_launch": kernel_ms,
        "setup_s": setup_s, "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_check": parity,
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
