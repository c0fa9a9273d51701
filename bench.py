#!/usr/bin/env python3
"""bench.py — Mrays/s (incl. bounces) of the HIP render path on BASELINE config C3.

Workload (N=1): dragon-class mesh (procedural stand-in, 870,400 triangles, seed
1234 — the reference's dragon.obj is missing from its snapshot) on a ground
plane under the sky gradient, 1920x1080, 256 spp per step (one launch with a 16x16
stratified grid), max_depth 8, reference slab arithmetic (division).  `--rng pcg`
(default) is the keyed PCG RNG that north_star names — every draw a hash of (pixel,
sample, bounce, dimension) — which lets the library run the samples of a pixel side by
side (kernel 9: one persistent kernel in which every wave fills, traces and shades a
window of path slots, then a resolve kernel); `--rng wang` is the reference's
sequential Wang hash: the default run measures it too and reports it under "wang_rng" —
as the reference samples (progressive frames of one sample per pixel, handed over as a
batch: kernel 9 again) and as one stratified frame (the megakernel, kernel 8).  A step is one full frame; successive
steps are successive progressive frames (RayTracing.slang:176-179).

N>1 (one process per GPU, RCCL): the scene is replicated and every frame is cut
into groups of 8 rows dealt round-robin to the ranks (`--shard interleave`, the
default: SURVEY.md §8(e)); after every step the groups are all-gathered, so every
rank holds the complete image, bit-identical to the one-GPU image.  The exchange
of step k runs on its own stream while step k+1 renders.  Total work is fixed =>
"scaling": "strong".  `--shard frames` (rank r renders frames r, r+N, ...; one
RCCL sum-reduce per step; weak scaling) and `--shard rows` (contiguous bands) remain.

The CPU baseline is the oracle (oracle/liboracle.so, kind "port": the
reference's own CPU path does not exist) timed on rank 0 on a
bounded sample of the same workload; the same oracle run supplies the
algorithmic bytes per ray of the roofline and a bit-exact
parity check of the GPU image.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_dragon_scene(R, renderer, defocus=False):
    """C3: dragon stand-in + ground plane, lambert, sky lit.  defocus=True: the C5 camera (thin lens focused on
    the dragon, Random.slang:57-70 semantics)."""
    mesh = R.Mesh.dragon(1700, 256, 1234)
    dragon = renderer.add_blas(mesh)
    body = renderer.add_lambert_material((0.55, 0.62, 0.35))
    ground = renderer.add_lambert_material((0.6, 0.6, 0.6))
    renderer.add_blas_instance(dragon, R.trs(), body)
    renderer.add_blas_instance(renderer.plane_blas_index, R.trs((0, -0.28, 0), 0.0, (0, 1, 0), (14, 1, 14)), ground)
    if defocus:
        cam = R.Camera((0.6, 1.7, 3.9), (0.15, 0.5, 0.0), (0, 1, 0), 40.0, defocus_angle=0.8, focus_dist=4.1)
    else:
        cam = R.Camera((0.6, 1.7, 3.9), (0.15, 0.5, 0.0), (0, 1, 0), 40.0)
    return cam, mesh.triangle_count + 2


def build_teapot_dragon_scene(R, renderer):
    """C3b: a second dragon-class mesh, NON-uniformly tessellated — assets/teapot.obj (1024 triangles whose areas
    span two orders of magnitude) subdivided 1:4 five times and displaced by seeded noise: 1,048,576 triangles
    (csrc/host/meshes.h, subdivide_displace) — on the same ground plane under the same sky.  Not the bench workload:
    it checks that the statistics and the kernel ranking measured on the regular tube are not artefacts of its grid."""
    import os
    (tea, _, _), = R.Mesh.load_obj(os.path.join(ROOT, "assets", "teapot.obj"))
    mesh = tea.subdivided(5, 0.02, 1234)
    blas = renderer.add_blas(mesh)
    body = renderer.add_lambert_material((0.62, 0.5, 0.35))
    ground = renderer.add_lambert_material((0.6, 0.6, 0.6))
    renderer.add_blas_instance(blas, R.trs((0, 0.55, 0), 0.6, (0, 1, 0), (1.1, 1.1, 1.1)), body)
    renderer.add_blas_instance(renderer.plane_blas_index, R.trs((0, -0.02, 0), 0.0, (0, 1, 0), (14, 1, 14)), ground)
    cam = R.Camera((1.9, 1.5, 2.3), (0.0, 0.5, 0.0), (0, 1, 0), 40.0)
    return cam, mesh.triangle_count + 2


def build_instances_scene(R, renderer, n_side=45, seed=7):
    """C4: n_side^2 transformed instances of teapot / sphere / cube on a jittered grid with random rotations and
    scales, lambert / metal / dielectric / emissive materials, on a ground plane (2025 instances by default)."""
    import math
    import os
    rng = np.random.default_rng(seed)
    (tea, _, _), = R.Mesh.load_obj(os.path.join(ROOT, "assets", "teapot.obj"))
    r = renderer
    blases = [r.cube_blas_index, r.sphere_blas_index, r.add_blas(tea)]
    mats = [r.default_material, r.add_lambert_material((0.8, 0.3, 0.2)), r.add_metal_material((0.8, 0.8, 0.9), 0.05),
            r.add_metal_material((0.9, 0.6, 0.2), 0.4), r.add_dielectric_material(1.5), r.add_emissive_material((4.0, 3.5, 3.0))]
    r.add_blas_instance(r.plane_blas_index, R.trs((0, -0.5, 0), 0.0, (0, 1, 0), (120, 1, 120)), mats[1])
    k = 0
    for iz in range(n_side):
        for ix in range(n_side):
            p = ((ix - (n_side - 1) / 2) * 1.6 + rng.uniform(-0.2, 0.2), rng.uniform(0.0, 0.4),
                 (iz - (n_side - 1) / 2) * 1.6 + rng.uniform(-0.2, 0.2))
            axis = rng.normal(size=3)
            axis /= np.linalg.norm(axis)
            sc = rng.uniform(0.6, 1.1)
            r.add_blas_instance(blases[k % 3], R.trs(p, float(rng.uniform(0, 2 * math.pi)), tuple(axis), (sc, sc, sc)), mats[k % 6])
            k += 1
    return R.Camera((30, 22, 40), (0, 0, 0), (0, 1, 0), 45.0), r.total_triangle_count


def reference_layout_bytes_per_ray(cn, spp_per_launch):
    """SURVEY.md §8(d), the figure `roofline.achieved` is defined on — REFERENCE layouts: per interior visit two
    32-B child nodes; per TLAS leaf a 144-B instance + 16-B BLAS record; per triangle test 4-B tri id + 48-B
    TriangleGeom; per shaded hit 80-B TriangleShading + 144-B instance + 8-B material (lambert index + 1x1 texel);
    per pixel-sample the rgba32f read-modify-write amortised over the launch's samples."""
    rays = max(cn["rays"], 1)
    trav = 64 * cn["tlas_interior"] + 160 * cn["tlas_leaf"] + 64 * cn["blas_interior"] + 52 * cn["tris_tested"]
    shade = (80 + 144 + 8) * cn["shaded_hits"]
    return (trav + shade) / rays, 32.0 / spp_per_launch


def product_layout_bytes_per_ray(cn, spp_per_launch):
    """The bytes THIS library's kernels request for the same visits (rt_device_layout.h): 56 of a 64-B pair record
    per interior visit, a 64-B rt_dev_inst per TLAS leaf, a 48-B rt_dev_tri per triangle test, per shaded hit the
    176-B rt_dev_inst_shade + 80-B TriangleShading."""
    rays = max(cn["rays"], 1)
    trav = 56 * (cn["tlas_interior"] + cn["blas_interior"]) + 64 * cn["tlas_leaf"] + 48 * cn["tris_tested"]
    shade = (176 + 80) * cn["shaded_hits"]
    return (trav + shade) / rays, 32.0 / spp_per_launch


KERNEL_NAMES = {0: "k_render_v0", 8: "k_render_v8",
                9: "kernel 9: k_wf_trace in its persistent mode (every wave fills, traces and shades its window of path slots) + k_wf_resolve"}


def host_threads(requested):
    """Threads for the CPU baseline: every CPU this process may use.  On the GPU box os.cpu_count() shows the whole
    machine while the job owns a share of it (OMP_NUM_THREADS is set to that share), so the smaller wins."""
    if requested > 0:
        return requested
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    omp = os.environ.get("OMP_NUM_THREADS", "")
    if omp.isdigit() and int(omp) > 0:
        n = min(n, int(omp))
    return max(n, 1)


def git_commit():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL,
                                       text=True).strip()
    except Exception:
        return None


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
    ap.add_argument("--rng", choices=["wang", "pcg"], default="pcg")
    ap.add_argument("--progressive", action="store_true",
                    help="a step = --spp progressive frames of ONE sample per pixel in one rt_render_frames call (the reference's own "
                         "sampling, Renderer.cpp:21) instead of one frame of --spp stratified samples")
    ap.add_argument("--no-wang", action="store_true", help="N=1, --rng pcg: skip the extra measurement with the reference's RNG")
    ap.add_argument("--defocus", action="store_true", help="the C5 camera: thin lens instead of the pinhole")
    ap.add_argument("--scene", choices=["dragon", "teapot-dragon"], default="dragon",
                    help="dragon: the C3 workload (procedural tube, regular grid); teapot-dragon: the non-uniformly tessellated "
                         "second stand-in (not the bench workload)")
    ap.add_argument("--variant", type=lambda s: int(s, 0), default=0xFFFFFFFF,
                    help="kernel variant (default: RT_KERNEL_VARIANT_AUTO, the library's choice)")
    ap.add_argument("--shard", choices=["interleave", "frames", "rows"], default="interleave",
                    help="N>1: interleaved 8-row groups + overlapped all-gather (default, bit-identical to one GPU), "
                         "progressive frames round-robin + reduce, or contiguous row bands + all-gather")
    ap.add_argument("--no-frame1024", action="store_true", help="skip the measured 1024-spp frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the gather-ceiling probe (rt_debug_probe_walk)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on a box with fewer GPUs than ranks: ranks share the GPUs (local_rank %% device_count) and "
                         "talk over gloo instead of RCCL.  Exercises the multi-rank code path; its numbers mean nothing.")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every CPU this process may use")
    args = ap.parse_args()
    if os.environ.get("BENCH_DUMP_AFTER"):  # debugging aid: Python stacks of every thread if the run takes this long
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_AFTER"]), exit=True)

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
    rng_mode = T.RNG_PCG if args.rng == "pcg" else T.RNG_WANG
    aabb_mode = T.AABB_RCP if args.aabb == "rcp" else T.AABB_DIV

    t_setup = time.time()
    r = R.Renderer(W, H, device=local_rank)
    cam, n_tris = build_dragon_scene(R, r, defocus=args.defocus) if args.scene == "dragon" else build_teapot_dragon_scene(R, r)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1 if args.progressive else args.spp)
    r.set_option(R.OPT_MAX_DEPTH, args.depth)
    r.set_option(R.OPT_AABB_MODE, aabb_mode)
    r.set_option(R.OPT_RNG_MODE, rng_mode)
    bands, interleave = None, None
    mode = "none" if world == 1 else args.shard
    if mode == "rows":  # every rank renders its row band of every frame
        bands = [sharding.row_shard(k, world, H) for k in range(world)]
        r.set_option(R.OPT_ROW_BEGIN, bands[rank][0])
        r.set_option(R.OPT_ROW_END, bands[rank][1])
    elif mode == "interleave":  # every rank renders row groups rank, rank+N, ... of every frame
        interleave = sharding.row_interleave(rank, world, H)
        assert interleave is not None, "more ranks than 8-row groups"
        r.set_option(R.OPT_ROW_BEGIN, interleave["row_begin"])
        r.set_option(R.OPT_ROW_END, interleave["row_end"])
        r.set_option(R.OPT_ROW_GROUP_STRIDE, interleave["row_group_stride"])
    elif mode == "frames":
        shard = sharding.frame_shard(rank, world)
        r.set_option(R.OPT_FRAME_STRIDE, shard["frame_stride"])
        r.set_option(R.OPT_FRAME_OFFSET, shard["frame_offset"])
    r.set_option(R.OPT_KERNEL_VARIANT, args.variant)
    ctx = C.c_void_p(r.hip_context_handle())
    L = _ffi.lib()

    # the accumulation image lives in a torch tensor so RCCL can read it
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(accum.data_ptr()), W, H))
    reduced = torch.empty_like(accum) if mode in ("frames",) else None
    s_render = s_comm = gather = None
    if mode == "interleave":
        # launches go on a torch stream of ours, so the exchange can be ordered behind them with stream events
        s_render, s_comm = torch.cuda.Stream(), torch.cuda.Stream()
        _ffi._check(L.rt_set_stream(ctx, C.c_void_p(s_render.cuda_stream)))
        gather = sharding.InterleavedGather(H, W, device="cuda")
        comm_done = [None, None]
    torch.cuda.synchronize()  # torch filled these buffers on ITS stream; everything below runs on other streams
    setup_s = time.time() - t_setup

    def step():
        if args.progressive:
            r.render_frames(cam, args.spp)  # args.spp frames of one sample per pixel, one call
        else:
            r.render(cam)      # one launch: args.spp samples for every pixel of this rank's rows
        if mode == "interleave":
            with torch.cuda.stream(s_render):
                k = gather.step & 1
                if comm_done[k] is not None:
                    s_render.wait_event(comm_done[k])   # the packed buffer of two steps ago has been sent
                slot = gather.pack(accum)               # behind the launch, on its stream
                packed = s_render.record_event()
            with torch.cuda.stream(s_comm):
                s_comm.wait_event(packed)
                gather.exchange(slot)                   # every rank: the complete image, bit-exact; the next
                comm_done[slot] = s_comm.record_event()  # step's launch does not wait for this
        elif mode == "rows":
            r.wait()
            sharding.gather_rows(accum, bands)          # every rank: the complete image, bit-exact
            torch.cuda.current_stream().synchronize()
        elif mode == "frames":
            r.wait()
            sharding.combine(accum, out=reduced, dst=0)  # rank 0: (1/N) * sum of the N running means
            torch.cuda.current_stream().synchronize()

    def fence():
        r.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # With no kernel named, the library warms kernel 9 up, then times kernel 9 and kernel 8 once each on a new launch
    # shape and keeps the faster (rt_hip.h, rt_set_kernel_variant): three launches of set-up, like the BVH build, before the warmup
    settle = 3 if ((args.rng == "pcg" or args.progressive) and args.variant == _ffi.KERNEL_VARIANT_AUTO) else 0
    for _ in range(settle):
        step()
    fence()
    if settle:
        r.reset_accumulation()
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
    per_rank_kernel_ms = [st.total_kernel_ms / max(st.launches, 1)]
    if world > 1:
        tmax = vals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = vals.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, rays_total = float(tmax[0]), float(tsum[1])
        allv = [torch.zeros_like(vals) for _ in range(world)]
        dist.all_gather(allv, vals)
        per_rank_kernel_ms = [float(v[2]) / max(st.launches, 1) for v in allv]
    else:
        elapsed_max, rays_total = elapsed, rays_local

    # N=1, PCG: the same number of samples per pixel with the reference's own RNG (sequential Wang hash), in the
    # reference's own way — progressive frames of ONE sample per pixel (Renderer.cpp:21, 625; RayTracing.slang:71-72,
    # 176-179) — handed over as one batch (rt_render_frames: the (pixel, frame) pairs run side by side through the
    # wavefront pipeline), same timing rules; beside it the single frame of args.spp stratified samples, which the
    # sequential state confines to the megakernel
    wang = None
    if world == 1 and args.rng == "pcg" and not args.no_wang and not args.progressive:
        def timed(fn, settle_n):
            r.reset_accumulation()
            for _ in range(settle_n + args.warmup):
                fn()
            fence()
            r.reset_accumulation()
            _ffi._check(L.rt_reset_stats(ctx))
            t = time.perf_counter()
            for _ in range(args.steps):
                fn()
            fence()
            t = time.perf_counter() - t
            sw = _ffi.Stats()
            _ffi._check(L.rt_get_stats(ctx, C.byref(sw)))
            return {"value": sw.rays_traced / t / 1e6, "unit": "Mrays/s", "ms_per_step": t / args.steps * 1e3,
                    "kernel": KERNEL_NAMES.get(sw.last_kernel, "?"), "kernel_ms_per_launch": sw.total_kernel_ms / max(sw.launches, 1)}
        r.set_option(R.OPT_RNG_MODE, T.RNG_WANG)
        single = timed(step, 3 if args.variant == _ffi.KERNEL_VARIANT_AUTO else 0)  # (2e6 pixels: a measured shape too)
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1)
        wang = timed(lambda: r.render_frames(cam, args.spp), 3 if args.variant == _ffi.KERNEL_VARIANT_AUTO else 0)
        wang["note"] = (f"same scene, size and depth with rng_mode = RT_RNG_WANG (Random.slang:7-19): a step = {args.spp} progressive "
                        "frames of 1 sample per pixel (the reference's own sampling, Renderer.cpp:21) in one rt_render_frames call")
        single["note"] = (f"one frame of {args.spp} stratified samples per pixel, Wang RNG (a pixel's samples run one after the other): "
                          "one launch of the kernel the measured choice settled on")
        wang["single_frame"] = single
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, args.spp)
        r.set_option(R.OPT_RNG_MODE, rng_mode)
        r.reset_accumulation()

    # N>1 parity: rank 0 re-renders, on its own, a sample of row groups that OTHER ranks rendered, over all
    # warmup + timed frames, and compares them with the gathered image bit for bit
    parity = None
    frames_done = args.warmup + args.steps
    if mode == "interleave" and rank == 0 and frames_done > 0:
        groups = (H + 7) // 8
        sample = [g for g in range(groups) if g % world != 0]
        sample = sample[:: max(1, len(sample) // 5)][:5]
        chk = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the launches below on the library's
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(chk.data_ptr()), W, H))
        camu = r.camera_uniforms(cam)
        _ffi._check(L.rt_set_camera(ctx, C.byref(camu)))
        for f in range(frames_done):
            for g in sample:
                p = T.RenderParams()
                p.image_width, p.image_height, p.frame_index, p.accum_count = W, H, f, f
                p.sqrt_spp, p.max_depth, p.rng_mode, p.aabb_mode = sqrt_spp, args.depth, rng_mode, aabb_mode
                p.row_begin, p.row_end = g * 8, min(g * 8 + 8, H)
                _ffi._check(L.rt_render(ctx, C.byref(p)))
        r.wait()
        torch.cuda.synchronize()
        rows = [y for g in sample for y in range(g * 8, min(g * 8 + 8, H))]
        a = gather.full[rows].cpu().numpy().view(np.uint32)
        b = chk[rows].cpu().numpy().view(np.uint32)
        parity = {"pixels": len(rows) * W, "differing": int((a != b).any(-1).sum()), "frames": frames_done,
                  "bar": "gathered N-rank image == rank 0's own render of row groups other ranks rendered, bit-exact"}
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(accum.data_ptr()), W, H))

    # The metric's second half, "s/frame at 1024 spp", measured with one launch of 1024 samples per pixel per rank
    # (after the library's kernel trials for the new launch shape, which also size its buffers) instead of
    # extrapolated from the 256-spp steps; all ranks take part, the frame time is the slowest rank's
    frame1024 = None
    if mode in ("none", "interleave") and args.spp != 1024 and not args.no_frame1024 and not args.progressive:
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, 1024)
        r.reset_accumulation()
        for _ in range(max(settle, 1)):
            step()
        fence()
        r.reset_accumulation()
        _ffi._check(L.rt_reset_stats(ctx))
        tf = time.perf_counter()
        step()
        fence()
        tf = time.perf_counter() - tf
        sf = _ffi.Stats()
        _ffi._check(L.rt_get_stats(ctx, C.byref(sf)))
        fv = torch.tensor([tf, sf.total_kernel_ms / max(sf.launches, 1), float(sf.rays_traced)], dtype=torch.float64,
                          device="cpu" if args.rehearse else "cuda")
        if world > 1:
            fs = fv.clone()
            dist.all_reduce(fv, op=dist.ReduceOp.MAX)
            dist.all_reduce(fs, op=dist.ReduceOp.SUM)
            fv[2] = fs[2]
        frame1024 = {"s_per_frame": float(fv[0]), "value": float(fv[2]) / float(fv[0]) / 1e6, "unit": "Mrays/s",
                     "kernel": KERNEL_NAMES.get(sf.last_kernel, "?"), "kernel_ms_per_launch_slowest_rank": float(fv[1]),
                     "note": f"one launch of 1024 spp per rank, same scene / size / depth / rng ({args.rng}) / sharding; "
                             "not the bench value"}
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, args.spp)
        r.reset_accumulation()

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    kernel_ms = st.total_kernel_ms / max(st.launches, 1)
    rays_per_launch = rays_local / max(st.launches, 1)
    value = rays_total / elapsed_max / 1e6
    kernel_id = st.last_kernel  # read before the parity launches: what the timed launches ran

    cpu_baseline, roofline = None, None
    bytes_per_ray = product_bytes = None
    cnd = None
    if not args.no_cpu_baseline:
        # rank 0 only.  N = 1: the timed CPU baseline; N > 1: one oracle frame, untimed, for the algorithmic byte
        # counters the roofline needs (cpu_baseline stays null there, as the contract asks)
        from tests import _oracle as O
        sa = r.scene_arrays()                 # the product-built arrays, handed to the checker
        camu = r.camera_uniforms(cam)
        nthreads = host_threads(args.cpu_threads)
        cn = O.Counters()
        img = np.zeros((H, W, 4), np.float32)
        cpu_sqrt = 2
        frames, t_cpu = 0, 0.0
        while (t_cpu < args.cpu_seconds and frames < 64) if world == 1 else frames < 1:
            p = O.render_params(W, H, frames, cpu_sqrt, args.depth, rng_mode=rng_mode, aabb_mode=aabb_mode)
            t1 = time.perf_counter()
            O.render(sa, camu, p, img, counters=cn, nthreads=nthreads)
            t_cpu += time.perf_counter() - t1
            frames += 1
        cnd = cn.as_dict()
        if world == 1:
            cpu_baseline = {"value": cnd["rays"] / t_cpu / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "port",
                            "host_cpus": os.cpu_count(),
                            "sample": f"{frames} progressive frame(s) of {cpu_sqrt * cpu_sqrt} spp, full {W}x{H}, "
                                      f"depth {args.depth}, {cnd['rays']} rays in {t_cpu:.2f} s (oracle/liboracle.so, "
                                      f"{nthreads} threads = the CPUs this job may use; the machine shows "
                                      f"{os.cpu_count()})"}
        rays_per_sample = cnd["rays"] / (frames * cpu_sqrt * cpu_sqrt * W * H)
        trav_b, accum_b = reference_layout_bytes_per_ray(cnd, args.spp)
        bytes_per_ray = trav_b + accum_b / rays_per_sample
        trav_p, accum_p = product_layout_bytes_per_ray(cnd, args.spp)
        product_bytes = trav_p + accum_p / rays_per_sample
    if not args.no_cpu_baseline and world == 1 and not args.progressive:
        # Parity of THE KERNEL THAT WAS TIMED at THE SHAPE THAT WAS TIMED: frame 0 of the timed configuration (same
        # kernel — named, not left to the automatic choice — same spp per launch, same wavefront budget) rendered again,
        # and three 8-row groups of it (top, middle, bottom: 24 rows x W pixels x spp paths) compared bit for bit with
        # the oracle at the same spp; then the whole frame at 4 spp (the megakernel's shape) against the oracle.
        chk = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the launches below on the library's
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(chk.data_ptr()), W, H))
        _ffi._check(L.rt_set_camera(ctx, C.byref(camu)))
        _ffi._check(L.rt_set_kernel_variant(ctx, kernel_id))
        pt = O.render_params(W, H, 0, sqrt_spp, args.depth, rng_mode=rng_mode, aabb_mode=aabb_mode)
        _ffi._check(L.rt_render(ctx, C.byref(pt)))
        r.wait()
        sp = _ffi.Stats()
        _ffi._check(L.rt_get_stats(ctx, C.byref(sp)))
        got = chk.cpu().numpy()
        stride = max(1, ((H + 7) // 8 - 1) // 2)
        rows = sharding.interleaved_rows(0, stride, H)
        want = np.zeros((H, W, 4), np.float32)
        po = O.render_params(W, H, 0, sqrt_spp, args.depth, rng_mode=rng_mode, aabb_mode=aabb_mode,
                             row_begin=0, row_end=H, row_group_stride=stride)
        O.render(sa, camu, po, want, nthreads=nthreads)
        bad = int((got[rows].view(np.uint32) != want[rows].view(np.uint32)).any(-1).sum())
        parity = {"kernel": KERNEL_NAMES.get(sp.last_kernel, "?"), "spp": args.spp, "rows": len(rows), "pixels": len(rows) * W,
                  "paths": len(rows) * W * args.spp, "differing": bad,
                  "bar": "bit-exact rgba32f vs oracle: the timed kernel at the timed launch shape (frame 0 rendered again, "
                         "three 8-row groups compared)"}
        _ffi._check(L.rt_set_kernel_variant(ctx, args.variant))
        chk.zero_()
        torch.cuda.synchronize()
        p0 = O.render_params(W, H, 0, cpu_sqrt, args.depth, rng_mode=rng_mode, aabb_mode=aabb_mode)
        _ffi._check(L.rt_render(ctx, C.byref(p0)))
        r.wait()
        sp0 = _ffi.Stats()
        _ffi._check(L.rt_get_stats(ctx, C.byref(sp0)))
        want0 = O.render(sa, camu, p0, nthreads=nthreads)
        got0 = chk.cpu().numpy()
        parity["full_frame_4spp"] = {"kernel": KERNEL_NAMES.get(sp0.last_kernel, "?"), "pixels": W * H,
                                     "differing": int((got0.view(np.uint32) != want0.view(np.uint32)).any(-1).sum())}
        _ffi._check(L.rt_bind_accum(ctx, C.c_void_p(accum.data_ptr()), W, H))
    if bytes_per_ray is not None:
        # The roofs ("Roofline"): the dominant kernel is the traversal kernel — k_wf_trace of the wavefront
        # pipeline, or the megakernel.  Its duration is measured HERE with HIP events on the launch stream; the
        # instruction and byte counts of one launch come from the committed PMC passes of the same workload
        # (profiles/roofline_latest.json, tools/roofline_from_pmc.py — PMC needs rocprofv3 runs of their own), and
        # only for that workload.  Every fraction is against a peak of MI355X_MICROARCH.md.
        VALU_PEAK = 256 * 4 * 2.4 / 2.0   # G wave-instr/s: 256 CUs x 4 SIMDs x 2.4 GHz, one wave64 VALU instruction per 2 cycles
        FABRIC_GATHER_PEAK = 7650.0       # GB/s: random rows of a 151 MB table (Infinity Cache resident), guide "Indexed rows": 7.4-7.9 TB/s
        algorithmic = bytes_per_ray * rays_per_launch / (kernel_ms * 1e-3) / 1e9
        per_kernel = None
        if world == 1 and kernel_id == 9:
            ms5, nl = (C.c_double * 5)(), C.c_uint64()
            _ffi._check(L.rt_debug_wavefront_timing(ctx, 1, None, None))
            _ffi._check(L.rt_reset_stats(ctx))
            _ffi._check(L.rt_set_kernel_variant(ctx, 9))
            for _ in range(2):
                step()
            fence()
            _ffi._check(L.rt_debug_wavefront_timing(ctx, 0, ms5, C.byref(nl)))
            _ffi._check(L.rt_set_kernel_variant(ctx, args.variant))
            n = max(nl.value, 1)
            per_kernel = {"k_wf_trace": ms5[0] / n, "k_wf_shade": ms5[1] / n, "k_wf_finish": ms5[2] / n,
                          "k_wf_resolve": ms5[3] / n, "first_fill": ms5[4] / n, "launches": nl.value,
                          "how": "HIP events on the launch stream around every dispatch of two extra launches (rt_debug_wavefront_timing)"}
        label = args.rng + ("-frames" if args.progressive else "")  # the profile's name for this sampling
        cdoc, ckey = None, f"{label}:{'k_wf_trace' if kernel_id == 9 else 'k_render_v8'}"
        cpath = os.path.join(ROOT, "profiles", "roofline_latest.json")
        if world == 1 and os.path.exists(cpath) and \
                (W, H, args.spp, args.depth, args.aabb, args.defocus, args.scene) == (1920, 1080, 256, 8, "div", False, "dragon"):
            try:
                cdoc = json.load(open(cpath))
                if ckey not in cdoc["kernels"]:
                    cdoc = None
            except Exception:
                cdoc = None
        trace_ms = per_kernel["k_wf_trace"] if per_kernel else kernel_ms
        roofline = {"kernel": "k_wf_trace" if kernel_id == 9 else KERNEL_NAMES.get(kernel_id, "?"),
                    "kernel_ms": trace_ms, "launch_ms": kernel_ms, "kernel_ms_by_kernel": per_kernel,
                    "rays_per_launch": rays_per_launch}
        if cdoc is not None:
            k = cdoc["kernels"][ckey]
            valu = k["valu_wave_insts"] / (trace_ms * 1e-3) / 1e9
            roofline.update({
                "bound": "valu", "achieved": valu, "peak": VALU_PEAK, "unit": "G wave-instr/s", "frac": valu / VALU_PEAK,
                "peak_source": "MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 VALU instruction issues over 2 cycles",
                "lane_utilisation": k["lane_utilisation"], "lane_weighted_frac": valu / VALU_PEAK * k["lane_utilisation"],
                "wait_any_over_wave_cycles": k["wait_any_over_wave_cycles"], "l2_hit_rate": k["l2_hit_rate"],
                "fabric_gather": {
                    "what": "L2-miss reads of the traversal kernel (FETCH_SIZE; includes Infinity-Cache hits: the 199 MB scene "
                            "lives there) against the guide's rate for random rows of a cache-resident table",
                    "achieved_GBps_raw": k["fetch_bytes_raw"] / (trace_ms * 1e-3) / 1e9,
                    "achieved_GBps_x2": 2 * k["fetch_bytes_raw"] / (trace_ms * 1e-3) / 1e9,
                    "peak": FABRIC_GATHER_PEAK, "frac_raw": k["fetch_bytes_raw"] / (trace_ms * 1e-3) / 1e9 / FABRIC_GATHER_PEAK,
                    "frac_x2": 2 * k["fetch_bytes_raw"] / (trace_ms * 1e-3) / 1e9 / FABRIC_GATHER_PEAK,
                    "note": "FETCH_SIZE tallies 128-byte requests as 64 (guide, HBM section): raw is exact for 64-byte requests, x2 "
                            "for wide streams; 16-byte divergent gathers are uncalibrated, the truth lies between"}})
            # `traffic`: every dispatch of the launch; what these counters are is said beside it
            tot_fetch = sum(v["fetch_bytes_raw"] for kk, v in cdoc["kernels"].items() if kk.startswith(label + ":") and
                            (v["kernel"].startswith("k_wf_") if kernel_id == 9 else v["kernel"] == "k_render_v8"))
            tot_write = sum(v["write_bytes"] for kk, v in cdoc["kernels"].items() if kk.startswith(label + ":") and
                            (v["kernel"].startswith("k_wf_") if kernel_id == 9 else v["kernel"] == "k_render_v8"))
            roofline["traffic"] = tot_fetch + tot_write
            roofline["traffic_is"] = ("L2-miss (fabric-side) bytes of one launch, all its dispatches: FETCH_SIZE raw + WRITE_SIZE. "
                                      "NOT HBM bytes — Infinity-Cache hits are counted (guide, HBM section)")
            roofline["counter_source"] = (f"profiles/roofline_latest.json ({k['pmc_dir']}, rocprofv3 --pmc passes of tools/profile_pmc.sh "
                                          f"on this workload, commit {cdoc.get('commit')}); durations measured in this run")
            sh = cdoc["kernels"].get(f"{label}:k_wf_shade")
            if kernel_id == 9 and sh and per_kernel and per_kernel["k_wf_shade"] > 0:  # (the rounds mode: a shade kernel of its own)
                # the shade pass streams the path state (wide coalesced 16-byte accesses: FETCH_SIZE x2 is the calibrated reading)
                hb = (2 * sh["fetch_bytes_raw"] + sh["write_bytes"]) / (per_kernel["k_wf_shade"] * 1e-3) / 1e9
                roofline["secondary"] = {"kernel": "k_wf_shade", "bound": "hbm", "kernel_ms": per_kernel["k_wf_shade"],
                                         "achieved": hb, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBPS,
                                         "bytes_per_launch": 2 * sh["fetch_bytes_raw"] + sh["write_bytes"],
                                         "note": "5.9 GB of path state streams through HBM every round; 6.29 TB/s is the guide's measured copy rate"}
        else:
            roofline.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                             "note": "no committed PMC counters for this workload / kernel: see gather_probe and algorithmic_GBps"})
        roofline["algorithmic_bytes_per_ray"] = bytes_per_ray
        roofline["algorithmic_GBps"] = algorithmic
        roofline["algorithmic_note"] = ("SURVEY.md 8(d): bytes counted with the REFERENCE's record sizes for the visits the oracle counts, over "
                                        "the whole launch; most are L1 / L2 / Infinity-Cache hits, so this is no bandwidth of any one level "
                                        "and carries no fraction")
        roofline["requested_bytes_per_ray"] = product_bytes
        roofline["requested_GBps"] = product_bytes * rays_per_launch / (kernel_ms * 1e-3) / 1e9
        if not args.no_probe and world == 1 and cnd is not None:
            # the L1 / texture-path ceiling for this access pattern, measured in this process: dependent divergent 64-B
            # record gathers over this scene's node records, no arithmetic, at the probe's best occupancy (4 waves per SIMD)
            best = 0.0
            for _ in range(3):
                ms, steps = C.c_double(), C.c_uint64()
                _ffi._check(L.rt_debug_probe_walk(ctx, 256 * 4, 2000, 4, C.byref(ms), C.byref(steps)))
                best = max(best, steps.value / (ms.value * 1e-3))
            visits_per_ray = (cnd["tlas_interior"] + cnd["blas_interior"] + cnd["tlas_leaf"] + cnd["tris_tested"]) / max(cnd["rays"], 1)
            kernel_rate = visits_per_ray * rays_per_launch / (trace_ms * 1e-3)
            if cdoc is not None and "l1_accesses" in cdoc["kernels"][ckey]:
                # The per-lane L1 access rate ("The third roof"): a per-lane 16-byte load is one L1 (TCP)
                # access, and a CU serves about one per clock — the probe reaches 0.93 per clock and CU with 2, 3 or 4 loads
                # per step alike (tools/probe_loads.py).  The kernel runs NEAR that rate but is not bound by it: taking 28 %
                # of its accesses away (the top records in LDS) bought 2-3 % (experiments/r03_lds_cache/).  Reported as a
                # utilisation, not as the binding roof.  No figure for this in the guide: the nominal peak is 1 access per
                # clock and CU at 2.4 GHz, the measured one the probe's.
                acc = cdoc["kernels"][ckey]["l1_accesses"] / (trace_ms * 1e-3) / 1e9
                roofline["l1_access"] = {"achieved": acc, "unit": "G L1 accesses/s (TCP_TOTAL_CACHE_ACCESSES of one launch / the kernel's time)",
                                         "nominal_peak": 256 * 2.4, "frac_nominal": acc / (256 * 2.4),
                                         "probe_ceiling": 4 * best / 1e9, "frac_probe": acc / (4 * best / 1e9),
                                         "note": "nominal = 256 CUs x 1 access per clock x 2.4 GHz (an assumption, not a guide figure); "
                                                 "probe_ceiling = 4 accesses per step x the probe's step rate, measured in this run; "
                                                 "a utilisation, not the binding roof (experiments/r03_lds_cache/)"}
            roofline["gather_probe"] = {"visits_per_s": best, "kernel_visits_per_s": kernel_rate,
                                        "frac": kernel_rate / best, "record_visits_per_ray": visits_per_ray,
                                        "how": "rt_debug_probe_walk: 1024 blocks x 256 lanes, 2000 dependent 64-B record gathers per "
                                               "lane over this scene's node records, best of 3; the kernel's visits over the traversal "
                                               "kernel's own duration.  A measured yardstick of the per-lane gather path, not a guide peak"}

    sharding_txt = {"none": "none", "interleave": "interleaved 8-row groups per rank, RCCL all-gather per step overlapped "
                    "with the next launch", "rows": "row bands per rank, RCCL all-gather per step",
                    "frames": "progressive frames round-robin over ranks, RCCL reduce per step"}[mode]
    strong = args.shard in ("interleave", "rows")
    out = {
        "metric": "Mrays/s (incl. bounces) dragon scene", "value": value, "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",  # N = 1 reports what --shard would do at N > 1 (a step is one frame whatever N)
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.rehearse else "synthetic (REHEARSAL: shared GPU, gloo)",
        "config": {"workload": ("C3 dragon-class stand-in (procedural, seed 1234)" if args.scene == "dragon" else
                                "C3b subdivided + displaced teapot.obj (non-uniform tessellation, seed 1234; NOT the bench workload)") +
                               (", C5 thin-lens camera" if args.defocus else ""),
                   "triangles": n_tris, "image": f"{W}x{H}", "spp_per_step": args.spp, "max_depth": args.depth,
                   "sampling": f"{args.spp} progressive frames of 1 spp per step (rt_render_frames)" if args.progressive else f"one frame of {args.spp} stratified spp per step",
                   "rng": args.rng, "aabb": args.aabb, "kernel": KERNEL_NAMES.get(kernel_id, "?"), "sharding": sharding_txt},
        "ranks_seen": dist.get_world_size() if world > 1 else 1,
        "kernel_ms_per_launch_by_rank": per_rank_kernel_ms,
        "s_per_frame_1024spp": elapsed_max / args.steps * (1024.0 / (args.spp * (1 if strong or world == 1 else world))),
        "rays_per_step": rays_total / args.steps, "kernel_ms_per_launch": kernel_ms,
        "setup_s": setup_s, "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_check": parity, "wang_rng": wang, "frame_1024spp_measured": frame1024, "kernel_choice_launches": settle,
        "commit": git_commit(),
    }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
