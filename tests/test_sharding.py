"""N>1 path: frame sharding + one sum-reduce (raytracing_amd/sharding.py).
CPU: two gloo ranks, each rendering its frames with the oracle as stand-in for
the GPU (the collective and the bookkeeping are what is under test).  GPU: two
renderers on one device play two ranks through the product path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import _types as T
from raytracing_amd import sharding
from . import _oracle as O
from . import _scenes as S

W, H, SQRT_SPP, DEPTH, FRAMES = 40, 24, 2, 4, 4


def _local_mean(sc, cam, rank, world):
    img = np.zeros((H, W, 4), np.float32)
    for k, f in enumerate(sharding.local_frames(rank, world, FRAMES)):
        O.render(sc, cam, O.render_params(W, H, f, SQRT_SPP, DEPTH, accum_count=k), img)
    return img


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = S.cornell_builtin().build()
        cam = S.cornell_camera(W, H)
        local = torch.from_numpy(_local_mean(sc, cam, rank, world))
        keep = local.clone()
        res = sharding.combine(local, dst=0)
        assert torch.equal(local, keep)  # the running mean is not disturbed
        if rank == 0:
            np.save(out_path, res.numpy())
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_combine(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "combined.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc = S.cornell_builtin().build()
    cam = S.cornell_camera(W, H)
    m0, m1 = _local_mean(sc, cam, 0, 2), _local_mean(sc, cam, 1, 2)
    want = (m0 + m1) * np.float32(0.5)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # equals the single-process progressive image up to fp32 summation order
    single = np.zeros((H, W, 4), np.float32)
    for f in range(FRAMES):
        O.render(sc, cam, O.render_params(W, H, f, SQRT_SPP, DEPTH), single)
    assert np.allclose(got, single, rtol=2e-6, atol=2e-6)
    assert (got[..., 3] == 1).all()


def _row_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = S.cornell_builtin().build()
        cam = S.cornell_camera(W, H)
        bands = [sharding.row_shard(r, world, H) for r in range(world)]
        img = np.zeros((H, W, 4), np.float32)
        for f in range(2):  # this rank's band of every frame (the oracle stands in for the GPU)
            p = O.render_params(W, H, f, SQRT_SPP, DEPTH)
            p.row_begin, p.row_end = bands[rank]
            O.render(sc, cam, p, img)
        full = sharding.gather_rows(torch.from_numpy(img), bands)
        np.save(out_path % rank, full.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_row_bands_equal_single_rank_bitwise(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rows_%d.npy")
    mp.spawn(_row_worker, args=(2, port, out), nprocs=2, join=True)
    sc = S.cornell_builtin().build()
    cam = S.cornell_camera(W, H)
    single = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        O.render(sc, cam, O.render_params(W, H, f, SQRT_SPP, DEPTH), single)
    for r in range(2):  # every rank ends up with the complete image, bit for bit the 1-rank image
        assert np.array_equal(np.load(out % r).view(np.uint32), single.view(np.uint32))


def test_row_shard_bookkeeping():
    for h in (1, 7, 8, 9, 24, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            bands = [sharding.row_shard(r, n, h) for r in range(n)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(n - 1))
            assert all(b % 8 == 0 for b, _ in bands)
    assert [sharding.row_shard(r, 8, 1080) for r in range(8)][3] == (400, 536)
    with pytest.raises(ValueError):
        sharding.row_shard(4, 4, 100)
    x = torch.ones(4, 2, 4)
    assert sharding.gather_rows(x, [(0, 4)]) is x  # no process group: nothing to gather


def test_frame_shard_bookkeeping():
    assert sharding.frame_shard(3, 8) == {"frame_stride": 8, "frame_offset": 3}
    assert sharding.local_frames(1, 4, 10) == [1, 5, 9]
    assert sorted(sum((sharding.local_frames(r, 4, 10) for r in range(4)), [])) == list(range(10))
    with pytest.raises(ValueError):
        sharding.frame_shard(2, 2)
    x = torch.ones(3)
    assert torch.equal(sharding.combine(x), x)  # no process group: identity copy


@pytest.mark.gpu
def test_frame_sharding_through_renderer_on_one_gpu():
    from raytracing_amd import renderer as R
    from .test_host_parity import _cornell_via_product
    cam = R.Camera((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0)
    means = []
    for rank in range(2):
        r = _cornell_via_product(device=0, w=W, h=H)
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, SQRT_SPP * SQRT_SPP)
        r.set_option(R.OPT_MAX_DEPTH, DEPTH)
        sh = sharding.frame_shard(rank, 2)
        r.set_option(R.OPT_FRAME_STRIDE, sh["frame_stride"])
        r.set_option(R.OPT_FRAME_OFFSET, sh["frame_offset"])
        for _ in range(FRAMES // 2):
            r.render(cam)
        means.append(r.read_image())
        sa, camu = r.scene_arrays(), r.camera_uniforms(cam)
        want = np.zeros((H, W, 4), np.float32)
        for k, f in enumerate(sharding.local_frames(rank, 2, FRAMES)):
            O.render(sa, camu, O.render_params(W, H, f, SQRT_SPP, DEPTH, accum_count=k), want)
        assert np.array_equal(means[-1].view(np.uint32), want.view(np.uint32))
        r.shutdown()
    assert not np.array_equal(means[0], means[1])


@pytest.mark.gpu
def test_row_sharding_through_renderer_on_one_gpu():
    """Two renderers play two ranks of the row-band mode: the stitched image is the
    single-renderer image bit for bit."""
    from raytracing_amd import renderer as R
    from .test_host_parity import _cornell_via_product
    cam = R.Camera((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0)
    bands = [sharding.row_shard(k, 2, H) for k in range(2)]
    stitched = np.zeros((H, W, 4), np.float32)
    for rank in range(2):
        r = _cornell_via_product(device=0, w=W, h=H)
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, SQRT_SPP * SQRT_SPP)
        r.set_option(R.OPT_MAX_DEPTH, DEPTH)
        r.set_option(R.OPT_ROW_BEGIN, bands[rank][0])
        r.set_option(R.OPT_ROW_END, bands[rank][1])
        for _ in range(2):
            r.render(cam)
        img = r.read_image()
        b, e = bands[rank]
        stitched[b:e] = img[b:e]
        assert (np.delete(img, np.s_[b:e], axis=0) == 0).all()  # rows outside the band are untouched
        r.shutdown()
    r = _cornell_via_product(device=0, w=W, h=H)
    r.set_option(R.OPT_SAMPLES_PER_PIXEL, SQRT_SPP * SQRT_SPP)
    r.set_option(R.OPT_MAX_DEPTH, DEPTH)
    for _ in range(2):
        r.render(cam)
    single = r.read_image()
    r.shutdown()
    assert np.array_equal(stitched.view(np.uint32), single.view(np.uint32))


# ---- interleaved 8-row groups ("row-interleaved tiles round-robin over GPUs") -----------------------
HI = 44  # 5 full groups + a partial one: ranks get 3 / 3 groups of unequal size


def _interleave_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = S.cornell_builtin().build()
        cam = S.cornell_camera(W, HI)
        opt = sharding.row_interleave(rank, world, HI)
        img = np.zeros((HI, W, 4), np.float32)
        for f in range(2):  # this rank's row groups of every frame (the oracle stands in for the GPU)
            O.render(sc, cam, O.render_params(W, HI, f, SQRT_SPP, DEPTH, **opt), img)
        mine = sharding.interleaved_rows(rank, world, HI)
        assert (np.delete(img, mine, axis=0) == 0).all() and (img[mine][..., 3] == 1).all()
        full = sharding.gather_interleaved(torch.from_numpy(img))
        np.save(out_path % rank, full.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_interleaved_rows_equal_single_rank_bitwise(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "inter_%d.npy")
    mp.spawn(_interleave_worker, args=(2, port, out), nprocs=2, join=True)
    sc = S.cornell_builtin().build()
    cam = S.cornell_camera(W, HI)
    single = np.zeros((HI, W, 4), np.float32)
    for f in range(2):
        O.render(sc, cam, O.render_params(W, HI, f, SQRT_SPP, DEPTH), single)
    for r in range(2):
        assert np.array_equal(np.load(out % r).view(np.uint32), single.view(np.uint32))


def _interleaved_gather_worker(rank, world, port, out_path, height):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = S.cornell_builtin().build()
        cam = S.cornell_camera(W, height)
        opt = sharding.row_interleave(rank, world, height)
        ig = sharding.InterleavedGather(height, W)
        img = np.zeros((height, W, 4), np.float32)
        acc = torch.from_numpy(img)  # shares memory: the oracle accumulates in place like the kernel
        fulls = []
        for f in range(3):  # three progressive frames, one exchange each, alternating packed buffers
            if opt is not None:
                O.render(sc, cam, O.render_params(W, height, f, SQRT_SPP, DEPTH, **opt), img)
            slot = ig.pack(acc)
            fulls.append(ig.exchange(slot).clone())
        assert (img[[r for r in range(height) if r not in set(sharding.interleaved_rows(rank, world, height))]] == 0).all()
        np.save(out_path % rank, torch.stack(fulls).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, HI), (3, 20), (2, 7)])
def test_gloo_interleaved_gather_class_equals_single_rank_bitwise(tmp_path, world, height):
    """sharding.InterleavedGather (the overlapped exchange of bench.py --gpus N): after every step every rank holds
    the one-rank progressive image bit for bit — ragged group counts and ranks without rows included."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ig_%d.npy")
    mp.spawn(_interleaved_gather_worker, args=(world, port, out, height), nprocs=world, join=True)
    sc = S.cornell_builtin().build()
    cam = S.cornell_camera(W, height)
    single = np.zeros((height, W, 4), np.float32)
    for f in range(3):
        O.render(sc, cam, O.render_params(W, height, f, SQRT_SPP, DEPTH), single)
        for r in range(world):
            assert np.array_equal(np.load(out % r)[f].view(np.uint32), single.view(np.uint32)), (f, r)


def test_row_interleave_bookkeeping():
    for h in (1, 7, 8, 9, 24, 44, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            rows = [sharding.interleaved_rows(r, n, h) for r in range(n)]
            assert sorted(sum(rows, [])) == list(range(h))
            for r in range(n):
                opt = sharding.row_interleave(r, n, h)
                assert (opt is None) == (len(rows[r]) == 0)
                if opt:
                    assert opt == {"row_begin": 8 * r, "row_end": h, "row_group_stride": n}
    assert sharding.interleaved_rows(1, 2, 20) == [8, 9, 10, 11, 12, 13, 14, 15]
    with pytest.raises(ValueError):
        sharding.row_interleave(2, 2, 100)
    x = torch.ones(4, 2, 4)
    assert sharding.gather_interleaved(x) is x  # no process group: nothing to gather


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 8])
def test_interleaved_rows_through_renderer_on_one_gpu(variant):
    """Three renderers play three ranks of the interleaved mode: the stitched image is the single-renderer image
    bit for bit, for every kernel family (they share the row mapping of rt_kernels.h)."""
    from raytracing_amd import renderer as R
    from .test_host_parity import _cornell_via_product
    cam = R.Camera((0, 0.75, 1.5), (0, 0.75, -1), (0, 1, 0), 90.0)
    world = 3

    def make():
        r = _cornell_via_product(device=0, w=W, h=HI)
        r.set_option(R.OPT_SAMPLES_PER_PIXEL, SQRT_SPP * SQRT_SPP)
        r.set_option(R.OPT_MAX_DEPTH, DEPTH)
        r.set_option(R.OPT_KERNEL_VARIANT, variant)
        return r

    stitched = np.zeros((HI, W, 4), np.float32)
    for rank in range(world):
        r = make()
        opt = sharding.row_interleave(rank, world, HI)
        r.set_option(R.OPT_ROW_BEGIN, opt["row_begin"])
        r.set_option(R.OPT_ROW_END, opt["row_end"])
        r.set_option(R.OPT_ROW_GROUP_STRIDE, opt["row_group_stride"])
        for _ in range(2):
            r.render(cam)
        img = r.read_image()
        mine = sharding.interleaved_rows(rank, world, HI)
        stitched[mine] = img[mine]
        assert (np.delete(img, mine, axis=0) == 0).all()  # rows of the other ranks are untouched
        r.shutdown()
    r = make()
    for _ in range(2):
        r.render(cam)
    single = r.read_image()
    r.shutdown()
    assert np.array_equal(stitched.view(np.uint32), single.view(np.uint32))
