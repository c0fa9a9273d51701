"""The native multi-GPU driver (include/rt_hip.h rt_mgpu_*, raytracing_amd/csrc/rt_multi.cpp: one rt_ctx and one host
thread per rank, interleaved 8-row groups, one image exchange per launch) on the ONE GPU a test box has: several ranks
on device 0 exchange by peer copies (RCCL cannot put two ranks on one device), one rank alone goes through
ncclAllGather with a world of one.  Bar: every rank's image equals the oracle's whole image, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import _types as T
from raytracing_amd import renderer as R
from . import _oracle as O
from . import _scenes as S

pytestmark = pytest.mark.gpu

CLI = os.path.join(os.path.dirname(os.path.abspath(R.__file__)), "bin", "renderer")
SCENES = os.path.join(S.ASSETS, "scenes")


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_on_one_gpu_stitch_the_oracle_image(ranks):
    sc = S.mixed_scene().build()
    w, h, depth = 200, 104, 6
    cam = S.mixed_camera(w, h)
    m = _ffi.MultiGpu([0] * ranks)
    assert not m.uses_rccl
    m.upload_scene(sc)
    m.set_camera(cam)
    m.resize(w, h)
    for rng_mode, sqrt_spp, variant in ((T.RNG_WANG, 2, 8), (T.RNG_PCG, 2, 9), (T.RNG_WANG, 1, 9), (T.RNG_PCG, 3, _ffi.KERNEL_VARIANT_AUTO)):
        want = np.zeros((h, w, 4), np.float32)
        for f in range(5):
            O.render(sc, cam, O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng_mode), want, nthreads=16)
        m.set_kernel_variant(variant)
        m.clear_accum()
        m.render_frames(O.render_params(w, h, 0, sqrt_spp, depth, rng_mode=rng_mode), 2)  # frames 0, 1 in one batch
        for f in (2, 3, 4):                                                                # then one by one
            m.render_frames(O.render_params(w, h, f, sqrt_spp, depth, rng_mode=rng_mode), 1)
        for rank in range(ranks):
            assert _same(m.read_accum(rank, w, h), want), (rng_mode, sqrt_spp, variant, rank)
    # every rank did its share of the rays
    rays = [m.stats(k).rays_traced for k in range(ranks)]
    assert all(r > 0 for r in rays) and max(rays) < 2 * min(rays)
    m.close()


def test_one_rank_through_rccl_all_gather():
    """A world of one: the pack -> ncclAllGather -> unpack path of the exchange, on the library's own dlopen'ed RCCL."""
    sc = S.cornell_builtin().build()
    w, h = 96, 56
    cam = S.cornell_camera(w, h)
    m = _ffi.MultiGpu([0], _ffi.EXCHANGE_RCCL)
    assert m.uses_rccl
    m.upload_scene(sc)
    m.set_camera(cam)
    m.resize(w, h)
    p = O.render_params(w, h, 0, 2, 3)
    m.render_frames(p, 1)
    assert _same(m.read_accum(0, w, h), O.render(sc, cam, p, nthreads=8))
    m.close()
    with pytest.raises(_ffi.RtError, match="share a device"):
        _ffi.MultiGpu([0, 0], _ffi.EXCHANGE_RCCL)


def test_instance_updates_reach_every_rank():
    o = S.mixed_scene()
    sc = o.build()
    w, h = 128, 64
    cam = S.mixed_camera(w, h)
    m = _ffi.MultiGpu([0, 0])
    m.upload_scene(sc)
    m.set_camera(cam)
    m.resize(w, h)
    o.set_blas_instance_transform(0, O.trs((0.3, 0.4, -0.2), O.angle_axis(0.5, (0, 1, 0)), (1.2, 0.8, 1.0)))
    sc2 = o.build()
    m.update_instances(sc2.instances, sc2.tlas_nodes, sc2.tlas_node_count)
    p = O.render_params(w, h, 0, 2, 5)
    m.render_frames(p, 1)
    want = O.render(sc2, cam, p, nthreads=8)
    assert _same(m.read_accum(0, w, h), want) and _same(m.read_accum(1, w, h), want)
    m.close()


def test_cli_multi_rank_image_equals_the_single_context_image(tmp_path):
    """`renderer <scene> -hip --devices 0,0,0` (three ranks) and `--gpus 1` against the plain command line, and the new
    switches --rng pcg / --frames / --kernel."""
    scene = os.path.join(SCENES, "multi.json")
    def run(name, *extra):
        raw = str(tmp_path / (name + ".f32"))
        p = subprocess.run([CLI, scene, "-hip", "-o", str(tmp_path / (name + ".png")), "--raw", raw, "--width", "160", "--spp", "4",
                            "--depth", "5", *extra], capture_output=True, text=True, timeout=180)
        assert p.returncode == 0, p.stderr
        return np.fromfile(raw, np.float32), p.stderr
    base, _ = run("base")
    one, _ = run("one", "--gpus", "1")
    three, log = run("three", "--devices", "0,0,0")
    assert "3 ranks" in log and "rank 2" in log
    assert np.array_equal(base.view(np.uint32), one.view(np.uint32)) and np.array_equal(base.view(np.uint32), three.view(np.uint32))
    pcg8, _ = run("pcg8", "--rng", "pcg", "--frames", "3", "--kernel", "8")
    pcg9, log9 = run("pcg9", "--rng", "pcg", "--frames", "3", "--kernel", "9", "--devices", "0,0")
    assert "kernel 9" in log9
    assert np.array_equal(pcg8.view(np.uint32), pcg9.view(np.uint32)) and not np.array_equal(pcg8.view(np.uint32), base.view(np.uint32))
