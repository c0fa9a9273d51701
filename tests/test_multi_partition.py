"""The bookkeeping of the native multi-GPU driver (rt_mgpu_partition, raytracing_amd/csrc/rt_multi.cpp; SURVEY.md §8(e):
interleaved 8-row groups) — callable without a GPU — against the Python partition the torch.distributed bench uses
(raytracing_amd/sharding.py), which the gloo tests of tests/test_sharding.py cover."""
import numpy as np
import pytest

from raytracing_amd import _ffi
from raytracing_amd import sharding


@pytest.mark.parametrize("height", [8, 9, 64, 135, 1080, 2160, 1081, 17])
@pytest.mark.parametrize("ranks", [1, 2, 3, 4, 8])
def test_partition_covers_the_image_once_and_equals_the_python_partition(height, ranks):
    groups = (height + 7) // 8
    if groups < ranks:
        pytest.skip("fewer row groups than ranks")
    seen = np.zeros(height, np.int32)
    for rank in range(ranks):
        sh = _ffi.mgpu_partition(height, ranks, rank)
        want = sharding.row_interleave(rank, ranks, height)
        assert (sh.row_begin, sh.row_end, sh.row_group_stride) == (want["row_begin"], want["row_end"], want["row_group_stride"])
        rows = sharding.interleaved_rows(rank, ranks, height)
        assert sh.rows == len(rows) and sh.groups == len(range(rank, groups, ranks))
        assert sh.groups_total == groups and sh.groups_per_rank == -(-groups // ranks) and sh.groups <= sh.groups_per_rank
        seen[rows] += 1
    assert (seen == 1).all()


def test_partition_rejects_bad_arguments():
    sh = _ffi.MgpuShare()
    import ctypes as C
    L = _ffi.lib()
    assert L.rt_mgpu_partition(0, 1, 0, C.byref(sh)) < 0
    assert L.rt_mgpu_partition(64, 0, 0, C.byref(sh)) < 0
    assert L.rt_mgpu_partition(64, 2, 2, C.byref(sh)) < 0
    assert L.rt_mgpu_partition(64, 2, 1, None) < 0
