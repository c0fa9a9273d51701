"""Multi-GPU sharding of the render path: one process per GPU, scene replicated.

The reference reseeds every pixel from (x, y, frame_index) (RayTracing.slang:71-72)
and averages frames with a running mean (:176-179), so progressive frames are
independent sample sets.  Rank r of N therefore renders frames r, r+N, r+2N, ...
into its own running mean; after k local frames the N-rank image is
(1/N) * sum_r mean_r  — one sum-reduce of W*H*4 floats per combine over RCCL
(`torch.distributed` backend "nccl" on ROCm), no collective on the data path
itself.  The same code runs on CPU tensors with the "gloo" backend.
"""
import torch
import torch.distributed as dist


def frame_shard(rank, world_size):
    """Renderer options for this rank: RNG frame = offset + k * stride."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return {"frame_stride": world_size, "frame_offset": rank}


def local_frames(rank, world_size, total_frames):
    """Frame indices this rank renders out of `total_frames` (round-robin)."""
    return list(range(rank, total_frames, world_size))


def combine(local_mean, out=None, dst=0, group=None):
    """Sum-reduce the per-rank running means onto `dst` and scale by 1/N.
    `local_mean` is left untouched (it is still being accumulated into).
    Returns the combined image on dst, None elsewhere.  Every rank must have
    rendered the same number of local frames."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if out is None:
        out = torch.empty_like(local_mean)
    out.copy_(local_mean)
    if world == 1:
        return out
    dist.reduce(out, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank(group) == dst:
        out.mul_(1.0 / world)
        return out
    return None
