"""Multi-GPU sharding of the render path: one process per GPU, scene replicated.

The reference reseeds every pixel from (x, y, frame_index) (RayTracing.slang:71-72)
and averages frames with a running mean (:176-179), so progressive frames are
independent sample sets.  Rank r of N therefore renders frames r, r+N, r+2N, ...
into its own running mean; after k local frames the N-rank image is
(1/N) * sum_r mean_r  — one sum-reduce of W*H*4 floats per combine over RCCL
(`torch.distributed` backend "nccl" on ROCm), no collective on the data path
itself.  The same code runs on CPU tensors with the "gloo" backend.

A second mode shards by image rows instead (`row_shard` / `gather_rows`): rank r
renders the contiguous row band [begin, end) of every frame through
rt_render_params.row_begin/row_end and the bands are all-gathered.  No floating
point is re-associated, so the N-rank image equals the 1-rank image bit for bit;
the price is load balance (a band of sky finishes early), which is why frames
are the default for the bench.

A third mode interleaves the rows (`row_interleave` / `gather_interleaved`): the
image is cut into groups of 8 rows and rank r renders groups r, r+N, r+2N, ...
(rt_render_params.row_group_stride), which spreads sky and geometry evenly over
the ranks; the groups are all-gathered, again bit for bit equal to one GPU.
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


def row_shard(rank, world_size, height, align=8):
    """Row band [begin, end) of rank `rank`: bands are multiples of `align` rows (the
    kernels fetch 8x8 pixel tiles) except the last, and differ by at most `align`."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    units = (height + align - 1) // align
    lo = (units * rank) // world_size
    hi = (units * (rank + 1)) // world_size
    return min(lo * align, height), min(hi * align, height)


def gather_rows(image, rank_rows, group=None):
    """All-gather the row bands: `image` is this rank's full-size (H, W, C) tensor with
    its own band filled in; `rank_rows[r]` = (begin, end) of rank r.  Returns the
    complete image on every rank (in place in `image`)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return image
    rank = dist.get_rank(group)
    longest = max(e - b for b, e in rank_rows)
    if longest == 0:
        return image
    shape = (longest,) + tuple(image.shape[1:])
    mine = torch.zeros(shape, dtype=image.dtype, device=image.device)
    b, e = rank_rows[rank]
    mine[: e - b].copy_(image[b:e])
    parts = [torch.empty(shape, dtype=image.dtype, device=image.device) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    for r, (rb, re) in enumerate(rank_rows):
        if r != rank and re > rb:
            image[rb:re].copy_(parts[r][: re - rb])
    return image


ROW_GROUP = 8  # rows per interleaving unit: the height of the kernels' pixel tiles


def row_interleave(rank, world_size, height):
    """Render options of rank `rank` for interleaved 8-row groups: dict with row_begin, row_end and
    row_group_stride (rt_render_params), or None if the image has fewer groups than ranks and this
    rank gets none."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if rank * ROW_GROUP >= height:
        return None
    return {"row_begin": rank * ROW_GROUP, "row_end": height, "row_group_stride": world_size}


def interleaved_rows(rank, world_size, height):
    """Row indices rank `rank` renders under `row_interleave` (for tests and bookkeeping)."""
    rows = []
    for g in range(rank, (height + ROW_GROUP - 1) // ROW_GROUP, world_size):
        rows.extend(range(g * ROW_GROUP, min((g + 1) * ROW_GROUP, height)))
    return rows


def gather_interleaved(image, group=None):
    """All-gather the interleaved row groups: `image` is this rank's full-size (H, W, C) tensor
    with its own groups filled in.  Returns the complete image on every rank (in place)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return image
    rank = dist.get_rank(group)
    height = image.shape[0]
    groups = (height + ROW_GROUP - 1) // ROW_GROUP
    per_rank = (groups + world - 1) // world
    shape = (per_rank * ROW_GROUP,) + tuple(image.shape[1:])
    mine = torch.zeros(shape, dtype=image.dtype, device=image.device)
    for k, g in enumerate(range(rank, groups, world)):
        rows = min(ROW_GROUP, height - g * ROW_GROUP)
        mine[k * ROW_GROUP: k * ROW_GROUP + rows].copy_(image[g * ROW_GROUP: g * ROW_GROUP + rows])
    parts = [torch.empty(shape, dtype=image.dtype, device=image.device) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    for r in range(world):
        if r == rank:
            continue
        for k, g in enumerate(range(r, groups, world)):
            rows = min(ROW_GROUP, height - g * ROW_GROUP)
            image[g * ROW_GROUP: g * ROW_GROUP + rows].copy_(parts[r][k * ROW_GROUP: k * ROW_GROUP + rows])
    return image


class InterleavedGather:
    """The all-gather of `gather_interleaved` with everything per-step precomputed and nothing in place, so that it
    can run on a side stream while the next launch already renders: `pack` copies this rank's row groups out of the
    accumulation image (one gather kernel, on the render stream), `exchange` all-gathers the packed groups of every
    rank and scatters them into `full`, a complete image of its own (one collective + one scatter kernel, on the
    communication stream).  Two packed buffers alternate, so step k+1's `pack` never overwrites what step k's
    collective still reads.  Values are moved, never re-associated: `full` equals the one-GPU image bit for bit.
    Works on CPU tensors with gloo (tests) and on GPU tensors with RCCL."""

    def __init__(self, height, width, channels=4, dtype=torch.float32, device="cpu", group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height = height
        groups = (height + ROW_GROUP - 1) // ROW_GROUP
        self.per_rank = (groups + self.world - 1) // self.world   # padded: every rank sends the same size
        rows_per_rank = self.per_rank * ROW_GROUP
        # row r*rows_per_rank + j of the gathered buffer = image row src[r][j]; padding rows point at row `height`
        # of a one-row-taller staging view, i.e. they are dropped on the way back
        idx = torch.full((self.world, rows_per_rank), height, dtype=torch.int64)
        for r in range(self.world):
            rows = interleaved_rows(r, self.world, height)
            idx[r, : len(rows)] = torch.tensor(rows, dtype=torch.int64)
        self.my_rows = idx[self.rank].clamp(max=height - 1).to(device)        # pack: padding re-reads the last row
        self.all_rows = idx.reshape(-1).to(device)                            # unpack: padding lands in the spare row
        shape = (rows_per_rank, width, channels)
        self.send = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(2)]
        self.recv = torch.zeros((self.world * rows_per_rank, width, channels), dtype=dtype, device=device)
        self.full_padded = torch.zeros((height + 1, width, channels), dtype=dtype, device=device)
        self.full = self.full_padded[:height]
        self.step = 0

    def pack(self, image):
        """Copies this rank's row groups out of `image` (H, W, C); returns the slot for `exchange`."""
        slot = self.step & 1
        self.step += 1
        torch.index_select(image, 0, self.my_rows, out=self.send[slot])
        return slot

    def exchange(self, slot):
        """All-gathers slot `slot` of every rank into `self.full` (complete image on every rank)."""
        if self.world == 1:
            self.recv.copy_(self.send[slot])
        elif self.recv.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal on a shared GPU (bench.py --rehearse): gloo is a CPU transport, stage through host memory
            got = torch.empty(self.recv.shape, dtype=self.recv.dtype)
            dist.all_gather_into_tensor(got, self.send[slot].cpu(), group=self.group)
            self.recv.copy_(got)
        else:
            dist.all_gather_into_tensor(self.recv, self.send[slot], group=self.group)
        self.full_padded.index_copy_(0, self.all_rows, self.recv)
        return self.full
