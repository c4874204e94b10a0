"""Sharding of the path across GPUs (one process + one gsx context per GPU, splats replicated).

Two natural shardings, no data-path collective until the frames are finished:
  * camera views   : rank r renders views r, r + R, r + 2R, ...
  * tile-row bands : rank r renders rows [y0, y1) of one frame, band edges on tile rows
The single exchange is the gather of RGBA8 pixels to rank 0 (torch.distributed: nccl = RCCL over
xGMI on the GPU box, gloo in the CPU tests).
"""
TILE_H = 16


def shard_views(n_views, rank, world):
    """Indices of the views rank `rank` renders (round-robin keeps per-rank load equal)."""
    return list(range(rank, n_views, world))


def shard_bands(height, world, tile_h=TILE_H):
    """[(y0, y1)] per rank: contiguous bands aligned to tile rows, covering [0, height) exactly.
    Ranks beyond the number of tile rows get an empty band (y0 == y1)."""
    rows = (height + tile_h - 1) // tile_h
    bands, start = [], 0
    for r in range(world):
        n = rows // world + (1 if r < rows % world else 0)
        y0, y1 = min(start * tile_h, height), min((start + n) * tile_h, height)
        bands.append((y0, y1))
        start += n
    return bands


def gather_to_rank0(dist, tensor, rank, world, bufs=None):
    """Gather equal-sized uint8 tensors to rank 0.  Returns the list on rank 0, None elsewhere."""
    import torch
    if rank == 0:
        if bufs is None:
            bufs = [torch.empty_like(tensor) for _ in range(world)]
        dist.gather(tensor, bufs, dst=0)
        return bufs
    dist.gather(tensor, None, dst=0)
    return None


def compose_bands(frames, bands, width):
    """Rank-0 side of band sharding: frames[r] is rank r's full-size RGBA8 buffer with its band filled."""
    out = frames[0].clone()
    row = width * 4
    for r, (y0, y1) in enumerate(bands):
        if r and y1 > y0:
            out[y0 * row:y1 * row] = frames[r][y0 * row:y1 * row]
    return out
