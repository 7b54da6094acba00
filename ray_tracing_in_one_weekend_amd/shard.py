"""Row-interleaved image sharding across ranks and the framebuffer gather.

Image row j belongs to rank (j // band) % world.  Each rank renders its rows into a local
[rows, nx, 3] f32 buffer in HBM; one gather to the rank that keeps the frame (or one all_gather when every rank wants it; RCCL over xGMI
when the backend is "nccl", gloo in the CPU tests) brings the bands together and `deinterleave` restores row order.  The
path has no other exchange step: scene and RNG keys are replicated, pixels are independent.
"""
import numpy as np

DEFAULT_BAND = 8


def shard_rows(ny, band, world, rank):
    band = band or 1
    if world <= 1:
        return np.arange(ny)
    j = np.arange(ny)
    return j[(j // band) % world == rank]


def max_shard_rows(ny, band, world):
    return max(len(shard_rows(ny, band, world, r)) for r in range(max(world, 1)))


def deinterleave(gathered, ny, band, world):
    """gathered: list (per rank) of [rows_pad, nx, 3] arrays/tensors -> [ny, nx, 3] numpy or torch."""
    first = gathered[0]
    is_torch = hasattr(first, "new_zeros")
    if is_torch:
        out = first.new_zeros((ny,) + tuple(first.shape[1:]))
    else:
        out = np.zeros((ny,) + tuple(first.shape[1:]), dtype=first.dtype)
    for r in range(world):
        rows = shard_rows(ny, band, world, r)
        if is_torch:
            import torch
            idx = torch.as_tensor(rows, device=first.device, dtype=torch.long)
            out.index_copy_(0, idx, gathered[r][:len(rows)])
        else:
            out[rows] = gathered[r][:len(rows)]
    return out


def gather_framebuffer(local, ny, band, group=None, dst=None):
    """The framebuffer exchange of a frame (torch.distributed; backend nccl == RCCL over xGMI).
    `local` is a torch tensor [rows_local, nx, 3].  dst=None: all_gather, the full [ny, nx, 3] image on every rank.
    dst=r: gather to rank r only — the other ranks send their bands and return None; they neither receive the 8 x band
    buffers nor de-interleave a frame nobody reads (config 3: 99.5 MB of index_copy_ per rank per frame)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    pad_rows = max_shard_rows(ny, band, world)
    padded = local.new_zeros((pad_rows,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    if dst is None:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        return deinterleave(parts, ny, band, world)
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst, group=group)
    return deinterleave(parts, ny, band, world) if rank == dst else None
