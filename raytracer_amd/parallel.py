"""Multi-GPU sharding of the render: one process per GPU, image rows dealt
round-robin over the ranks, finished rows gathered to rank 0.

The reference's unit of parallel work is the image row
(/root/reference/src/render/integrator.rs:196-206, 226-243); rows are
independent given the read-only scene, and in counter RNG mode every pixel
owns its random streams, so a sharded render is bit-identical to a
single-GPU one.  Rows are interleaved (rank r owns rows r, r+N, ...) because
the adaptive sampler's cost is spatially clustered.  The only exchange is the
final gather (RCCL over xGMI when the backend is "nccl"; gloo on CPU).
"""
import numpy as np


def shard_rows(height, rank, world):
    """Row indices owned by `rank`."""
    return list(range(rank, height, world))


def shard_config(cfg, rank, world):
    c = cfg.copy()
    c.row_first = rank
    c.row_stride = world
    return c


def max_rows(height, world):
    return (height + world - 1) // world


def gather_rows(local, height, rank, world, dst=0):
    """Gather per-rank row blocks [rows_r, W, C] into the full [H, W, C] image on `dst`.

    `local` is a torch tensor (CUDA for nccl, CPU for gloo).  Ranks may own
    different row counts when world does not divide height; blocks are padded to
    the largest count for the collective.  Returns the assembled tensor on
    `dst`, None elsewhere.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    mr = max_rows(height, world)
    pad = local
    if local.shape[0] < mr:
        pad = torch.zeros((mr,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    pad = pad.contiguous()
    if rank == dst:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=parts, dst=dst)
        full = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            n = len(range(r, height, world))
            full[r::world] = parts[r][:n]
        return full
    dist.gather(pad, gather_list=None, dst=dst)
    return None


def interleave_numpy(parts, height):
    """CPU equivalent of the de-interleave step: parts[r] holds rows r::len(parts)."""
    world = len(parts)
    full = np.empty((height,) + parts[0].shape[1:], dtype=parts[0].dtype)
    for r in range(world):
        full[r::world] = parts[r][: len(range(r, height, world))]
    return full
