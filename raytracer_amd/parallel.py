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


ROW_BLOCK = 1  # rows dealt together; see shard_config


def shard_rows(height, rank, world, row_block=ROW_BLOCK):
    """Image rows owned by `rank`, in the order its buffers hold them: the blocks rank, rank + world, ...
    of `row_block` consecutive rows each."""
    b = max(1, row_block)
    n_blocks = (height + b - 1) // b
    return [y for blk in range(rank, n_blocks, world) for y in range(blk * b, min((blk + 1) * b, height))]


def shard_config(cfg, rank, world, row_block=ROW_BLOCK):
    """The configuration of `rank`'s shard: blocks of `row_block` rows dealt round-robin (RtConfig.row_first /
    row_stride / row_block).  A single process renders the whole image whatever the block size."""
    c = cfg.copy()
    c.row_first = rank
    c.row_stride = world
    c.row_block = max(1, row_block) if world > 1 else 0
    return c


def max_rows(height, world, row_block=ROW_BLOCK):
    return max(len(shard_rows(height, r, world, row_block)) for r in range(world))


_buffers = {}  # persistent staging of gather_rows: one set per (shape, dtype, device, sharding)


def gather_rows(local, height, rank, world, dst=0, row_block=ROW_BLOCK, row_dim=0):
    """Gather per-rank shards into the full image on `dst` with ONE collective.

    `local` is a torch tensor (CUDA for nccl, CPU for gloo) whose dimension `row_dim` runs over the
    rank's rows (shard_rows order); other leading dimensions may stack several planes (image and
    heat-map) so that they travel together.  Ranks may own different row counts; shards are padded
    to the largest for the collective.  Returns the assembled tensor on `dst`, None elsewhere.

    A contiguous shard with the largest row count is sent as it is.  Whatever the call needs beyond
    that -- the padded send buffer of a shorter shard (its pad rows zeroed once), the receive buffers,
    the row indices -- is allocated at the first call for a shape and reused by the later ones; only the
    assembled tensor is new every time (uninitialised memory from torch's allocator, every row then written
    once), because the caller keeps it.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    mr = max_rows(height, world, row_block)
    rows_here = local.shape[row_dim]
    key = (tuple(local.shape), row_dim, local.dtype, str(local.device), height, rank, world, dst, row_block)
    buf = _buffers.get(key)
    if buf is None:
        shape = list(local.shape)
        shape[row_dim] = mr
        buf = {}
        if rows_here < mr or not local.is_contiguous():
            buf["send"] = torch.empty(shape, dtype=local.dtype, device=local.device)
            buf["send"].narrow(row_dim, rows_here, mr - rows_here).zero_()
        if rank == dst:
            buf["parts"] = [torch.empty(shape, dtype=local.dtype, device=local.device) for _ in range(world)]
            buf["rows"] = [torch.as_tensor(shard_rows(height, r, world, row_block), device=local.device) for r in range(world)]
        _buffers[key] = buf
    send = local
    if "send" in buf:
        send = buf["send"]
        send.narrow(row_dim, 0, rows_here).copy_(local)
    if rank == dst:
        dist.gather(send, gather_list=buf["parts"], dst=dst)
        shape = list(local.shape)
        shape[row_dim] = height
        full = torch.empty(shape, dtype=local.dtype, device=local.device)
        for r in range(world):
            rows = buf["rows"][r]
            full.index_copy_(row_dim, rows, buf["parts"][r].narrow(row_dim, 0, len(rows)))
        return full
    dist.gather(send, gather_list=None, dst=dst)
    return None


def interleave_numpy(parts, height, row_block=ROW_BLOCK):
    """CPU equivalent of the de-interleave step: parts[r] holds the rows shard_rows(height, r, len(parts))."""
    world = len(parts)
    full = np.empty((height,) + parts[0].shape[1:], dtype=parts[0].dtype)
    for r in range(world):
        rows = shard_rows(height, r, world, row_block)
        full[rows] = parts[r][: len(rows)]
    return full


class RankComm:
    """A communicator of the C library (rt_hip_comm_init): the gather runs inside librt_hip.so over RCCL, without
    torch.distributed.  `comm_id` is made once by rank 0 (RankComm.new_id()) and handed to every rank by the caller."""

    @staticmethod
    def new_id():
        import ctypes as C
        from . import abi
        buf = C.create_string_buffer(abi.RT_COMM_ID_BYTES)
        abi.check_hip(abi.hip_lib().rt_hip_comm_id(buf))
        return buf.raw

    def __init__(self, comm_id, rank, world, device):
        import ctypes as C
        from . import abi
        assert comm_id is None or len(comm_id) == abi.RT_COMM_ID_BYTES  # None: this process plays every rank in turn (no RCCL)
        self._c = C.c_void_p()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        abi.check_hip(abi.hip_lib().rt_hip_comm_init(comm_id, self.rank, self.world, self.device, C.byref(self._c)))

    def close(self):
        from . import abi
        if self._c:
            abi.hip_lib().rt_hip_comm_free(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, resident, cfg, device_out=None):
        """rt_hip_render_ranks: collective.  Rank 0 gets the whole image (numpy arrays `pixels`, `sample_pixels`, or written
        into the torch CUDA tensors of `device_out`); the other ranks get None.  Returns (image or None, RtStats)."""
        import ctypes as C
        from . import abi
        stats = abi.RtStats()
        img = abi.RtImage()
        out = None
        if self.rank == 0:
            if device_out is not None:
                img.memory = abi.RT_MEM_DEVICE
                img.pixels = device_out["pixels"].data_ptr()
                if device_out.get("sample_pixels") is not None:
                    img.sample_pixels = device_out["sample_pixels"].data_ptr()
                out = device_out
            else:
                out = {"pixels": np.empty((cfg.img_height, cfg.img_width, 3), dtype=np.float64),
                       "sample_pixels": np.empty((cfg.img_height, cfg.img_width, 3), dtype=np.float64)}
                img.memory = abi.RT_MEM_HOST
                img.pixels = out["pixels"].ctypes.data
                img.sample_pixels = out["sample_pixels"].ctypes.data
        abi.check_hip(abi.hip_lib().rt_hip_render_ranks(resident._h, self._c, C.byref(cfg), C.byref(img) if self.rank == 0 else None, C.byref(stats)))
        return out, stats
