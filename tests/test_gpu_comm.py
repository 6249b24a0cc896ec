"""rt_hip_render_ranks: one image rendered by several processes, gathered inside the C library over RCCL.  The test box
has one GPU, so what runs here is a communicator of ONE rank (every step of the path except the sends and receives
themselves: render into the rank's device buffer, the counters' exchange, the assembly of the image on the device, the
hand-over) for single rows and for row blocks, and the argument checks.  Two ranks need two GPUs: RCCL refuses two ranks
on one device."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, parallel

from helpers import c1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", [1, 8])
def test_one_rank_communicator_renders_the_single_process_image(block):
    ri = c1(width=40, height=33, batch=8, max_batches=5, min_batches=1)
    res = ri.resident(0)
    comm = parallel.RankComm(parallel.RankComm.new_id(), 0, 1, 0)
    cfg = ri.config.copy()
    cfg.row_block = block
    cfg.row_first, cfg.row_stride = 5, 7                            # set by the call, whatever the caller left there
    out, st = comm.render(res, cfg)
    one, st1 = res.render(ri.config.copy())
    assert np.array_equal(out["pixels"], one["pixels"]) and np.array_equal(out["sample_pixels"], one["sample_pixels"])
    assert st.rays == st1.rays and st.samples == st1.samples and list(st.rays_by_depth) == list(st1.rays_by_depth)
    # again on the same communicator (its buffers are kept), into device memory this time
    import torch
    planes = torch.empty(2, 33, 40, 3, dtype=torch.float64, device="cuda")
    comm.render(res, cfg, device_out={"pixels": planes[0], "sample_pixels": planes[1]})
    torch.cuda.synchronize()
    assert np.array_equal(planes[0].cpu().numpy(), one["pixels"]) and np.array_equal(planes[1].cpu().numpy(), one["sample_pixels"])
    comm.close()


def test_render_ranks_rejects_what_it_cannot_gather():
    ri = c1(width=16, height=16, batch=2)
    res = ri.resident(0)
    lib = abi.hip_lib()
    cid = parallel.RankComm.new_id()
    h = C.c_void_p()
    assert lib.rt_hip_comm_init(cid, 1, 1, 0, C.byref(h)) == abi.RT_ERR_BAD_ARG          # rank outside [0, world)
    assert lib.rt_hip_comm_init(cid, 0, 1, 99, C.byref(h)) == abi.RT_ERR_BAD_ARG         # no such device
    comm = parallel.RankComm(cid, 0, 1, 0)
    cfg = ri.config.copy()
    img = abi.RtImage()
    stats = abi.RtStats()
    assert lib.rt_hip_render_ranks(res._h, comm._c, C.byref(cfg), C.byref(img), C.byref(stats)) == abi.RT_ERR_BAD_ARG   # rank 0 without pixels
    px = np.empty((16, 16, 3))
    lin = np.empty((16, 16, 3))
    img.pixels, img.linear, img.memory = px.ctypes.data, lin.ctypes.data, abi.RT_MEM_HOST
    assert lib.rt_hip_render_ranks(res._h, comm._c, C.byref(cfg), C.byref(img), C.byref(stats)) == abi.RT_ERR_UNSUPPORTED
    assert b"render_line" in lib.rt_hip_last_error()
    comm.close()


@pytest.mark.parametrize("world,block", [(2, 1), (3, 1), (4, 8), (5, 2), (40, 1)])
def test_one_process_playing_every_rank_assembles_the_single_process_image(world, block):
    """A communicator made without an id renders the shards of all ranks in turn and puts them together exactly as rank 0 of a
    real job does (same row bookkeeping, same assembly kernel): the image must be the single-process one for any number of
    ranks and any block size, also when some ranks get no row at all (40 ranks, 33 rows)."""
    ri = c1(width=40, height=33, batch=8, max_batches=5, min_batches=1)
    res = ri.resident(0)
    comm = parallel.RankComm(None, 0, world, 0)
    cfg = ri.config.copy()
    cfg.row_block = block
    out, st = comm.render(res, cfg)
    one, st1 = res.render(ri.config.copy())
    assert np.array_equal(out["pixels"], one["pixels"]) and np.array_equal(out["sample_pixels"], one["sample_pixels"])
    assert st.rays == st1.rays and st.samples == st1.samples
    comm.close()
