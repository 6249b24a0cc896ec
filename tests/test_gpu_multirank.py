"""Two ranks render interleaved row shards of one image and gather them on rank 0:
the assembled image must equal a single-process render bit for bit.  Both ranks use
the one GPU of the test box, so the gather runs over gloo here; bench.py uses the
same sharding with RCCL (one GPU per rank)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import parallel
from helpers import c1
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
B = int(sys.argv[2])
ri = c1(width=40, height=33, batch=8, max_batches=5, min_batches=1)
res = ri.resident(0)
cfg = parallel.shard_config(ri.config, rank, world, B)
assert cfg.rows() == parallel.shard_rows(33, rank, world, B)
out, st = res.render(cfg)
H = cfg.img_height
full = {k: parallel.gather_rows(torch.from_numpy(np.ascontiguousarray(out[k]).astype(np.float64)), H, rank, world, row_block=B) for k in ("pixels", "sample_pixels", "linear")}
rays = torch.tensor([float(st.rays)], dtype=torch.float64)
dist.all_reduce(rays)
if rank == 0:
    one, st1 = res.render(ri.config.copy())
    for k in full:
        assert np.array_equal(full[k].numpy(), one[k]), k
    assert rays.item() == float(st1.rays)
    print("OK", int(rays.item()))
dist.destroy_process_group()
'''


@pytest.mark.gpu
@pytest.mark.parametrize("block", [1, 8])
def test_two_ranks_reproduce_the_single_process_image(tmp_path, block):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29611 + block), str(script), ROOT, str(block)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "OK" in out.stdout


RCCL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
except Exception as e:  # the box's RCCL, not this repository
    print("NO-RCCL", repr(e))
    sys.exit(0)
planes = torch.arange(2 * 5 * 7 * 3, dtype=torch.float64, device="cuda").reshape(2, 5, 7, 3)
t = planes.clone()
dist.all_reduce(t)
parts = [torch.empty_like(planes)]
dist.gather(planes, gather_list=parts, dst=0)
mx = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
dist.all_reduce(mx, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(t, planes) and torch.equal(parts[0], planes) and mx.tolist() == [1.5, 2.5]
print("OK")
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_the_rccl_backend_and_the_collectives_bench_uses_run_on_this_gpu(tmp_path):
    """bench.py --gpus N gathers with RCCL (backend "nccl"), which needs one GPU per rank and so cannot run two ranks on
    the test box's one GPU.  What can be run here is run: the backend initialised as bench.py does it, and the three
    collectives it issues (gather, all-reduce sum and max, barrier) on f64 tensors of the gathered buffer's shape."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    if "NO-RCCL" in out.stdout:
        pytest.skip("RCCL could not be initialised on this box: " + out.stdout.strip()[-300:])
    assert "OK" in out.stdout
