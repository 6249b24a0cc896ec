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
