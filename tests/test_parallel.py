"""Row sharding over ranks and the gather to rank 0, exercised with two gloo
processes on CPU (the data path itself needs a GPU; here the rows carry a
deterministic pattern instead of pixels)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import abi, parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H, W = int(sys.argv[2]), 5
rows = parallel.shard_rows(H, rank, world)
local = torch.tensor([[[y * 1000 + x * 10 + c for c in range(3)] for x in range(W)] for y in rows], dtype=torch.float64).reshape(len(rows), W, 3)
full = parallel.gather_rows(local, H, rank, world)
counts = torch.tensor([float(len(rows))], dtype=torch.float64)
dist.all_reduce(counts)
if rank == 0:
    expect = torch.tensor([[[y * 1000 + x * 10 + c for c in range(3)] for x in range(W)] for y in range(H)], dtype=torch.float64)
    assert torch.equal(full, expect), "gathered image differs"
    assert counts.item() == H
    print("OK", H)
else:
    assert full is None
dist.destroy_process_group()
'''


def test_row_shards_partition_the_image():
    for h in (1, 2, 7, 1200):
        for world in (1, 2, 3, 8):
            rows = [parallel.shard_rows(h, r, world) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(h))
            assert max(len(r) for r in rows) == parallel.max_rows(h, world)
    cfg = abi.RtConfig()
    cfg.img_height, cfg.img_width, cfg.row_first, cfg.row_stride = 10, 4, 0, 1
    s = parallel.shard_config(cfg, 2, 4)
    assert (s.row_first, s.row_stride, s.n_rows()) == (2, 4, 2) and cfg.row_stride == 1


def test_interleave_numpy():
    parts = [np.arange(4)[:, None] * 3 + r for r in range(3)]
    assert parallel.interleave_numpy(parts, 11).ravel().tolist() == list(range(11))


@pytest.mark.parametrize("height", [8, 7])
def test_gather_two_ranks_gloo(tmp_path, height):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + height), str(script), ROOT, str(height)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"OK {height}" in out.stdout
