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
H, W, B = int(sys.argv[2]), 5, int(sys.argv[3])
rows = parallel.shard_rows(H, rank, world, B)
local = torch.tensor([[[y * 1000 + x * 10 + c for c in range(3)] for x in range(W)] for y in rows], dtype=torch.float64).reshape(len(rows), W, 3)
full = parallel.gather_rows(local, H, rank, world, row_block=B)
# two planes stacked in front of the rows travel in the same single collective
stacked = torch.stack([local, -local])
both = parallel.gather_rows(stacked, H, rank, world, row_block=B, row_dim=1)
if rank == 0:
    assert torch.equal(both[0], full) and torch.equal(both[1], -full)
# a second gather of the same shape reuses the staging buffers of the first but must not write into its result
again = parallel.gather_rows(local * 2.0, H, rank, world, row_block=B)
if rank == 0:
    assert torch.equal(again, full * 2.0) and torch.equal(both[0], full)
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
    for h in (1, 2, 7, 33, 1200):
        for world in (1, 2, 3, 8):
            for block in (1, 2, 8):
                rows = [parallel.shard_rows(h, r, world, block) for r in range(world)]
                assert sorted(sum(rows, [])) == list(range(h))
                assert max(len(r) for r in rows) == parallel.max_rows(h, world, block)
                # the ctypes mirror of RtConfig counts and lists the same rows
                for r in range(world):
                    cfg = abi.RtConfig()
                    cfg.img_height, cfg.img_width = h, 4
                    s = parallel.shard_config(cfg, r, world, block)
                    assert s.rows() == rows[r] and s.n_rows() == len(rows[r])
    cfg = abi.RtConfig()
    cfg.img_height, cfg.img_width, cfg.row_first, cfg.row_stride = 10, 4, 0, 1
    s = parallel.shard_config(cfg, 2, 4)
    assert (s.row_first, s.row_stride, s.n_rows()) == (2, 4, 2) and cfg.row_stride == 1
    s = parallel.shard_config(cfg, 1, 2, 4)  # blocks of 4 rows dealt to 2 ranks: rank 1 owns rows 4..7
    assert s.rows() == [4, 5, 6, 7] and s.row_block == 4


def test_interleave_numpy():
    parts = [np.arange(4)[:, None] * 3 + r for r in range(3)]
    assert parallel.interleave_numpy(parts, 11).ravel().tolist() == list(range(11))
    rows = [parallel.shard_rows(21, r, 3, 4) for r in range(3)]
    parts = [np.array(r)[:, None] for r in rows]
    assert parallel.interleave_numpy(parts, 21, row_block=4).ravel().tolist() == list(range(21))


@pytest.mark.parametrize("height,block", [(8, 1), (7, 1), (21, 8), (16, 8)])
def test_gather_two_ranks_gloo(tmp_path, height, block):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + height + block), str(script), ROOT, str(height), str(block)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"OK {height}" in out.stdout


def test_gather_eight_ranks_gloo_with_uneven_shards(tmp_path):
    """The largest rank count a node has: 21 rows over 8 ranks (three rows on some ranks, two on others, padded for the collective)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), ROOT, "21", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK 21" in out.stdout


def test_oracle_renders_a_block_shard_as_the_rows_of_the_whole_image():
    """RtConfig.row_block on the CPU oracle (per-path streams: a shard's pixels do not depend on the sharding)."""
    from helpers import c1
    from oracle import oracle
    ri = c1(width=12, height=21, batch=4, max_batches=3, min_batches=0)
    whole, st = oracle.render(ri.scene, ri.config.copy(), 4)
    rays = 0
    for rank in range(2):
        cfg = parallel.shard_config(ri.config, rank, 2, 8)
        rows = parallel.shard_rows(21, rank, 2, 8)
        assert rows == ([0, 1, 2, 3, 4, 5, 6, 7, 16, 17, 18, 19, 20] if rank == 0 else list(range(8, 16)))
        out, s = oracle.render(ri.scene, cfg, 2)
        rays += s.rays
        assert np.array_equal(out["linear"], whole["linear"][rows]) and np.array_equal(out["samples"], whole["samples"][rows])
        p, t = oracle.primary_hits(ri.scene, cfg)
        p2, t2 = oracle.primary_hits_threads(ri.scene, cfg, 3)
        assert np.array_equal(p, p2) and np.array_equal(t, t2)
    assert rays == st.rays


def test_bench_starts_its_own_ranks_when_typed_plainly():
    """`python bench.py --gpus N` outside a launcher must start N ranks as child processes (never re-executing a
    process that has touched the GPU): --dry-run prints the plan without running it."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    plan = json.loads(out.stdout)
    cmd = plan["spawn"]
    assert plan["ranks"] == 4 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(bench)
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]       # the ranks get the same arguments
    assert plan["env"]["MASTER_ADDR"] == "127.0.0.1" and plan["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one GPU, or already inside a launcher: nothing is spawned
    one = subprocess.run([sys.executable, bench, "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert json.loads(one.stdout)["spawn"] is None
    inside = subprocess.run([sys.executable, bench, "--gpus", "4", "--dry-run"], env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                            capture_output=True, text=True, timeout=120)
    assert json.loads(inside.stdout) == {"spawn": None, "ranks": 4}
