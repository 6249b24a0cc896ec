#!/usr/bin/env python3
"""Experiment: one rank's 1/N share of the headline image rendered as K interleaved sub-shards at the same time (each
from its own host thread with its own scene handle, stream and working memory) against the share rendered alone:
do the tails of one sub-shard's short launches fill with the other's work?

  python3 tools/concurrent_subshards.py [N]
"""
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import RenderIntegrator, parallel, scenes
from raytracer_amd.integrator import ResidentScene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
scene_path, config_path = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
for K in (1, 2, 4, 1, 2):
    handles = [ResidentScene(ri.scene, 0) for _ in range(K)]
    cfgs = [parallel.shard_config(ri.config, k * N, K * N) for k in range(K)]  # rows 0::N as rows (k*N)::(K*N)
    stats = [None] * K

    def work(r):
        _, stats[r] = handles[r].render(cfgs[r], want_linear=False, want_samples=False)

    best = 1e9
    for rep in range(4):
        ts = [threading.Thread(target=work, args=(r,)) for r in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    rays = sum(s.rays for s in stats)
    print(f"1/{N} of the image as K={K} concurrent sub-shards: {best*1e3:8.1f} ms  {rays/best/1e6:8.1f} Mrays/s", flush=True)
    del handles
