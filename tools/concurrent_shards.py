#!/usr/bin/env python3
"""Experiment: render the benchmark image as K row-shards at the same time on ONE GPU, each
from its own host thread with its own scene handle (own stream and workspace), to see whether
kernels of different shards overlap usefully (k_extend is latency-bound, k_shade is not)."""
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import RenderIntegrator, abi, parallel, scenes
from raytracer_amd.integrator import ResidentScene

workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
scene_path, config_path = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
for K in (1, 2, 3, 4):
    handles = [ResidentScene(ri.scene, 0) for _ in range(K)]
    cfgs = [parallel.shard_config(ri.config, r, K) for r in range(K)]
    stats = [None] * K

    def work(r):
        _, stats[r] = handles[r].render(cfgs[r])

    for rep in range(2):
        ts = [threading.Thread(target=work, args=(r,)) for r in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    rays = sum(s.rays for s in stats)
    print(f"K={K}: {dt*1e3:8.1f} ms  {rays/dt/1e6:8.1f} Mrays/s", flush=True)
    del handles
