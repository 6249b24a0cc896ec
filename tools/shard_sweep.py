#!/usr/bin/env python3
"""Time one rank's 1/N share of the benchmark image under the environment it is started with (the k_extend / round knobs are
read once per process): tools/shard_sweep.py [N=8] prints one line.  Used to look for settings that help small shards."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import RenderIntegrator, abi, parallel, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
scene_path, config_path = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
res = ri.resident(0)
cfg = parallel.shard_config(ri.config, 0, n, parallel.ROW_BLOCK)
cfg.flags = abi.RT_FLAG_TIME_KERNELS
rows, W = cfg.n_rows(), cfg.img_width
dev = torch.device("cuda", 0)
out = {"pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev), "sample_pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev)}
res.render(cfg, device_out=out)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, st = res.render(cfg, device_out=out)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
knobs = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RT_HIP_"))
print(f"1/{n} shard: {1e3*best:7.1f} ms  extend {st.trace_kernel_ms:6.1f} shade {st.shade_kernel_ms:6.1f} other {st.other_kernel_ms:5.1f} rounds {st.batches} traced/accepted {st.rays_traced/st.rays:.3f}  [{knobs}]", flush=True)
